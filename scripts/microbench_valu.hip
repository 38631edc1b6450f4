// Micro-benchmark: sustained issue rate of v_fma_f32, v_pk_fma_f32, v_sqrt_f32, v_min_f32, v_fma_f64, v_rcp_f64 and
// v_mfma_f64_16x16x4_f64 per SIMD on gfx950.
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_valu.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 0.001f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
                p4 = __builtin_elementwise_fma(p4, pm, pc); p5 = __builtin_elementwise_fma(p5, pm, pc);
                p6 = __builtin_elementwise_fma(p6, pm, pc); p7 = __builtin_elementwise_fma(p7, pm, pc);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_sqrtf(a0); a1 = __builtin_amdgcn_sqrtf(a1); a2 = __builtin_amdgcn_sqrtf(a2); a3 = __builtin_amdgcn_sqrtf(a3);
                a4 = __builtin_amdgcn_sqrtf(a4); a5 = __builtin_amdgcn_sqrtf(a5); a6 = __builtin_amdgcn_sqrtf(a6); a7 = __builtin_amdgcn_sqrtf(a7);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fminf(a0 + c, a1); a1 = __builtin_fminf(a1 + c, a2); a2 = __builtin_fminf(a2 + c, a3); a3 = __builtin_fminf(a3 + c, a4);
                a4 = __builtin_fminf(a4 + c, a5); a5 = __builtin_fminf(a5 + c, a6); a6 = __builtin_fminf(a6 + c, a7); a7 = __builtin_fminf(a7 + c, a0);
            }
        }
    }
    if (MODE >= 4) {
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
        const double dm = 0.999, dc = 0.001;
        typedef double d4v __attribute__((ext_vector_type(4)));
        d4v acc0 = {d0, d1, d2, d3}, acc1 = {d4, d5, d6, d7};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 4) {
                    d0 = __builtin_fma(d0, dm, dc); d1 = __builtin_fma(d1, dm, dc); d2 = __builtin_fma(d2, dm, dc); d3 = __builtin_fma(d3, dm, dc);
                    d4 = __builtin_fma(d4, dm, dc); d5 = __builtin_fma(d5, dm, dc); d6 = __builtin_fma(d6, dm, dc); d7 = __builtin_fma(d7, dm, dc);
                } else if (MODE == 5) {
                    d0 = __builtin_amdgcn_rcp(d0); d1 = __builtin_amdgcn_rcp(d1); d2 = __builtin_amdgcn_rcp(d2); d3 = __builtin_amdgcn_rcp(d3);
                    d4 = __builtin_amdgcn_rcp(d4); d5 = __builtin_amdgcn_rcp(d5); d6 = __builtin_amdgcn_rcp(d6); d7 = __builtin_amdgcn_rcp(d7);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d3, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d4, d5, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d6, d7, acc1, 0, 0, 0);
                }
            }
        }
        a0 = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* out, int instr_per_iter) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * instr_per_iter;       // wave-instructions issued
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;             // per SIMD
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f G wave-instr/s/SIMD  => %.2f cycles/wave-instr @2.4GHz\n", name,
           waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma", w, out, 64);
        run<1>("v_pk_fma", w, out, 64);
        run<2>("v_sqrt", w, out, 64);
        run<3>("add+min", w, out, 128);
        run<4>("v_fma_f64", w, out, 64);
        run<5>("v_rcp_f64", w, out, 64);
        run<6>("mfma_f64", w, out, 32);
    }
    return 0;
}
