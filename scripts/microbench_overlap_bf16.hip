// Micro-benchmark (round 4): does a bf16 MFMA (v_mfma_f32_16x16x32_bf16: 8 192 multiply-adds, ~16 cycles) run UNDER the fp32
// VALU work of another wave of the same SIMD on gfx950?  The fp32 MFMA does not (scripts/microbench_overlap.hip: it occupies
// the fp32 multipliers the vector instructions use).  Same protocol:
//   m : waves 0-3 of a 512-thread block (one per SIMD) issue bf16 MFMAs only (4 accumulators), waves 4-7 leave
//   v : waves 4-7 issue v_fma_f32 only (16 per MFMA slot of the other mode), waves 0-3 leave
//   c : both at once (two waves per SIMD, one of each kind)        overlap: t_c ~ max(t_m, t_v); none: t_m + t_v
//   c2: as c with TWO bf16 MFMAs per 16 FMAs
//   d : one wave per SIMD issues both kinds interleaved in ONE stream (1 MFMA, then 16 independent FMAs)
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_overlap_bf16.hip -o /tmp/mbob && /tmp/mbob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FMA8(a) a##0 = __builtin_fmaf(a##0, m, c); a##1 = __builtin_fmaf(a##1, m, c); a##2 = __builtin_fmaf(a##2, m, c); a##3 = __builtin_fmaf(a##3, m, c); \
                a##4 = __builtin_fmaf(a##4, m, c); a##5 = __builtin_fmaf(a##5, m, c); a##6 = __builtin_fmaf(a##6, m, c); a##7 = __builtin_fmaf(a##7, m, c);
#define FMA16(a) FMA8(a) FMA8(a)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    f32x4 acc0 = {a0, a1, a2, a3}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.001f * (threadIdx.x + i)); B[i] = (__bf16)(1.0f - 0.002f * i); }
    const int per = (MODE == 3) ? 2 : 1;
    const bool do_m = (MODE == 0 || MODE == 2 || MODE == 3) && wave < 4, do_v = (MODE == 1 || MODE == 2 || MODE == 3) && wave >= 4;
    if (do_m) {
        for (int i = 0; i < iters * per; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc3, 0, 0, 0);
        }
    }
    if (do_v) {
        for (int i = 0; i < iters; ++i) { FMA16(a) FMA16(a) FMA16(a) FMA16(a) }
    }
    if (MODE == 4 && wave < 4) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc0, 0, 0, 0);
            FMA16(a)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc1, 0, 0, 0);
            FMA16(a)
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc2, 0, 0, 0);
            FMA16(a)
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc3, 0, 0, 0);
            FMA16(a)
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

template <int MODE>
float run(const char* name, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 100, 1.0f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s %.3f ms   (%.1f cycles @2.4GHz per slot)\n", name, best, best * 1e-3 * 2.4e9 / (iters * 4.0));
    return best;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0>("m: bf16 MFMA waves alone (1 per slot)", out);
    run<1>("v: FMA waves alone (16 per slot)", out);
    run<2>("c: one bf16-MFMA wave + one FMA wave per SIMD", out);
    run<3>("c2: the same with 2 MFMAs per 16 FMAs", out);
    run<4>("d: one wave per SIMD, interleaved stream", out);
    return 0;
}
