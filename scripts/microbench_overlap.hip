// Micro-benchmark: does the f32 matrix pipe (v_mfma_f32_16x16x4_f32) work UNDER the VALU work of the same SIMD on gfx950?
//   m : waves 0-3 of a 512-thread block (one per SIMD) issue MFMAs only, waves 4-7 leave
//   v : waves 4-7 issue v_fma_f32 only (16 per MFMA of the other mode: the same nominal time), waves 0-3 leave
//   c : both at once (two waves per SIMD, one of each kind)        overlap: t_c ~ max(t_m, t_v); none: t_m + t_v
//   d : waves 0-3 issue both kinds interleaved in ONE instruction stream (1 MFMA, then 16 independent FMAs), 4-7 leave
//   s : waves 0-3 issue the MFMAs of an iteration first and its FMAs behind them (4 + 64), 4-7 leave
//   fp64 m / v / c : the same three with v_mfma_f64_16x16x4_f64 and v_fma_f64
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_overlap.hip -o /tmp/mbo && /tmp/mbo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FMA8(a) a##0 = __builtin_fmaf(a##0, m, c); a##1 = __builtin_fmaf(a##1, m, c); a##2 = __builtin_fmaf(a##2, m, c); a##3 = __builtin_fmaf(a##3, m, c); \
                a##4 = __builtin_fmaf(a##4, m, c); a##5 = __builtin_fmaf(a##5, m, c); a##6 = __builtin_fmaf(a##6, m, c); a##7 = __builtin_fmaf(a##7, m, c);
#define FMA16(a) FMA8(a) FMA8(a)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    f32x4 acc0 = {a0, a1, a2, a3}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool do_m = (MODE == 0 || MODE == 2) && wave < 4, do_v = (MODE == 1 || MODE == 2) && wave >= 4;
    if (do_m) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, a3, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, a5, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a6, a7, acc3, 0, 0, 0);
        }
    }
    if (do_v) {
        for (int i = 0; i < iters; ++i) { FMA16(a) FMA16(a) FMA16(a) FMA16(a) }
    }
    if (MODE == 3 && wave < 4) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, m, acc0, 0, 0, 0);
            FMA16(a)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, c, acc1, 0, 0, 0);
            FMA16(a)
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(m, seed, acc2, 0, 0, 0);
            FMA16(a)
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, seed, acc3, 0, 0, 0);
            FMA16(a)
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    if (MODE == 4 && wave < 4) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, m, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, c, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(m, seed, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, seed, acc3, 0, 0, 0);
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));     // the FMAs wait for the products
            FMA16(a) FMA16(a) FMA16(a) FMA16(a)
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

// the same question in fp64: v_mfma_f64_16x16x4_f64 (64 cycles) against 16 v_fma_f64 (4 cycles each) per slot
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define DFMA8(a) a##0 = __builtin_fma(a##0, m, c); a##1 = __builtin_fma(a##1, m, c); a##2 = __builtin_fma(a##2, m, c); a##3 = __builtin_fma(a##3, m, c); \
                 a##4 = __builtin_fma(a##4, m, c); a##5 = __builtin_fma(a##5, m, c); a##6 = __builtin_fma(a##6, m, c); a##7 = __builtin_fma(a##7, m, c);
#define DFMA16(a) DFMA8(a) DFMA8(a)
template <int MODE>
__global__ __launch_bounds__(512) void k64(float* out, int iters, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const double m = 0.999, c = 0.001;
    f64x4 acc0 = {d0, d1, d2, d3}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool do_m = (MODE == 0 || MODE == 2) && wave < 4, do_v = (MODE == 1 || MODE == 2) && wave >= 4;
    if (do_m) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d3, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(d4, d5, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(d6, d7, acc3, 0, 0, 0);
        }
    }
    if (do_v) {
        for (int i = 0; i < iters; ++i) { DFMA16(d) DFMA16(d) DFMA16(d) DFMA16(d) }
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + acc0[0] + acc1[1] + acc2[2] + acc3[3]);
}

template <int MODE>
float run64(const char* name, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k64<MODE>, dim3(256), dim3(512), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k64<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms   (%.1f cycles @2.4GHz per {1 MFMA + 16 FMA} slot)\n", name, best, best * 1e-3 * 2.4e9 / (iters * 4.0));
    return best;
}

template <int MODE>
float run(const char* name, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms   (%.1f cycles @2.4GHz per {1 MFMA + 16 FMA} slot)\n", name, best, best * 1e-3 * 2.4e9 / (iters * 4.0));
    return best;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0>("m: MFMA waves alone", out);
    run<1>("v: FMA waves alone", out);
    run<2>("c: one MFMA wave + one FMA wave per SIMD", out);
    run<3>("d: one wave per SIMD, interleaved stream", out);
    run<4>("s: one wave per SIMD, MFMAs then FMAs", out);
    run64<0>("fp64 m: MFMA waves alone", out);
    run64<1>("fp64 v: FMA waves alone", out);
    run64<2>("fp64 c: one MFMA wave + one FMA wave per SIMD", out);
    return 0;
}
