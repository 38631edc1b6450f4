// Micro-benchmark (round 4): issue cost of the PACKED fp32 instructions of gfx950 -- v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32, two fp32 operations per lane and instruction on 64-bit register pairs -- next to their
// scalar forms, same protocol as scripts/microbench_rates.hip (eight independent chains per wave, four waves per SIMD,
// inline asm so that the instruction is what is named).  Question: does a packed pair issue in less than two scalar
// instructions when its three sources are distinct VGPR pairs (the scalar v_fma_f32 with three distinct VGPR sources
// takes 4 cycles)?
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_pk.hip -o /tmp/mbpk && /tmp/mbpk
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAIN8(OP)                                                                                       \
    asm volatile(OP(%0) "\n" OP(%1) "\n" OP(%2) "\n" OP(%3) "\n" OP(%4) "\n" OP(%5) "\n" OP(%6) "\n" OP(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                 : "v"(k0), "v"(k1), "s"(s0))

#define OP_PKFMA3(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9"
#define OP_PKFMA2(x) "v_pk_fma_f32 " #x ", " #x ", " #x ", %8"
#define OP_PKFMAS(x) "v_pk_fma_f32 " #x ", " #x ", %8, %10"
#define OP_PKMUL(x) "v_pk_mul_f32 " #x ", " #x ", %8"
#define OP_PKADD(x) "v_pk_add_f32 " #x ", " #x ", %8"
#define OP_PKMULS(x) "v_pk_mul_f32 " #x ", " #x ", %10"
#define OP_PKFMA_OPSEL(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9 op_sel:[0,1,0] op_sel_hi:[1,0,1]"
#define OP_PKMOV(x) "v_pk_mov_b32 " #x ", " #x ", %8 op_sel:[1,0]"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const f2 k0 = {0.999f, 0.998f}, k1 = {0.5f, 0.25f};
    f2 s0 = {0.997f, 0.996f};
    s0.x = __builtin_amdgcn_readfirstlane(s0.x);
    s0.y = __builtin_amdgcn_readfirstlane(s0.y);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) CHAIN8(OP_PKFMA3);
            if (MODE == 1) CHAIN8(OP_PKFMA2);
            if (MODE == 2) CHAIN8(OP_PKFMAS);
            if (MODE == 3) CHAIN8(OP_PKMUL);
            if (MODE == 4) CHAIN8(OP_PKADD);
            if (MODE == 5) CHAIN8(OP_PKMULS);
            if (MODE == 6) CHAIN8(OP_PKFMA_OPSEL);
            if (MODE == 7) CHAIN8(OP_PKMOV);
        }
    }
    const f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 2000, wps = 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, 50, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-40s %.3f ms  %.2f cycles @2.4GHz per wave-instruction (= 2 fp32 ops per lane)\n", name, best,
           best * 1e-3 * 2.4e9 / (wps * (double)iters * 64.0));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    run<0>("v_pk_fma_f32 x,x,K0,K1 (3 VGPR pairs)", out);
    run<1>("v_pk_fma_f32 x,x,x,K0 (2 VGPR pairs)", out);
    run<2>("v_pk_fma_f32 x,x,K0,s[..] (SGPR pair)", out);
    run<6>("v_pk_fma_f32 3 pairs + op_sel swizzle", out);
    run<3>("v_pk_mul_f32 x,x,K0", out);
    run<5>("v_pk_mul_f32 x,x,s[..]", out);
    run<4>("v_pk_add_f32 x,x,K0", out);
    run<7>("v_pk_mov_b32 x,x,K0 op_sel", out);
    return 0;
}
