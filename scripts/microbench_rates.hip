// Micro-benchmark: issue cost (cycles per wave-instruction at 4 waves per SIMD) of the VALU instructions the collision walk
// is made of, on gfx950.  Eight independent dependency chains per wave, inline asm so that the instruction is what is named.
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_rates.hip -o /tmp/mbr && /tmp/mbr
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAIN8(OP)                                                                                       \
    asm volatile(OP(%0) "\n" OP(%1) "\n" OP(%2) "\n" OP(%3) "\n" OP(%4) "\n" OP(%5) "\n" OP(%6) "\n" OP(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                 : "v"(k0), "v"(k1))

#define OP_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9"
#define OP_FLOOR(x) "v_floor_f32 " #x ", " #x
#define OP_MED3(x) "v_med3_f32 " #x ", " #x ", %8, %9"
#define OP_CVTFLR(x) "v_cvt_flr_i32_f32 " #x ", " #x
#define OP_CVTU(x) "v_cvt_u32_f32 " #x ", " #x
#define OP_CVTI(x) "v_cvt_i32_f32 " #x ", " #x
#define OP_MADU24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9"
#define OP_MINU(x) "v_min_u32 " #x ", " #x ", %8"
#define OP_SQRT(x) "v_sqrt_f32 " #x ", " #x
#define OP_MADU64(x) "v_mul_lo_u32 " #x ", " #x ", %8"
#define OP_CNDMASK(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc"
#define OP_BFE(x) "v_bfe_u32 " #x ", " #x ", 8, 8"
#define OP_SUB(x) "v_sub_f32 " #x ", " #x ", %8"
#define OP_MIN(x) "v_min_f32 " #x ", " #x ", %8"
#define OP_SIN(x) "v_sin_f32 " #x ", " #x
#define OP_MUL(x) "v_mul_f32 " #x ", " #x ", %8"
#define OP_ADD(x) "v_add_f32 " #x ", " #x ", %8"
#define OP_MAX(x) "v_max_f32 " #x ", " #x ", %8"
#define OP_FMAC(x) "v_fmac_f32 " #x ", %8, %9"
#define OP_FMA_SGPR(x) "v_fma_f32 " #x ", " #x ", %8, s4"
#define OP_FMA_2(x) "v_fma_f32 " #x ", " #x ", " #x ", %8"
#define OP_FMAMK(x) "v_fmamk_f32 " #x ", " #x ", 0x3f7fbe77, %8"
#define OP_ADDU(x) "v_add_u32 " #x ", " #x ", %8"
#define OP_AND(x) "v_and_b32 " #x ", " #x ", %8"
#define OP_XOR(x) "v_xor_b32 " #x ", " #x ", %8"
#define OP_LSHL(x) "v_lshlrev_b32 " #x ", 1, " #x
#define OP_CNDMASK_S(x) "v_cndmask_b32 " #x ", " #x ", %8, s[6:7]"
#define OP_MADU64R(x) "v_mad_u64_u32 v[40:41], s[6:7], " #x ", %8, v[40:41]"
#define OP_PKFMA(x) "v_mov_b32 " #x ", " #x
#define OP_MULHI(x) "v_mul_hi_u32 " #x ", " #x ", %8"
#define OP_MAD64(x) "v_mad_u64_u32 v[50:51], s[6:7], " #x ", %8, v[50:51]"
#define OP_LOG(x) "v_log_f32 " #x ", " #x
#define OP_EXP(x) "v_exp_f32 " #x ", " #x
#define OP_RCP(x) "v_rcp_f32 " #x ", " #x
#define OP_CMP(x) "v_cmp_lt_f32 vcc, " #x ", %8"
#define OP_READLANE(x) "v_mov_b32_dpp " #x ", " #x " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float k0 = 0.999f, k1 = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) CHAIN8(OP_FMA);
            if (MODE == 1) CHAIN8(OP_FLOOR);
            if (MODE == 2) CHAIN8(OP_MED3);
            if (MODE == 3) CHAIN8(OP_CVTFLR);
            if (MODE == 4) CHAIN8(OP_CVTU);
            if (MODE == 5) CHAIN8(OP_CVTI);
            if (MODE == 6) CHAIN8(OP_MADU24);
            if (MODE == 7) CHAIN8(OP_MINU);
            if (MODE == 8) CHAIN8(OP_SQRT);
            if (MODE == 9) CHAIN8(OP_MADU64);
            if (MODE == 10) CHAIN8(OP_CNDMASK);
            if (MODE == 11) CHAIN8(OP_BFE);
            if (MODE == 12) CHAIN8(OP_SUB);
            if (MODE == 13) CHAIN8(OP_MIN);
            if (MODE == 14) CHAIN8(OP_SIN);
            if (MODE == 15) CHAIN8(OP_READLANE);
            if (MODE == 16) CHAIN8(OP_MUL);
            if (MODE == 17) CHAIN8(OP_ADD);
            if (MODE == 18) CHAIN8(OP_MAX);
            if (MODE == 19) CHAIN8(OP_FMAC);
            if (MODE == 20) CHAIN8(OP_FMA_SGPR);
            if (MODE == 21) CHAIN8(OP_FMA_2);
            if (MODE == 22) CHAIN8(OP_FMAMK);
            if (MODE == 23) CHAIN8(OP_ADDU);
            if (MODE == 24) CHAIN8(OP_AND);
            if (MODE == 25) CHAIN8(OP_XOR);
            if (MODE == 26) CHAIN8(OP_LSHL);
            if (MODE == 27) CHAIN8(OP_CNDMASK_S);
            if (MODE == 28) CHAIN8(OP_PKFMA);
            if (MODE == 29) CHAIN8(OP_MULHI);
            if (MODE == 30) CHAIN8(OP_MAD64);
            if (MODE == 31) CHAIN8(OP_LOG);
            if (MODE == 32) CHAIN8(OP_EXP);
            if (MODE == 33) CHAIN8(OP_RCP);
            if (MODE == 34) CHAIN8(OP_CMP);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
    // per SIMD: wps waves x iters x 64 instructions
    printf("%-22s %.3f ms  %.2f cycles @2.4GHz per wave-instruction\n", name, best, best * 1e-3 * 2.4e9 / (wps * (double)iters * 64.0));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_f32 v,v,v,v", out); run<12>("v_sub_f32", out); run<13>("v_min_f32", out); run<1>("v_floor_f32", out); run<2>("v_med3_f32", out);
    run<3>("v_cvt_flr_i32_f32", out); run<5>("v_cvt_i32_f32", out); run<4>("v_cvt_u32_f32", out); run<6>("v_mad_u32_u24", out);
    run<7>("v_min_u32", out); run<11>("v_bfe_u32", out); run<10>("v_cndmask_b32", out); run<9>("v_mul_lo_u32", out);
    run<16>("v_mul_f32", out); run<17>("v_add_f32", out); run<18>("v_max_f32", out); run<19>("v_fmac_f32 (VOP2)", out);
    run<20>("v_fma_f32 v,v,v,s", out); run<21>("v_fma_f32 x,x,x,v", out); run<22>("v_fmamk_f32", out); run<23>("v_add_u32", out);
    run<24>("v_and_b32", out); run<25>("v_xor_b32", out); run<26>("v_lshlrev_b32", out); run<27>("v_cndmask_b32 sgpr", out);
    run<28>("v_mov_b32", out);
    run<29>("v_mul_hi_u32", out); run<30>("v_mad_u64_u32 (one acc)", out); run<34>("v_cmp_lt_f32", out);
    run<31>("v_log_f32", out); run<32>("v_exp_f32", out); run<33>("v_rcp_f32", out);
    run<8>("v_sqrt_f32", out); run<14>("v_sin_f32", out); run<15>("v_mov_b32_dpp", out);
    return 0;
}
