// Dispatch ramp of a grid that fills the chip once (MI355X: 256 CUs x 16 waves): how long after the first wave does
// the last wave of the launch start, as a function of workgroup size, LDS per workgroup and registers per wave?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/launch_ramp scripts/launch_ramp.hip && scripts/launch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS, int LDS_BYTES, int VGPRS>
__global__ __launch_bounds__(THREADS) void ramp_kernel(unsigned long long* entry, unsigned long long* exit_, float* sink, int spin) {
    __shared__ float lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) acc[i] = threadIdx.x * 0.5f + i;
    for (int k = 0; k < spin; ++k) {
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) acc[i] = fmaf(acc[i], 1.0001f, 0.25f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) s += acc[i];
    if (LDS_BYTES > 0) { lds[threadIdx.x] = s; __syncthreads(); s = lds[(threadIdx.x + 1) % THREADS]; }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * THREADS + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { entry[wave] = t0; exit_[wave] = t1; }
    if (s == 12345.678f) sink[0] = s;
}

template <int THREADS, int LDS_BYTES, int VGPRS>
void run(const char* name, int spin) {
    const int waves = 4096, blocks = waves * 64 / THREADS;
    unsigned long long *e, *x; float* sink;
    hipMalloc(&e, waves * 8); hipMalloc(&x, waves * 8); hipMalloc(&sink, 4);
    std::vector<unsigned long long> he(waves), hx(waves);
    double skew50 = 0, skew100 = 0, span = 0, life = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipLaunchKernelGGL((ramp_kernel<THREADS, LDS_BYTES, VGPRS>), dim3(blocks), dim3(THREADS), 0, 0, e, x, sink, spin);
        hipDeviceSynchronize();
        if (r < 3) continue;
        hipMemcpy(he.data(), e, waves * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), x, waves * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(he.begin(), he.end());
        std::vector<double> sk(waves);
        double l = 0;
        for (int i = 0; i < waves; ++i) { sk[i] = (he[i] - t0) / 100.0; l += (hx[i] - he[i]) / 100.0; }
        std::sort(sk.begin(), sk.end());
        skew50 += sk[waves / 2]; skew100 += sk[waves - 1];
        span += (*std::max_element(hx.begin(), hx.end()) - t0) / 100.0;
        life += l / waves;
    }
    printf("%-44s blocks %4d  entry skew p50 %5.2f us  max %5.2f us   mean wave life %5.2f us  first entry -> last exit %5.2f us\n",
           name, blocks, skew50 / reps, skew100 / reps, life / reps, span / reps);
    hipFree(e); hipFree(x); hipFree(sink);
}

int main() {
    const int spin = 200;
    run<256, 0, 16>("256 thr, no LDS, few VGPR", spin);
    run<256, 37888, 16>("256 thr, 37.9 KB LDS, few VGPR", spin);
    run<256, 37888, 96>("256 thr, 37.9 KB LDS, ~100 VGPR", spin / 6);
    run<256, 0, 96>("256 thr, no LDS, ~100 VGPR", spin / 6);
    run<512, 58368, 96>("512 thr, 57 KB LDS, ~100 VGPR", spin / 6);
    run<1024, 99328, 96>("1024 thr, 97 KB LDS, ~100 VGPR", spin / 6);
    run<64, 0, 96>("64 thr, no LDS, ~100 VGPR", spin / 6);
    return 0;
}
