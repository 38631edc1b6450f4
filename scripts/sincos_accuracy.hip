// Accuracy against fp64 sin / cos over |x| <= 20 rad of: the hardware v_sin_f32 / v_cos_f32 (input in revolutions); the
// product's fast_sincos (csrc/mpb_geom.h: round 4, reduction by pi, one xor for the sign); the pi/2 form it replaced.
//   hipcc --offload-arch=gfx950 -O3 -Imotion_planning_baselines_amd/csrc scripts/sincos_accuracy.hip -o /tmp/sc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
char* mpb_err_buf() { static char b[512]; return b; }
#include "../motion_planning_baselines_amd/csrc/mpb_geom.h"
__device__ __forceinline__ void sincos_pi2_form(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.6366197466850281f);
    float r = fmaf(-k, 1.5707963705062866f, x);
    r = fmaf(-k, -4.371138828673793e-08f, r);
    r = fmaf(-k, -1.7763568394002505e-15f, r);
    const float r2 = r * r;
    const float s = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float c = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                         fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float a = (q & 1) ? c : s;
    const float b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
__global__ void k(double* err, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = -20.0f + 40.0f * (float)i / (float)n;
    const double sd = sin((double)x), cd = cos((double)x);
    // hardware: revolutions; v_fract keeps the argument in [0,1)
    const float rev = x * 0.15915494309189535f;
    const float fr = rev - floorf(rev);
    const float hs = __builtin_amdgcn_sinf(fr), hc = __builtin_amdgcn_cosf(fr);
    float ps, pc, os, oc;
    fast_sincos(x, ps, pc);
    sincos_pi2_form(x, os, oc);
    err[6 * i + 0] = fabs((double)hs - sd);
    err[6 * i + 1] = fabs((double)hc - cd);
    err[6 * i + 2] = fabs((double)ps - sd);
    err[6 * i + 3] = fabs((double)pc - cd);
    err[6 * i + 4] = fabs((double)os - sd);
    err[6 * i + 5] = fabs((double)oc - cd);
}
int main() {
    const int n = 1 << 22;
    double* d; (void)hipMalloc(&d, sizeof(double) * 6 * n);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double* h = (double*)malloc(sizeof(double) * 6 * n);
    (void)hipMemcpy(h, d, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int c = 0; c < 6; ++c) if (h[6 * i + c] > m[c]) m[c] = h[6 * i + c];
    printf("max abs error over [-20, 20] rad: v_sin %.3e  v_cos %.3e   fast_sincos (pi form) sin %.3e cos %.3e   pi/2 form sin %.3e cos %.3e\n",
           m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
