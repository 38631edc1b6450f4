// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against fp64 sin / cos, and of the polynomial
// fast_sincos used by the FK code, over |x| <= 8 rad.   hipcc --offload-arch=gfx950 -O3 scripts/sincos_accuracy.hip -o /tmp/sc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cs) {
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
    const float x = -8.0f + 16.0f * (float)i / (float)n;
    const double sd = sin((double)x), cd = cos((double)x);
    // hardware: revolutions; v_fract keeps the argument in [0,1)
    const float rev = x * 0.15915494309189535f;
    const float fr = rev - floorf(rev);
    const float hs = __builtin_amdgcn_sinf(fr), hc = __builtin_amdgcn_cosf(fr);
    float ps, pc;
    fast_sincos(x, ps, pc);
    err[4 * i + 0] = fabs((double)hs - sd);
    err[4 * i + 1] = fabs((double)hc - cd);
    err[4 * i + 2] = fabs((double)ps - sd);
    err[4 * i + 3] = fabs((double)pc - cd);
}
int main() {
    const int n = 1 << 22;
    double* d; hipMalloc(&d, sizeof(double) * 4 * n);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double* h = (double*)malloc(sizeof(double) * 4 * n);
    hipMemcpy(h, d, sizeof(double) * 4 * n, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int c = 0; c < 4; ++c) if (h[4 * i + c] > m[c]) m[c] = h[4 * i + c];
    printf("max abs error over [-8, 8] rad: v_sin %.3e  v_cos %.3e   polynomial sin %.3e cos %.3e\n", m[0], m[1], m[2], m[3]);
    return 0;
}
