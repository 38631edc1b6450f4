// Accuracy of v_rcp_f64 on gfx950 with 0, 1 and 2 Newton steps (the pivot-block reciprocal of the GPMP2 solve), against 1/x in
// long double on the host: maximum and mean relative error in units of 2^-53 over 2^20 random inputs spanning 1e-12 .. 1e12.
// hipcc --offload-arch=gfx950 -O3 scripts/rcp_accuracy.hip -o /tmp/rcpacc && /tmp/rcpacc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    double *hx = (double*)malloc(n * 8), *h[3];
    srand(7);
    for (int i = 0; i < n; ++i) {
        double m = 1.0 + rand() / (double)RAND_MAX, e = -40 + 80.0 * rand() / (double)RAND_MAX;
        hx[i] = ldexp(m, (int)e) * ((rand() & 1) ? 1 : -1);
    }
    double *dx, *d[3];
    hipMalloc(&dx, n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    for (int j = 0; j < 3; ++j) { hipMalloc(&d[j], n * 8); h[j] = (double*)malloc(n * 8); }
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], n);
    for (int j = 0; j < 3; ++j) hipMemcpy(h[j], d[j], n * 8, hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; ++j) {
        long double mx = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            long double ex = 1.0L / (long double)hx[i];
            long double rel = fabsl(((long double)h[j][i] - ex) / ex) / ldexpl(1.0L, -53);
            if (rel > mx) mx = rel;
            sum += rel;
        }
        printf("v_rcp_f64 + %d Newton step(s): max relative error %.3Lf x 2^-53, mean %.3Lf x 2^-53\n", j, mx, sum / n);
    }
    return 0;
}
