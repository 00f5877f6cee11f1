#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *cyc, int n)
{
    const int lane = threadIdx.x;
    double x0 = 1.0 + lane * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { x0 = fma(x0, 1.0000001, 1e-9); }
    long long t1 = clock64();
    for (int i = 0; i < n; ++i) { x0 = fma(x0, 1.0000001, 1e-9); x1 = fma(x1, 1.0000001, 1e-9); }
    long long t2 = clock64();
    for (int i = 0; i < n; ++i) { x0 = fma(x0, 1.0000001, 1e-9); x1 = fma(x1, 1.0000001, 1e-9); x2 = fma(x2, 1.0000001, 1e-9); x3 = fma(x3, 1.0000001, 1e-9); }
    long long t3 = clock64();
    for (int i = 0; i < n; ++i) { x0 = fma(x0, 1.0000001, 1e-9); x1 = fma(x1, 1.0000001, 1e-9); x2 = fma(x2, 1.0000001, 1e-9); x3 = fma(x3, 1.0000001, 1e-9);
                                  x4 = fma(x4, 1.0000001, 1e-9); x5 = fma(x5, 1.0000001, 1e-9); x6 = fma(x6, 1.0000001, 1e-9); x7 = fma(x7, 1.0000001, 1e-9); }
    long long t4 = clock64();
    float f0 = 1.0f + lane * 1e-6f, f1 = f0 + 1;
    for (int i = 0; i < n; ++i) { f0 = fmaf(f0, 1.0000001f, 1e-9f); }
    long long t5 = clock64();
    for (int i = 0; i < n; ++i) { x0 = x0 + 1e-9; x1 = x1 + 1e-9; x2 = x2 + 1e-9; x3 = x3 + 1e-9; }
    long long t6 = clock64();
    out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1;
    if (lane == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 64 * 8 * 4096); hipMalloc(&cyc, 64);
    const int n = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize();
        long long c[8]; hipMemcpy(c, cyc, 48, hipMemcpyDeviceToHost);
        printf("waves/WG %d: cycles per iteration: 1 chain %.1f, 2 chains %.1f, 4 chains %.1f, 8 chains %.1f; fp32 1 chain %.1f; 4 f64 adds %.1f\n", waves, (double)c[0] / n, (double)c[1] / n, (double)c[2] / n, (double)c[3] / n, (double)c[4] / n, (double)c[5] / n);
    }
    return 0;
}
