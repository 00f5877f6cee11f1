// tools/dev/ubench_fp64.hip -- developer micro-benchmark: fp64 latency and issue rate of ONE wavefront on MI355X, sixteen operations to a loop
// iteration (tools/ubench.hip's "fma 32 cycles" is one operation plus ~24 cycles of loop: not a latency).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
__global__ void k(double *out, long long *cyc, int n)
{
    const int lane = threadIdx.x;
    double x0 = 1.0 + lane * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double c = 1.0000001, d = 1e-9;
    long long t[12]; int q = 0;
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = fma(x0, c, d);) }                                        // dependent fma
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = x0 + d;) }                                               // dependent add
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = x0 * c;) }                                               // dependent mul
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = fma(x0, c, d); x1 = fma(x1, c, d);) }                    // 2 chains
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = fma(x0, c, d); x1 = fma(x1, c, d); x2 = fma(x2, c, d); x3 = fma(x3, c, d);) }      // 4 chains
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = fma(x0, c, d); x1 = fma(x1, c, d); x2 = fma(x2, c, d); x3 = fma(x3, c, d); x4 = fma(x4, c, d); x5 = fma(x5, c, d); x6 = fma(x6, c, d); x7 = fma(x7, c, d);) }
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = x0 + d; x1 = x1 + d; x2 = x2 + d; x3 = x3 + d;) }        // 4 independent adds
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = (x0 < x1) ? x0 + d : x1;) }                              // compare + select chain
    t[q++] = clock64();
    for (int i = 0; i < n; ++i) { R16(x0 = x0 + __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x0), 3), __builtin_amdgcn_readlane(__double2loint(x0), 3));) }   // readlane + add chain
    t[q++] = clock64();
    out[blockIdx.x * blockDim.x + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (lane == 0 && blockIdx.x == 0) for (int i = 0; i + 1 < q; ++i) cyc[i] = t[i + 1] - t[i];
}
int main()
{
    double *out; long long *cyc; (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 128);
    const int n = 4000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); (void)hipDeviceSynchronize();
        long long c[12]; (void)hipMemcpy(c, cyc, 96, hipMemcpyDeviceToHost);
        const double u = 16.0 * n;
        printf("waves/CU %d: cycles per operation (16 to an iteration): dependent fma %.1f add %.1f mul %.1f; per fma with 2 chains %.1f, 4 chains %.1f, 8 chains %.1f; per add with 4 chains %.1f; compare+add+select chain %.1f; 2 readlanes + add chain %.1f\n",
               waves, c[0] / u, c[1] / u, c[2] / u, c[3] / u / 2, c[4] / u / 4, c[5] / u / 8, c[6] / u / 4, c[7] / u, c[8] / u);
    }
    return 0;
}
