// tools/ubench.hip -- developer micro-benchmarks (single wave latencies on MI355X); not part of the product
// (round 6: sixteen operations to a loop iteration.  With ONE -- rounds 2-5 -- the loop's own counter, compare and branch were ~24 of the
//  "32 cycles" a dependent fma was quoted at; tools/dev/ubench_fp64.hip: a wavefront issues a vector fp64 operation every 5-6 cycles,
//  dependent on the one before or not)
#define R16(x) x x x x x x x x x x x x x x x x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_lat(double *out, long long *cyc, int n, const double *gmem)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (double)((i * 37 + 11) & 1023);
    __syncthreads();
    double x = 1.0 + lane * 1e-9;
    long long t0w = wall_clock64(), t0 = clock64();
    for (int i = 0; i < n / 16; ++i) { R16(x = fma(x, 1.0000001, 1e-9);) }
    long long t1 = clock64(), t1w = wall_clock64();
    double y = 0.5 + lane * 1e-9;
    for (int i = 0; i < n / 16; ++i) { R16(y = exp(-y * 0.5);) }
    long long t2 = clock64();
    double z = 2.0 + lane * 1e-9;
    for (int i = 0; i < n / 16; ++i) { R16(z = log(z + 3.0);) }
    long long t3 = clock64();
    int idx = lane;
    for (int i = 0; i < n / 16; ++i) { R16(idx = (int)lds[idx & 1023];) }
    long long t4 = clock64();
    int g = lane;
    for (int i = 0; i < n / 8; ++i) g = (int)gmem[g & 4095];
    long long t5 = clock64();
    double w = 1.0 + lane;
    for (int i = 0; i < n / 16; ++i) { R16(w = w / 1.0000001;) }
    long long t6 = clock64();
    out[lane] = x + y + z + idx + g + w;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t1w - t0w; cyc[6] = t6 - t5; }
}
int main()
{
    double *out, *gm; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64); hipMalloc(&gm, 4096 * 8);
    double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (double)((i * 613 + 7) & 4095);
    hipMemcpy(gm, h, sizeof h, hipMemcpyHostToDevice);
    const int n = 16000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, n, gm); hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c[8]; hipMemcpy(c, cyc, 56, hipMemcpyDeviceToHost);
        double tot = (double)(c[0] + c[1] + c[2] + c[3] + c[4] + c[6]);
        std::printf("kernel %.3f ms; shader cycles total %.0f => %.1f MHz effective; wall_clock ticks for fma loop %lld (100MHz => %.1f us) vs %lld cycles => %.0f MHz\n",
                    ms, tot, tot / (ms * 1e3), c[5], c[5] / 100.0, c[0], c[0] / (c[5] / 100.0));
        std::printf("  per-iteration cycles: fma %.1f  exp %.1f  log %.1f  lds-chase %.1f  global-chase %.1f  div %.1f\n",
                    (double)c[0] / n, (double)c[1] / n, (double)c[2] / n, (double)c[3] / n, (double)c[4] / (n / 8), (double)c[6] / n);
    }
    return 0;
}
