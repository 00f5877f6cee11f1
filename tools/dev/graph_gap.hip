// developer micro-benchmark: time per dependent kernel boundary, stream launches against a captured hipGraph
// build: hipcc -O2 --offload-arch=gfx950 tools/dev/graph_gap.hip -o tools/dev/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_busy(double *x, int iters) { double v = x[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9; x[threadIdx.x] = v; }
int main()
{
    double *d; hipMalloc(&d, 8 * 1024); hipMemset(d, 0, 8 * 1024);
    hipStream_t st; hipStreamCreate(&st);
    const int NK = 3, REP = 2000;
    for (int iters : {100, 20000}) {
        // stream
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, st, d, iters);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, st, d, iters);
        hipStreamSynchronize(st);
        double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // graph of NK kernels
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, st, d, iters);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 20; ++w) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        double tg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // one kernel alone (duration)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st); hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, st, d, iters); hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("iters %d: kernel ~%.1f us; per kernel: stream %.2f us, graph %.2f us\n", iters, ms * 1e3, ts / (REP * NK) * 1e6, tg / (REP * NK) * 1e6);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
