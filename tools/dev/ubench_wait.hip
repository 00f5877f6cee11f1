// dev: what a host round trip costs -- a small kernel that writes into pinned host memory is launched, the host waits for it and launches the next:
// (A) polling hipStreamQuery (what Engine::sync_point and the cohort's fiber wait do), (B) polling a stamp the kernel writes behind its data
// (what the contraction kernels' notification does).  hipcc --offload-arch=gfx950 tools/dev/ubench_wait.hip -o /tmp/ubench_wait
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_copy(const int *src, int *dst, int n, volatile unsigned *stamp, unsigned seq, unsigned *counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
    if (!stamp) return;
    __threadfence_system();
    __shared__ bool last;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) { *counter = 0; __threadfence_system(); *stamp = seq; }
}
int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int *d, *h; unsigned *hs, *dc;
    hipMalloc(&d, 1 << 20); hipHostMalloc(&h, 1 << 20, hipHostMallocMapped); hipHostMalloc(&hs, 64, hipHostMallocMapped); hipMalloc(&dc, 4); hipMemset(dc, 0, 4);
    *hs = 0;
    for (int blocks : {1, 6, 69}) {
        const int n = blocks * 256, N = 300;
        for (int mode = 0; mode < 2; ++mode) {
            double tot = 0;
            for (int it = 0; it < N + 20; ++it) {
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, d, h, n, mode ? hs : nullptr, (unsigned)(it + 1 + 1000 * blocks), dc);
                if (mode == 0) { while (hipStreamQuery(st) == hipErrorNotReady) __builtin_ia32_pause(); }
                else { while (*(volatile unsigned *)hs != (unsigned)(it + 1 + 1000 * blocks)) __builtin_ia32_pause(); }
                const auto t1 = std::chrono::steady_clock::now();
                if (it >= 20) tot += std::chrono::duration<double>(t1 - t0).count();
            }
            hipStreamSynchronize(st);
            std::printf("blocks %3d  %s: %.1f us per launch + wait\n", blocks, mode ? "stamp in pinned memory" : "hipStreamQuery polling ", tot / N * 1e6);
        }
    }
    return 0;
}
