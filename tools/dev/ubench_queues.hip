// dev: what a small kernel costs when N streams each run a dependent chain of them (launch + dispatch + completion, per kernel),
// and the same with every stream's kernel spinning ~20 us on ONE workgroup.  hipcc -O2 --offload-arch=gfx950 ubench_queues.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_spin(long long cycles) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} }
int main()
{
    for (int mode = 0; mode < 2; ++mode)
        for (int N : {1, 2, 4, 8, 16, 32}) {
            std::vector<hipStream_t> st(N);
            for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            const int K = 400;
            auto work = [&](int i) { for (int k = 0; k < K; ++k) { if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[i], (int *)nullptr); else hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 48000LL); } hipStreamSynchronize(st[i]); };
            for (int i = 0; i < N; ++i) work(i);     // warm
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int i = 0; i < N; ++i) th.emplace_back(work, i);
            for (auto &t : th) t.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%s N=%2d streams: %.1f us per kernel per stream (%.2f M kernels/s total)\n", mode ? "spin20us" : "empty   ", N, dt / K * 1e6, N * K / dt / 1e6);
            for (auto &s : st) hipStreamDestroy(s);
        }
    return 0;
}
