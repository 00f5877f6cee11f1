#include <chrono>
#include <cmath>
#include <cstdio>
#include "polychord_hip.hpp"
static long g_calls = 0;
static double my_gaussian(double *theta, int nDims, double *phi, int nDerived)
{
    ++g_calls;
    double r2 = 0.0;
    for (int i = 0; i < nDims; ++i) r2 += (theta[i] - 0.5) * (theta[i] - 0.5);
    if (nDerived > 0) phi[0] = std::sqrt(r2);
    return -nDims * (std::log(0.1) + 0.5 * std::log(2 * M_PI)) - 0.5 * r2 / 0.01;
}
int main(int argc, char **argv)
{
    Settings s(20, 1);
    s.nlive = 500; s.num_repeats = 40; s.seed = 3; s.feedback = 0; s.write_stats = false; s.write_prior = false; s.maximise = false;
    s.base_dir = "/tmp";
    for (int rep = 0; rep < 2; ++rep) {
        g_calls = 0;
        auto t0 = std::chrono::steady_clock::now();
        run_polychord(my_gaussian, s);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("host-callback run: %ld calls in %.3f s = %.2f M calls/s\n", g_calls, dt, g_calls / dt / 1e6);
    }
    return 0;
}
