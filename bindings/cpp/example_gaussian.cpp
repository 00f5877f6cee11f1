// A C++ driver in the style of the reference's src/drivers/polychord_CC.cpp: Settings + run_polychord, first with the
// likelihood fused into the sampling kernel (library functions as callbacks), then with a likelihood written here.
//   g++ -std=c++17 -I include bindings/cpp/example_gaussian.cpp -L polychordlite_amd -lpolychord_hip -Wl,-rpath,$PWD/polychordlite_amd
#include <cmath>
#include <cstdio>
#include "polychord_hip.hpp"

static long g_calls = 0;
static double my_gaussian(double *theta, int nDims, double *phi, int nDerived)
{   // likelihoods/examples/gaussian.f90: N(0.5, 0.1^2 I), phi_1 = radius
    ++g_calls;
    double r2 = 0.0;
    for (int i = 0; i < nDims; ++i) r2 += (theta[i] - 0.5) * (theta[i] - 0.5);
    if (nDerived > 0) phi[0] = std::sqrt(r2);
    return -nDims * (std::log(0.1) + 0.5 * std::log(2 * M_PI)) - 0.5 * r2 / 0.01;
}
static int g_dumps = 0;
static void my_dumper(int ndead, int nlive, int npars, double *, double *, double *, double logZ, double logZerr)
{
    ++g_dumps;
    if (nlive == 0) std::printf("final dump: ndead %d npars %d logZ %.4f +/- %.4f\n", ndead, npars, logZ, logZerr);
}

int main(int argc, char **argv)
{
    Settings s(6, 1);
    s.nlive = 120; s.num_repeats = 12; s.seed = 3; s.feedback = 0; s.write_stats = true; s.write_prior = false;
    s.base_dir = argc > 1 ? argv[1] : "chains"; s.file_root = "cpp_device";
    polychord_hip_set_gaussian(0.5, 0.1);
    run_polychord(polychord_hip_gaussian, polychord_hip_uniform_prior, my_dumper, s);     // fused on the GPU
    const int dumps_device = g_dumps;
    s.file_root = "cpp_host";
    run_polychord(my_gaussian, my_dumper, s);                                               // host callback, default prior
    std::printf("dumps %d %d host likelihood calls %ld\n", dumps_device, g_dumps - dumps_device, g_calls);
    return (g_calls > 1000 && dumps_device > 2) ? 0 : 1;
}
