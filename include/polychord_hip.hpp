// polychord_hip.hpp -- the C++ face of libpolychord_hip.so: `struct Settings` and the `run_polychord` overloads a C++ caller of
// PolyChordLite uses (reference src/polychord/interfaces.hpp:8-87, defaults of src/polychord/c_interface.cpp:6-39), header only,
// over the C entry points of polychord_hip.h.  A driver written against the reference's interfaces.hpp compiles against this
// header unchanged:
//
//     #include "polychord_hip.hpp"
//     double loglike(double *theta, int nDims, double *phi, int nDerived);
//     int main() { Settings s(20, 2); s.nlive = 2000; s.num_repeats = 40; s.write_stats = true; run_polychord(loglike, s); }
//
// Passing polychord_hip_gaussian / _rastrigin / _twin_gaussian / _corr_gaussian (and polychord_hip_uniform_prior) as the
// callbacks makes the engine evaluate them inside the sampling kernel.  There is no MPI in this engine: the MPI_Comm
// overloads of the reference have no counterpart; one process drives one GPU.
#pragma once
#include <string>
#include <vector>
#include "polychord_hip.h"

struct Settings {
    int nDims, nDerived;
    int nlive = 500, num_repeats, nprior = -1, nfail = -1;
    bool do_clustering = false;
    int feedback = 1;
    double precision_criterion = 0.001, logzero = -1e30;
    int max_ndead = -1;
    double boost_posterior = 0.0;
    bool posteriors = false, equals = false, cluster_posteriors = false, write_resume = false, write_paramnames = false,
         read_resume = false, write_stats = false, write_live = false, write_dead = false, write_prior = true, maximise = true;
    double compression_factor = 0.36787944117144233;
    bool synchronous = true;
    std::string base_dir = "chains", file_root = "test";
    std::vector<double> grade_frac;
    std::vector<int> grade_dims;
    std::vector<double> loglikes;
    std::vector<int> nlives;
    int seed = -1;
    Settings(int _nDims = 0, int _nDerived = 0)
        : nDims(_nDims), nDerived(_nDerived), num_repeats(_nDims * 5), grade_frac{1.0}, grade_dims{_nDims} {}
};

inline double default_loglikelihood(double *, int, double *, int) { return 0.0; }
inline void default_prior(double *cube, double *theta, int nDims) { for (int i = 0; i < nDims; ++i) theta[i] = cube[i]; }
inline void default_dumper(int, int, int, double *, double *, double *, double, double) {}

inline void run_polychord(double (*loglikelihood)(double *, int, double *, int), void (*prior)(double *, double *, int),
                          void (*dumper)(int, int, int, double *, double *, double *, double, double), Settings s)
{
    int comm = 0;
    polychord_c_interface(loglikelihood, prior, dumper, s.nlive, s.num_repeats, s.nprior, s.nfail, s.do_clustering, s.feedback,
                          s.precision_criterion, s.logzero, s.max_ndead, s.boost_posterior, s.posteriors, s.equals,
                          s.cluster_posteriors, s.write_resume, s.write_paramnames, s.read_resume, s.write_stats, s.write_live,
                          s.write_dead, s.write_prior, s.maximise, s.compression_factor, s.synchronous, s.nDims, s.nDerived,
                          const_cast<char *>(s.base_dir.c_str()), const_cast<char *>(s.file_root.c_str()), (int)s.grade_frac.size(),
                          s.grade_frac.data(), s.grade_dims.data(), (int)s.loglikes.size(), s.loglikes.data(), s.nlives.data(),
                          s.seed, &comm);
}
inline void run_polychord(double (*loglikelihood)(double *, int, double *, int),
                          void (*dumper)(int, int, int, double *, double *, double *, double, double), Settings s)
{
    run_polychord(loglikelihood, default_prior, dumper, s);
}
inline void run_polychord(double (*loglikelihood)(double *, int, double *, int), void (*prior)(double *, double *, int), Settings s)
{
    run_polychord(loglikelihood, prior, default_dumper, s);
}
inline void run_polychord(double (*loglikelihood)(double *, int, double *, int), Settings s)
{
    run_polychord(loglikelihood, default_prior, default_dumper, s);
}
inline void run_polychord(double (*loglikelihood)(double *, int, double *, int), void (*setup_loglikelihood)(), std::string inifile)
{
    int comm = 0;
    polychord_c_interface_ini(loglikelihood, setup_loglikelihood, const_cast<char *>(inifile.c_str()), &comm);
}
