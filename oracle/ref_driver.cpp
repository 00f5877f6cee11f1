/*
 * ref_driver.cpp -- drives the REFERENCE's own C entry point
 * polychord_c_interface (src/polychord/interfaces.h:2-45) with C likelihoods that
 * restate the Fortran examples (gaussian.f90, rastrigin.f90, twin_gaussian.f90, random_gaussian.f90),
 * exactly as SURVEY.md 8(c) prescribes.  Used (a) to generate tests/golden/*.json
 * in this container and (b) as bench.py's cpu_baseline kind="reference" on the GPU box.
 * Test infrastructure only.
 *
 * usage: ref_driver <like> <nDims> <nDerived> <nlive> <nrepeats> <seed> <clustering> <base_dir> <root> [write_dead [resume_snapshot_after_ndead]]
 * like = corr_gaussian (likelihoods/examples/random_gaussian.f90:17-30, log_gauss utils.F90:1028-1048): the inverse covariance
 *   comes from the file REF_COV_FILE names -- nDims x nDims doubles (row-major), nDims means, log det(covariance) -- because
 *   the reference draws its matrix from the compiler's generator BEFORE the seed is applied (gotcha 6): the caller
 *   builds the matrix by the same construction (random_utils.F90:581-614) and hands the same one to the engine.
 * REF_MAX_NDEAD=<n>: stop after n dead points (settings%max_ndead), for bounded timing samples of long configurations.
 * prints one JSON line: {"logZ":..,"logZerr":..,"ndead":..,"nlike":..,"wall":..}
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <chrono>
#include <sys/resource.h>
#include <sys/stat.h>

extern "C" void polychord_c_interface(
    double (*)(double *, int, double *, int), void (*)(double *, double *, int),
    void (*)(int, int, int, double *, double *, double *, double, double),
    int, int, int, int, bool, int, double, double, int, double, bool, bool, bool, bool, bool, bool, bool,
    bool, bool, bool, bool, double, bool, int, int, char *, char *, int, double *, int *, int, double *,
    int *, int, int &);
#ifdef REF_MPI
#include <mpi.h>
extern "C" void pc_shim_set_rank(unsigned) __attribute__((weak));
#endif
extern "C" void pc_shim_reset(unsigned) __attribute__((weak));
extern "C" unsigned long pc_shim_consumed(void) __attribute__((weak));

static long g_calls = 0;
static double g_lo = 0.0, g_hi = 1.0;
static const double LOG_TWO_PI = 1.8378770664093453;

static double gaussian(double *th, int D, double *phi, int nDer)
{   // likelihoods/examples/gaussian.f90:12-41
    g_calls++;
    const double mu = 0.5, sigma = 0.1;
    double s = 0, r2 = 0;
    for (int d = 0; d < D; ++d) { double z = (th[d] - mu) / sigma; s += z * z; r2 += (th[d] - mu) * (th[d] - mu); }
    if (nDer >= 1) phi[0] = std::sqrt(r2);
    if (nDer >= 2) phi[1] = std::log(std::pow(phi[0], (double)D) * std::pow(std::sqrt(3.14159265358979323846), (double)D) / std::tgamma(1.0 + D / 2.0));
    return -(double)D * (std::log(sigma) + LOG_TWO_PI / 2.0) - s / 2.0;
}
static double rastrigin(double *th, int D, double *, int)
{   // likelihoods/examples/rastrigin.f90:20-35
    g_calls++;
    double s = 0;
    for (int d = 0; d < D; ++d) s += std::log(4991.21750) + th[d] * th[d] - 10.0 * std::cos(6.283185307179586 * th[d]);
    return -s;
}
static double twin(double *th, int D, double *phi, int nDer)
{   // likelihoods/examples/twin_gaussian.f90:14-56
    g_calls++;
    const double sigma = 0.1;
    double n = -(double)D * (std::log(sigma) + LOG_TWO_PI / 2.0), s1 = 0, s2 = 0;
    for (int d = 0; d < D; ++d) {
        double m1 = d < 2 ? -0.5 : 0.0, m2 = d < 2 ? 0.5 : 0.0;
        double z1 = (th[d] - m1) / sigma, z2 = (th[d] - m2) / sigma; s1 += z1 * z1; s2 += z2 * z2;
    }
    if (nDer >= 1) phi[0] = th[0] > 0.5 ? 1.0 : -1.0;
    double a = n - s1 / 2, b = n - s2 / 2;
    double la = a > b ? a + std::log(std::exp(b - a) + 1) : b + std::log(std::exp(a - b) + 1);
    return la - std::log(2.0);
}
static std::vector<double> g_invcov, g_mean; static double g_logdet = 0.0;
static double corr_gaussian(double *th, int D, double *, int)
{   // likelihoods/examples/random_gaussian.f90:17-30 -> utils.F90 log_gauss
    g_calls++;
    double q = 0;
    for (int a = 0; a < D; ++a) {
        double t = 0;
        const double *row = g_invcov.data() + (size_t)a * D;
        for (int b = 0; b < D; ++b) t += row[b] * (th[b] - g_mean[b]);
        q += (th[a] - g_mean[a]) * t;
    }
    return -((double)D * LOG_TWO_PI + g_logdet) / 2.0 - q / 2.0;
}
static void prior(double *cube, double *theta, int D) { for (int d = 0; d < D; ++d) theta[d] = g_lo + (g_hi - g_lo) * cube[d]; }
// with write_resume the dumper keeps a copy of the first .resume file written after `g_snap_after` deaths:
// the file at the end of a run holds no live points any more (golden fixture for the resume grammar)
static std::string g_resume_path; static int g_snap_after = -1; static bool g_snapped = false;
static void dumper(int ndead, int, int, double *, double *, double *, double, double)
{
    if (g_snap_after < 0 || g_snapped || ndead < g_snap_after) return;
    FILE *in = std::fopen(g_resume_path.c_str(), "r");
    if (!in) return;
    FILE *out = std::fopen((g_resume_path + "_mid").c_str(), "w");
    char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, in)) > 0) std::fwrite(buf, 1, n, out);
    std::fclose(in); std::fclose(out); g_snapped = true;
}

int main(int argc, char **argv)
{
    if (argc < 10) { std::fprintf(stderr, "usage: %s like nDims nDerived nlive nrepeats seed clustering base_dir root [write_dead]\n", argv[0]); return 2; }
    struct rlimit rl; getrlimit(RLIMIT_STACK, &rl); rl.rlim_cur = rl.rlim_max; setrlimit(RLIMIT_STACK, &rl);
    std::string like = argv[1];
    int nDims = atoi(argv[2]), nDer = atoi(argv[3]), nlive = atoi(argv[4]), nrep = atoi(argv[5]), seed = atoi(argv[6]);
    bool clustering = atoi(argv[7]) != 0;
    std::string base = argv[8], root = argv[9];
    bool write_dead = argc > 10 && atoi(argv[10]) != 0;
    bool write_resume = argc > 11 && atoi(argv[11]) >= 0;
    if (write_resume) { g_snap_after = atoi(argv[11]); g_resume_path = base + "/" + root + ".resume"; }
    mkdir(base.c_str(), 0755); mkdir((base + "/clusters").c_str(), 0755);
    double (*fn)(double *, int, double *, int) = gaussian;
    if (like == "rastrigin") { fn = rastrigin; g_lo = -5.12; g_hi = 5.12; }
    else if (like == "twin_gaussian") { fn = twin; g_lo = -1.0; g_hi = 1.0; }
    else if (like == "corr_gaussian") {
        const char *cf = std::getenv("REF_COV_FILE");
        FILE *f = cf ? std::fopen(cf, "rb") : nullptr;
        if (!f) { std::fprintf(stderr, "corr_gaussian needs REF_COV_FILE\n"); return 2; }
        g_invcov.resize((size_t)nDims * nDims); g_mean.resize(nDims);
        const bool ok = std::fread(g_invcov.data(), sizeof(double), g_invcov.size(), f) == g_invcov.size() &&
                        std::fread(g_mean.data(), sizeof(double), nDims, f) == (size_t)nDims && std::fread(&g_logdet, sizeof(double), 1, f) == 1;
        std::fclose(f);
        if (!ok) { std::fprintf(stderr, "REF_COV_FILE: short read\n"); return 2; }
        fn = corr_gaussian;
    }
    else if (like != "gaussian") { std::fprintf(stderr, "unknown likelihood %s\n", like.c_str()); return 2; }
    // optional: REF_GRADES="dims,dims;repeats,repeats" -- explicit repeats per grade (every grade_frac > 1:
    // the deterministic branch of generate.F90:303-309)
    double grade_frac[8] = { 1.0 }; int grade_dims[8] = { nDims }; int nGrade = 1;
    if (const char *e = std::getenv("REF_GRADES")) {
        std::string t = e; const size_t semi = t.find(';');
        std::string a = t.substr(0, semi), b = t.substr(semi + 1);
        nGrade = 0;
        for (size_t pos = 0; pos <= a.size() && nGrade < 8;) { size_t k = a.find(',', pos); if (k == std::string::npos) k = a.size(); grade_dims[nGrade++] = atoi(a.substr(pos, k - pos).c_str()); pos = k + 1; }
        int m = 0;
        for (size_t pos = 0; pos <= b.size() && m < 8;) { size_t k = b.find(',', pos); if (k == std::string::npos) k = b.size(); grade_frac[m++] = atof(b.substr(pos, k - pos).c_str()); pos = k + 1; }
    }
    // optional: REF_NPRIOR=<n>, REF_NLIVES="logL:n,logL:n" (dynamic nlive, run_time_info.f90:766-779)
    double loglikes[8] = { 0 }; int nlives[8] = { 0 }; int n_nlives = 0, nprior = -1;
    if (const char *e = std::getenv("REF_NPRIOR")) nprior = atoi(e);
    if (const char *e = std::getenv("REF_NLIVES")) {
        std::string t = e; size_t pos = 0;
        while (pos < t.size() && n_nlives < 8) {
            size_t c = t.find(':', pos), k = t.find(',', pos);
            if (c == std::string::npos) break;
            if (k == std::string::npos) k = t.size();
            loglikes[n_nlives] = atof(t.substr(pos, c - pos).c_str()); nlives[n_nlives] = atoi(t.substr(c + 1, k - c - 1).c_str());
            n_nlives++; pos = k + 1;
        }
    }
    int comm = 0, rank = 0;
    if (pc_shim_reset) pc_shim_reset((unsigned)seed);
#ifdef REF_MPI
    // the reference's farm: rank 0 administers, ranks 1 .. nprocs-1 sample (REF_SYNC=0: its asynchronous mode, not reproducible)
    MPI_Init(&argc, &argv);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    comm = (int)MPI_Comm_c2f(MPI_COMM_WORLD);
    if (pc_shim_set_rank) pc_shim_set_rank((unsigned)rank);
#endif
    const bool synchronous = !(std::getenv("REF_SYNC") && atoi(std::getenv("REF_SYNC")) == 0);
    auto t0 = std::chrono::steady_clock::now();
    const bool maximise = std::getenv("REF_MAXIMISE") != nullptr;       // optional: <root>.maximum (maximiser.F90)
    // optional: REF_POSTERIORS = "pe" / "p" / "e" (weighted and / or equally weighted posterior files, update_posteriors
    // run_time_info.f90:955-1066), REF_BOOST = boost_posterior, REF_CLUSTER_POST = per-cluster posterior files
    const char *rp = std::getenv("REF_POSTERIORS");
    const bool posteriors = rp && std::strchr(rp, 'p'), equals = rp && std::strchr(rp, 'e');
    const double boost = std::getenv("REF_BOOST") ? atof(std::getenv("REF_BOOST")) : 0.0;
    const bool cluster_post = std::getenv("REF_CLUSTER_POST") != nullptr;
    const int max_ndead = std::getenv("REF_MAX_NDEAD") ? atoi(std::getenv("REF_MAX_NDEAD")) : -1;
    polychord_c_interface(fn, prior, dumper, nlive, nrep, nprior, -1, clustering, 0, 0.001, -1e30, max_ndead, boost,
                          posteriors, equals, cluster_post, write_resume, false, false, true, false, write_dead, false, maximise,
                          0.36787944117144233, synchronous, nDims, nDer, (char *)base.c_str(), (char *)root.c_str(),
                          nGrade, grade_frac, grade_dims, n_nlives, loglikes, nlives, seed, comm);
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
#ifdef REF_MPI
    {
        long calls_all = 0;
        MPI_Reduce(&g_calls, &calls_all, 1, MPI_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
        g_calls = calls_all;
        unsigned long used = pc_shim_consumed ? pc_shim_consumed() : 0ul;
        if (std::getenv("REF_RNG_TRACE")) std::fprintf(stderr, "rank %d consumed %lu draws\n", rank, used);
        MPI_Finalize();
        if (rank != 0) return 0;
    }
#endif
    // parse <base>/<root>.stats (read_write.F90:842-889)
    std::string fn_stats = base + "/" + root + ".stats";
    FILE *f = std::fopen(fn_stats.c_str(), "r");
    double logZ = 0, err = 0; long ndead = 0, nlike = 0, npost = 0, nequals = 0; int ncl = 0; char line[512]; char nlike_line[256] = "";
    while (f && std::fgets(line, sizeof line, f)) {
        if (std::strncmp(line, "log(Z)", 6) == 0 && std::strstr(line, "+/-")) {
            const char *eq = std::strchr(line, '='); if (eq) std::sscanf(eq + 1, "%lf +/- %lf", &logZ, &err);
        }
        if (std::strstr(line, "ndead:")) std::sscanf(std::strstr(line, "ndead:") + 6, "%ld", &ndead);
        if (std::strstr(line, " nlike:")) { std::sscanf(std::strstr(line, "nlike:") + 6, "%ld", &nlike); std::snprintf(nlike_line, sizeof nlike_line, "%s", std::strstr(line, "nlike:") + 6); for (char *q = nlike_line; *q; ++q) if (*q == '\n') *q = 0; }
        if (std::strstr(line, "ncluster:")) std::sscanf(std::strstr(line, "ncluster:") + 9, "%d", &ncl);
        if (std::strstr(line, "nposterior:")) std::sscanf(std::strstr(line, "nposterior:") + 11, "%ld", &npost);
        if (std::strstr(line, "nequals:")) std::sscanf(std::strstr(line, "nequals:") + 8, "%ld", &nequals);
    }
    if (f) std::fclose(f);
    std::printf("{\"like\":\"%s\",\"nDims\":%d,\"nlive\":%d,\"num_repeats\":%d,\"seed\":%d,\"logZ\":%.15g,\"logZerr\":%.15g,"
                "\"ndead\":%ld,\"nlike\":%ld,\"ncluster\":%d,\"nposterior\":%ld,\"nequals\":%ld,\"calls\":%ld,\"rng_consumed\":%lu,\"wall\":%.4f,\"nlike_grades\":\"%s\"}\n",
                like.c_str(), nDims, nlive, nrep, seed, logZ, err, ndead, nlike, ncl, npost, nequals, g_calls,
                pc_shim_consumed ? pc_shim_consumed() : 0ul, wall, nlike_line);
    return 0;
}
