// pc_ini.hip -- ini-file front end: replaces read_params / get_params (src/polychord/ini.f90:44-95,
// 354-458), create_priors + hypercube_to_physical for the prior types used by the shipped examples
// (src/polychord/priors.f90:40-55 uniform, :98-119 log_uniform, :160-183 gaussian, :245-290
// sorted_uniform) and run_polychord_ini (interfaces.F90:232-283, 496-519).
#include "../../include/polychord_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include <algorithm>

namespace {

[[noreturn]] void halt_program(const std::string &msg) { std::fprintf(stderr, "%s\n", msg.c_str()); std::exit(1); }

std::string trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

struct Param { std::string name, latex; int speed = 1; std::string prior; int block = 1; std::vector<double> pp; };

struct Ini {
    std::vector<std::pair<std::string, std::string>> kv;
    std::vector<Param> params; std::vector<std::pair<std::string, std::string>> derived;
    bool has(const std::string &k) const { for (auto &p : kv) if (p.first == k) return true; return false; }
    std::string str(const std::string &k, const std::string &d) const { for (auto &p : kv) if (p.first == k) return p.second; return d; }
    int integer(const std::string &k, int d) const { return has(k) ? std::atoi(str(k, "").c_str()) : d; }
    int integer_required(const std::string &k) const { if (!has(k)) halt_program("ini error: missing key '" + k + "'"); return integer(k, 0); }
    double dbl(const std::string &k, double d) const { return has(k) ? std::atof(str(k, "").c_str()) : d; }
    bool logical(const std::string &k, bool d) const
    {
        if (!has(k)) return d;
        const std::string v = str(k, "");
        return !v.empty() && (v[0] == 'T' || v[0] == 't' || v == ".true.");
    }
    std::vector<double> dbls(const std::string &k) const
    {
        std::vector<double> out; std::stringstream ss(str(k, "")); double v;
        while (ss >> v) out.push_back(v);
        return out;
    }
};

std::vector<std::string> split_bar(const std::string &s)
{
    std::vector<std::string> out; std::stringstream ss(s); std::string item;
    while (std::getline(ss, item, '|')) out.push_back(trim(item));
    return out;
}

Ini read_ini(const std::string &file)
{
    std::ifstream f(file);
    if (!f) halt_program("ini error: cannot open " + file);
    Ini ini; std::string line;
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line);
        if (line.empty() || line[0] == '[') continue;
        if ((line[0] == 'P' || line[0] == 'D') && line.find(':') != std::string::npos && trim(line.substr(1, line.find(':') - 1)).empty()) {
            const auto parts = split_bar(line.substr(line.find(':') + 1));
            if (line[0] == 'P') {           // P : name | latex | speed | prior type | prior block | prior params
                if (parts.size() < 6) halt_program("ini error: malformed parameter line: " + line);
                Param p; p.name = parts[0]; p.latex = parts[1]; p.speed = std::atoi(parts[2].c_str()); p.prior = parts[3];
                p.block = std::atoi(parts[4].c_str());
                if (!p.name.empty() && p.name.back() == '*') p.name.pop_back();       // sub-clustering marker
                std::stringstream ss(parts[5]); double v; while (ss >> v) p.pp.push_back(v);
                ini.params.push_back(p);
            } else {
                if (parts.size() < 2) halt_program("ini error: malformed derived line: " + line);
                ini.derived.push_back({parts[0], parts[1]});
            }
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        ini.kv.push_back({trim(line.substr(0, eq)), trim(line.substr(eq + 1))});
    }
    return ini;
}

// prior blocks of the run in progress (process-global, like the reference's module state)
std::vector<Param> g_params;

double inv_normal_cdf_host(double p)
{   // Wichura AS241 PPND16 (utils.F90:806-966)
    static const double a[8] = { 3.3871328727963666080e+00, 1.3314166789178437745e+02, 1.9715909503065514427e+03, 1.3731693765509461125e+04, 4.5921953931549871457e+04, 6.7265770927008700853e+04, 3.3430575583588128105e+04, 2.5090809287301226727e+03 };
    static const double b[8] = { 1.0, 4.2313330701600911252e+01, 6.8718700749205790830e+02, 5.3941960214247511077e+03, 2.1213794301586595867e+04, 3.9307895800092710610e+04, 2.8729085735721942674e+04, 5.2264952788528545610e+03 };
    static const double c[8] = { 1.42343711074968357734e+00, 4.63033784615654529590e+00, 5.76949722146069140550e+00, 3.64784832476320460504e+00, 1.27045825245236838258e+00, 2.41780725177450611770e-01, 2.27238449892691845833e-02, 7.74545014278341407640e-04 };
    static const double d[8] = { 1.0, 2.05319162663775882187e+00, 1.67638483018380384940e+00, 6.89767334985100004550e-01, 1.48103976427480074590e-01, 1.51986665636164571966e-02, 5.47593808499534494600e-04, 1.05075007164441684324e-09 };
    static const double e[8] = { 6.65790464350110377720e+00, 5.46378491116411436990e+00, 1.78482653991729133580e+00, 2.96560571828504891230e-01, 2.65321895265761230930e-02, 1.24266094738807843860e-03, 2.71155556874348757815e-05, 2.01033439929228813265e-07 };
    static const double f[8] = { 1.0, 5.99832206555887937690e-01, 1.36929880922735805310e-01, 1.48753612908506148525e-02, 7.86869131145613259100e-04, 1.84631831751005468180e-05, 1.42151175831644588870e-07, 2.04426310338993978564e-15 };
    auto poly = [](const double *q, double x) { double v = 0; for (int i = 7; i >= 0; --i) v = v * x + q[i]; return v; };
    if (p <= 0) return -1.7976931348623157e308;
    if (p >= 1) return 1.7976931348623157e308;
    const double q = p - 0.5;
    if (std::fabs(q) <= 0.425) { const double r = 0.180625 - q * q; return q * poly(a, r) / poly(b, r); }
    double r = std::sqrt(-std::log(q < 0 ? p : 1 - p)), v;
    if (r <= 5) { r -= 1.6; v = poly(c, r) / poly(d, r); } else { r -= 5; v = poly(e, r) / poly(f, r); }
    return q < 0 ? -v : v;
}

std::vector<int> g_hyper;   // hypercube index of physical parameter i: parameters are ordered by speed in the cube (priors.f90:708-737)

// minimum number of prior parameters of a base type, or -1 if the type is not supported
int base_prior_nparams(const std::string &t)
{
    if (t == "uniform" || t == "log_uniform" || t == "gaussian" || t == "half_gaussian") return 2;
    if (t == "exponential") return 1;
    if (t == "power_uniform") return 3;
    return -1;
}
std::string base_of(const std::string &prior) { return prior.rfind("sorted_", 0) == 0 ? prior.substr(7) : prior; }

// separable transforms of priors.f90:40-204 on one coordinate y in [0,1]
double base_transform(const std::string &t, double y, const std::vector<double> &pp, const std::string &name)
{
    if (t == "uniform") return pp[0] + (pp[1] - pp[0]) * y;                                   // priors.f90:40-55
    if (t == "log_uniform") return pp[0] * std::pow(pp[1] / pp[0], y);                       // :114-128
    if (t == "gaussian") return pp[0] + pp[1] * inv_normal_cdf_host(y);                      // :73-88
    if (t == "half_gaussian") return pp[0] + pp[1] * inv_normal_cdf_host(0.5 + 0.5 * y);    // :172-187
    if (t == "exponential") return -std::log(1.0 - y) / pp[0];                               // :192-204
    if (t == "power_uniform") {                                                               // :151-167
        const double a = std::pow(pp[0], 1.0 / pp[2]), b = std::pow(pp[1], 1.0 / pp[2]);
        return std::pow(a - y * std::fabs(a - b), pp[2]);
    }
    halt_program("get_priors error: Unknown prior type for parameter " + name);
}

void ini_prior(double *cube_h, double *theta, int nDims)
{   // hypercube_to_physical (priors.f90:494-556): separable blocks, and sorted_* blocks = the order statistics of the
    // block's coordinates (sort_hypercube, priors.f90:245-262) pushed through the separable transform
    std::vector<double> cube(nDims);
    for (int k = 0; k < nDims; ++k) cube[k] = cube_h[g_hyper[k]];
    int i = 0;
    while (i < nDims) {
        const Param &p = g_params[i];
        const std::string base = base_of(p.prior);
        if (p.prior != base) {                    // sorted block: consecutive parameters of the same type and block
            int j = i;
            while (j < nDims && g_params[j].prior == p.prior && g_params[j].block == p.block) ++j;
            const int n = j - i;
            double prev = 1.0;                    // y_n = x_n^(1/n); y_k = y_{k+1} x_k^(1/k)
            for (int k = n; k >= 1; --k) {
                prev = prev * std::pow(cube[i + k - 1], 1.0 / k);
                theta[i + k - 1] = base_transform(base, prev, g_params[i + k - 1].pp, g_params[i + k - 1].name);
            }
            i = j;
            continue;
        }
        theta[i] = base_transform(base, cube[i], p.pp, p.name);
        ++i;
    }
}

}  // namespace

// AS241 / PPND16 on the host (utils.F90:806-966), the inverse normal CDF every Gaussian prior uses
extern "C" double polychord_hip_inv_normal_cdf(double p) { return inv_normal_cdf_host(p); }

// the prior block of an ini file evaluated at one hypercube point (tests; tools that want theta for a cube sample):
// returns the number of parameters, or -1 when `n` is too small
extern "C" int polychord_hip_ini_prior(const char *inifile, const double *cube, double *theta, int n)
{
    const Ini ini = read_ini(inifile ? inifile : "");
    const int nDims = (int)ini.params.size();
    if (nDims > n) return -1;
    g_params = ini.params;
    // hypercube order = parameters by speed (priors.f90:708-737), as in polychord_c_interface_ini
    std::vector<int> distinct;
    for (auto &p : ini.params) distinct.push_back(p.speed);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    g_hyper.assign(nDims, 0);
    int h = 0;
    for (size_t g = 0; g < distinct.size(); ++g)
        for (int i = 0; i < nDims; ++i) if (ini.params[i].speed == distinct[g]) g_hyper[i] = h++;
    for (int i = 0; i < nDims; ++i) {
        const int need = base_prior_nparams(base_of(ini.params[i].prior));
        if (need < 0) halt_program("get_priors error: Unknown prior type for parameter " + ini.params[i].name);
        if ((int)ini.params[i].pp.size() < need) halt_program("ini error: parameter " + ini.params[i].name + " needs " + std::to_string(need) + " prior parameters");
    }
    std::vector<double> c(cube, cube + nDims);
    ini_prior(c.data(), theta, nDims);
    return nDims;
}

extern "C" void polychord_c_interface_ini(polychord_loglike_fn loglikelihood, void (*setup_loglikelihood)(void), char *inifile, int *comm)
{
    const Ini ini = read_ini(inifile ? inifile : "");
    g_params = ini.params;
    const int nDims = (int)ini.params.size(), nDerived = (int)ini.derived.size();
    if (nDims == 0) halt_program("ini error: no 'P :' parameter lines");
    if (setup_loglikelihood) setup_loglikelihood();            // interfaces.F90:273
    // uniform-only priors run on the device (when the likelihood is a built-in); anything else is a host prior
    // grades from the speed column (priors.f90:708-737): speeds relabelled 1,2,3.. in increasing order, the hypercube
    // lists the parameters grade by grade (file order within a grade), grade_dims = parameters per grade
    std::vector<int> speeds(nDims), grade_dims;
    {
        std::vector<int> distinct;
        for (auto &p : ini.params) distinct.push_back(p.speed);
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        g_hyper.assign(nDims, 0);
        int h = 0;
        for (size_t g = 0; g < distinct.size(); ++g) {
            int cnt = 0;
            for (int i = 0; i < nDims; ++i) if (ini.params[i].speed == distinct[g]) { g_hyper[i] = h++; cnt++; }
            grade_dims.push_back(cnt);
        }
    }
    bool identity = true;
    for (int i = 0; i < nDims; ++i) identity = identity && g_hyper[i] == i;
    bool all_uniform = identity;               // the device prior maps cube coordinate i to parameter i
    std::vector<double> lo(nDims), hi(nDims);
    for (int i = 0; i < nDims; ++i) {
        all_uniform &= ini.params[i].prior == "uniform";
        const int need = base_prior_nparams(base_of(ini.params[i].prior));
        if (need < 0) halt_program("get_priors error: Unknown prior type for parameter " + ini.params[i].name);
        if ((int)ini.params[i].pp.size() < need) halt_program("ini error: parameter " + ini.params[i].name + " needs " + std::to_string(need) + " prior parameters");
        lo[i] = ini.params[i].pp[0]; hi[i] = ini.params[i].pp.size() > 1 ? ini.params[i].pp[1] : 0.0;
    }
    polychord_prior_fn prior = ini_prior;
    if (all_uniform) { polychord_hip_set_uniform_prior(nDims, lo.data(), hi.data()); prior = polychord_hip_uniform_prior; }
    std::vector<double> grade_frac = ini.dbls("grade_frac");
    if (grade_frac.empty()) grade_frac = {1.0};
    if (grade_frac.size() != grade_dims.size()) {
        if (grade_dims.size() == 1) grade_frac.resize(1);
        else halt_program("ini error: grade_frac needs one entry per parameter speed");
    }
    std::vector<double> loglikes = ini.dbls("loglikes"), nl = ini.dbls("nlives");
    std::vector<int> nlives(nl.begin(), nl.end());
    const int n_nlives = (int)std::min(loglikes.size(), nlives.size());
    std::string base = ini.str("base_dir", "chains"), root = ini.str("file_root", "test");
    // write the .paramnames file (read_write.F90:964+) when asked
    if (ini.logical("write_paramnames", false)) {
        FILE *f = std::fopen((base + "/" + root + ".paramnames").c_str(), "w");
        if (f) {
            for (auto &p : ini.params) std::fprintf(f, "%s      %s\n", p.name.c_str(), p.latex.c_str());
            for (auto &d : ini.derived) std::fprintf(f, "%s*     %s\n", d.first.c_str(), d.second.c_str());
            std::fclose(f);
        }
        f = std::fopen((base + "/" + root + ".properties.ini").c_str(), "w");      // read_write.F90:996-1014
        if (f) { std::fprintf(f, "sampler=nested\nlabel=%s\n", root.c_str()); std::fclose(f); }
    }
    polychord_c_interface(loglikelihood, prior, nullptr, ini.integer_required("nlive"), ini.integer_required("num_repeats"),
                          ini.integer("nprior", -1), ini.integer("nfail", -1), ini.logical("do_clustering", false), ini.integer("feedback", 1),
                          ini.dbl("precision_criterion", 1e-3), ini.dbl("logzero", -1e30), ini.integer("max_ndead", -1),
                          ini.dbl("boost_posterior", 0.0), ini.logical("posteriors", false), ini.logical("equals", false),
                          ini.logical("cluster_posteriors", false), ini.logical("write_resume", false), false /* written above with the ini names */,
                          ini.logical("read_resume", false), ini.logical("write_stats", true), ini.logical("write_live", false),
                          ini.logical("write_dead", true), ini.logical("write_prior", false), ini.logical("maximise", false),
                          ini.dbl("compression_factor", std::exp(-1.0)), ini.logical("synchronous", true), nDims, nDerived,
                          (char *)base.c_str(), (char *)root.c_str(), (int)grade_dims.size(), grade_frac.data(), grade_dims.data(), n_nlives,
                          loglikes.empty() ? nullptr : loglikes.data(), nlives.empty() ? nullptr : nlives.data(), ini.integer("seed", -1), comm);
}
