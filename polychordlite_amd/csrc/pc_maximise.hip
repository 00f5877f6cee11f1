// pc_maximise.hip -- `maximise = T`: polish the best live points into the maximum-likelihood and maximum-posterior points
// and write <root>.maximum.
//
// Restates maximise / do_maximisation / dXdtheta (reference src/polychord/maximiser.F90:32-224), nelder_mead
// (src/polychord/nelder_mead.f90:7-83) and write_max_file (src/polychord/read_write.F90:754-807).  Host code: a simplex of
// nDims + 1 points and a few hundred likelihood calls, made through the same prior / loglikelihood callbacks as the run
// (the built-in device likelihoods are ordinary host functions too, include/polychord_hip.h).  It runs on the live set
// at termination, before the final kill-off, like the reference (nested_sampling.F90:379).
#include "../../include/polychord_hip.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

namespace {

// determinant by elimination with row swaps on zero pivots (nelder_mead.f90:168-209; column-major n x n, destroyed)
double det_cm(std::vector<double> &M, int n)
{
    int sign = 1;
    auto at = [&](int r, int c) -> double & { return M[(size_t)c * n + r]; };
    for (int k = 0; k < n - 1; ++k) {
        if (at(k, k) == 0.0) {
            bool found = false;
            for (int i = k + 1; i < n; ++i)
                if (at(i, k) != 0.0) {
                    for (int j = 0; j < n; ++j) std::swap(at(i, j), at(k, j));
                    found = true; sign = -sign;
                    break;
                }
            if (!found) return 0.0;
        }
        for (int j = k + 1; j < n; ++j) {
            const double m = at(j, k) / at(k, k);
            for (int i = k + 1; i < n; ++i) at(j, i) -= m * at(k, i);
        }
    }
    double d = sign;
    for (int i = 0; i < n; ++i) d *= at(i, i);
    return d;
}

// simplex x[n+1][n] (one point per row), values f[n+1]; maximises func; returns the best vertex (nelder_mead.f90:7-83)
std::vector<double> nelder_mead(const std::function<double(const double *)> &func, std::vector<double> x, std::vector<double> f, double dl)
{
    const int n = (int)f.size() - 1;
    const double alpha = 1.0, gamma = 2.0, rho = 0.5, sigma = 0.5;
    std::vector<int> idx(n + 1);
    std::vector<double> xo(n), xr(n), xe(n), xc(n), E((size_t)n * n);
    double det0 = -1.0;
    auto P = [&](int v) { return x.data() + (size_t)v * n; };
    for (int iter = 0; iter < 200000; ++iter) {
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return f[a] < f[b]; });      // ascending: idx[0] worst, idx[n] best
        for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) E[(size_t)c * n + r] = P(idx[c])[r] - P(idx[n])[r];   // edges from the best vertex
        const double det1 = std::fabs(det_cm(E, n));
        if (det0 < 0.0) det0 = det1;
        if (f[idx[n]] - f[idx[0]] < dl || !(det0 > 0.0) || std::pow(det1 / det0, 1.0 / n) < dl) break;
        for (int r = 0; r < n; ++r) { double s = 0.0; for (int k = 1; k <= n; ++k) s += P(idx[k])[r]; xo[r] = s / n; }   // centroid of all but the worst
        const int w = idx[0];
        for (int r = 0; r < n; ++r) xr[r] = xo[r] + alpha * (xo[r] - P(w)[r]);
        const double fr = func(xr.data());
        if (fr <= f[idx[n]] && f[idx[1]] < fr) { f[w] = fr; std::copy(xr.begin(), xr.end(), P(w)); }
        else if (fr > f[idx[n]]) {                                       // expansion
            for (int r = 0; r < n; ++r) xe[r] = xo[r] + gamma * (xr[r] - xo[r]);
            const double fe = func(xe.data());
            if (fe > fr) { f[w] = fe; std::copy(xe.begin(), xe.end(), P(w)); } else { f[w] = fr; std::copy(xr.begin(), xr.end(), P(w)); }
        } else {                                                          // contraction, else shrink towards the best
            for (int r = 0; r < n; ++r) xc[r] = xo[r] + rho * (P(w)[r] - xo[r]);
            const double fc = func(xc.data());
            if (fc > f[w]) { f[w] = fc; std::copy(xc.begin(), xc.end(), P(w)); }
            else for (int j = 0; j < n; ++j) {
                const int v = idx[j];
                for (int r = 0; r < n; ++r) P(v)[r] = P(idx[n])[r] + sigma * (P(v)[r] - P(idx[n])[r]);
                f[v] = func(P(v));
            }
        }
    }
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return f[a] < f[b]; });
    return std::vector<double>(P(idx[n]), P(idx[n]) + n);
}

std::string e24(double v)
{   // Fortran E24.15E3:  "  0.626931681801488E-001"
    char buf[64];
    if (v == 0.0 || !std::isfinite(v)) {
        if (!std::isfinite(v)) { std::snprintf(buf, sizeof buf, "%24s", std::isnan(v) ? "NaN" : (v > 0 ? "Infinity" : "-Infinity")); return buf; }
        return std::string("   0.000000000000000E+000");
    }
    char t[64];
    std::snprintf(t, sizeof t, "%.14E", std::fabs(v));     // d.ddddddddddddddE+XX
    const char *e = std::strchr(t, 'E');
    const int ex = std::atoi(e + 1) + 1;
    std::string digits;
    digits += t[0];
    digits.append(t + 2, 14);
    std::snprintf(buf, sizeof buf, "%s0.%sE%c%03d", v < 0 ? "-" : "", digits.c_str(), ex < 0 ? '-' : '+', std::abs(ex));
    char out[64];
    std::snprintf(out, sizeof out, "%24s", buf);
    return out;
}

}  // namespace

// live: [nlive][nTotal] rows [cube | theta | phi | birth | logL]; cluster: [nlive] labels; post_mean: [nDims + nDerived] or null
extern "C" int pchip_maximise(polychord_loglike_fn loglike, polychord_prior_fn prior, int nDims, int nDerived, double logzero,
                              const double *live, const int *cluster, int nlive, const double *post_mean, const char *path)
{
    const int D = nDims, nT = 2 * D + nDerived + 2, l0 = nT - 1;
    std::vector<double> theta(D), phi(std::max(1, nDerived));
    auto prior_of = [&](const double *cube, double *th) { prior(const_cast<double *>(cube), th, D); };
    auto dXdtheta = [&](const double *cube) {                             // maximiser.F90:179-207
        const double dx = 1e-5;
        std::vector<double> c0(D), t0(D), t1(D), J((size_t)D * D);
        prior_of(cube, t0.data());
        int s = 1;
        for (int i = 0; i < D; ++i) {
            std::copy(cube, cube + D, c0.begin());
            if (c0[i] + dx >= 1.0) { c0[i] -= dx; s = -s; } else c0[i] += dx;
            prior_of(c0.data(), t1.data());
            for (int r = 0; r < D; ++r) J[(size_t)i * D + r] = t1[r] - t0[r];
        }
        return D * std::log(dx) - std::log(s * det_cm(J, D));
    };
    auto point_of = [&](const double *cube, std::vector<double> &pt) {   // calculate_point (calculate.f90:6-50)
        pt.assign(nT, 0.0);
        std::copy(cube, cube + D, pt.begin());
        bool inside = true;
        for (int d = 0; d < D; ++d) inside = inside && cube[d] >= 0.0 && cube[d] <= 1.0;
        if (!inside) { pt[l0] = logzero; return; }
        prior_of(cube, pt.data() + D);
        pt[l0] = loglike(pt.data() + D, D, pt.data() + 2 * D, nDerived);
    };
    auto maximisation = [&](bool posterior, std::vector<double> &best) -> bool {     // do_maximisation (maximiser.F90:92-161)
        int ncl = 0;
        for (int i = 0; i < nlive; ++i) ncl = std::max(ncl, cluster[i] + 1);
        double max_l = logzero;
        std::vector<double> simplex, f;
        for (int c = 0; c < ncl; ++c) {
            std::vector<std::pair<double, int>> l;
            for (int i = 0; i < nlive; ++i)
                if (cluster[i] == c) l.push_back({ live[(size_t)i * nT + l0] + (posterior ? dXdtheta(live + (size_t)i * nT) : 0.0), i });
            if ((int)l.size() < D + 1) continue;
            std::stable_sort(l.begin(), l.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
            if (l.back().first > max_l) {
                max_l = l.back().first;
                simplex.clear(); f.clear();
                for (size_t k = l.size() - (D + 1); k < l.size(); ++k) {
                    simplex.insert(simplex.end(), live + (size_t)l[k].second * nT, live + (size_t)l[k].second * nT + D);
                    f.push_back(l[k].first);
                }
            }
        }
        if (!(max_l > logzero)) { std::printf("Could not construct simplex\n"); return false; }
        std::vector<double> pt;
        auto func = [&](const double *x) {                                // maximisation_func (maximiser.F90:163-177)
            point_of(x, pt);
            double v = pt[l0];
            if (posterior && v > logzero) v += dXdtheta(x);
            return v;
        };
        const std::vector<double> x = nelder_mead(func, simplex, f, 1e-5);
        point_of(x.data(), best);
        return true;
    };
    std::vector<double> pmax, ppost;
    std::printf("-------------------------------------\nMaximising Likelihood\n");
    if (!maximisation(false, pmax)) return 1;
    std::printf("-------------------------------------\nMaximising Posterior\n");
    if (!maximisation(true, ppost)) return 1;
    const double dX = dXdtheta(ppost.data());
    FILE *fo = std::fopen(path, "w");
    if (!fo) return 2;
    auto row = [&](const double *v, int n) { std::string s; for (int k = 0; k < n; ++k) s += e24(v[k]); std::fprintf(fo, "%s\n", s.c_str()); };
    std::fprintf(fo, "Maximum LogLikelihood:\n"); row(&pmax[l0], 1);
    std::fprintf(fo, "Maximum Likelihood point:\n"); row(pmax.data() + D, D + nDerived); std::fprintf(fo, "\n");
    const double mp = ppost[l0] + dX;
    std::fprintf(fo, "Maximum Posterior:\n"); row(&mp, 1);
    std::fprintf(fo, "Maximum Likelihood at posterior:\n"); row(&ppost[l0], 1);
    std::fprintf(fo, "Maximum Posterior point:\n"); row(ppost.data() + D, D + nDerived); std::fprintf(fo, "\n");
    if (post_mean) {                                                       // maximiser.F90:77-80: likelihood at the posterior mean
        std::vector<double> mean(post_mean, post_mean + D + nDerived);
        const double lm = loglike(mean.data(), D, mean.data() + D, nDerived);
        std::fprintf(fo, "LogLikelihood(mean):\n"); row(&lm, 1);
        std::fprintf(fo, "mean point:\n"); row(mean.data(), D + nDerived);
    }
    std::fclose(fo);
    return 0;
}
