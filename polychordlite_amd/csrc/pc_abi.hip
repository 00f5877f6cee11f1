// pc_abi.hip -- the drop-in boundary: the reference's C symbols on top of the HIP engine.
//
//   polychord_c_interface      replaces src/polychord/interfaces.F90:285-436 (prototype interfaces.h:2-45)
//   polychord_c_interface_ini  replaces src/polychord/interfaces.F90:496-519 (prototype interfaces.h:47-56)
// Output files follow src/polychord/read_write.F90 (formats E24.15E3 / I8, utils.F90:19-21):
//   <root>.stats (:809-910), <root>_dead.txt / <root>_dead-birth.txt (:679-719),
//   <root>_phys_live.txt / -birth (:621-676), <root>.paramnames is written by the Python layer.
// Fatal conditions follow abort.F90:19-29: message on stderr, exit status 1.
#include "../../include/polychord_hip.h"
#include "pc_resume.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <sys/stat.h>
#include <chrono>

extern "C" double polychord_hip_keyed_uniform(unsigned seed, unsigned dom, unsigned shi, unsigned slo, unsigned idx);

namespace {

struct Builtins {
    double g_mu = 0.5, g_sigma = 0.1;                 // likelihoods/examples/gaussian.f90:25-26
    int cg_D = 0; std::vector<double> cg_invcov, cg_mean; double cg_logdet = 0.0;
    int up_D = 0; std::vector<double> up_lo, up_hi;
    int batch = 0, device = -1, epoch_discard = 0;
    bool halt_returns = false;                        // fatal conditions return to the caller instead of `stop 1` (language bindings)
    std::string last_error;
} G;

const double LOG_TWO_PI = 1.8378770664093453;

struct HaltRequest { std::string msg; };
[[noreturn]] void halt_program(const char *msg)
{   // abort.F90:19-29: message + `stop 1`.  A language binding that must survive (polychord_hip_set_option
    // "halt_returns") gets the message through polychord_hip_last_error() and the entry point returns instead.
    std::fprintf(stderr, "%s\n", msg);
    if (G.halt_returns) throw HaltRequest{msg};
    std::exit(1);
}

// Fortran E24.15E3:  "  0.626931681801488E-001"
std::string fmt_e24(double v)
{
    char buf[64];
    if (v == 0.0 || !std::isfinite(v)) {
        if (!std::isfinite(v)) { std::snprintf(buf, sizeof buf, "%24s", std::isnan(v) ? "NaN" : (v > 0 ? "Infinity" : "-Infinity")); return buf; }
        return std::string("   0.000000000000000E+000");
    }
    char t[64];
    std::snprintf(t, sizeof t, "%.14E", std::fabs(v));     // d.ddddddddddddddE+XX
    const char *e = std::strchr(t, 'E');
    int ex = std::atoi(e + 1) + 1;
    std::string digits;
    digits += t[0];
    digits.append(t + 2, 14);
    std::snprintf(buf, sizeof buf, "%s0.%sE%c%03d", v < 0 ? "-" : "", digits.c_str(), ex < 0 ? '-' : '+', std::abs(ex));
    char out[64];
    std::snprintf(out, sizeof out, "%24s", buf);
    return out;
}

// Files of a run (read_write.F90:479-910, names :1022-1224), written from the engine's update hook:
//   <root>_dead.txt / _dead-birth.txt   appended as points die (the reference rewrites them in full at
//                                       every update -- most of its wall time at small nlive)
//   <root>_phys_live.txt / -birth.txt   rewritten at every update
//   <root>.stats                        rewritten at every update ("Still Active" clusters listed first)
//   <root>.txt / _equal_weights.txt     weighted posterior from the dead points at the end; the equally weighted one is
//                                       thinned at every update like the reference's and written at the end
struct FileSink {
    std::string base, root;
    int nDims = 0, nDer = 0;
    bool write_stats = false, write_live = false, write_dead = false, posteriors = false, equals = false, write_prior = false,
         cluster_posteriors = false;
    unsigned seed = 0; double logzero = -1e30, compression = 0.36787944117144233; int num_repeats = 1;
    long dead_written = 0, nlike_last = 0; int nposterior = 0, nequals = 0;
    // the equally weighted posterior, thinned at every update like the reference's (run_time_info.f90:975-1026): entries =
    // (log weight the entry was last accepted at, index of the point: dead point, or ndead_final + phantom kept by boost)
    std::vector<double> eq_w; std::vector<long> eq_i; std::vector<char> eq_x;
    long eq_done = 0, eq_xdone = 0; unsigned eq_round = 0; int eq_ncd_seen = 0; double eq_max = -1.7e308;
    std::vector<long> nlike_last_g;
    std::vector<double> mu, sig;
    int feedback = 0, nlive_set = 1;
    int epoch_note = -1;          // >= 0: the run's epoch_discard rule goes to the end of <root>.stats (clustered runs with a nursery of several chains)

    // progress block of an update, in the layout of feedback.f90:221-315 (the rows this engine keeps on the host:
    // live points per cluster, counters, evidences -- per cluster in order of decreasing evidence)
    void progress(const pchip_update &u) const
    {
        int wmax = 1;
        for (int p = 0; p < u.ncluster; ++p) wmax = std::max(wmax, u.nlive_p[p]);
        const int iw = std::max(1, (int)std::ceil(std::log10((double)wmax)));
        std::string bar((size_t)(iw + 2) * (size_t)std::max(1, u.ncluster) + 11, '_');
        std::printf("%s\nlives      |", bar.c_str());
        for (int p = 0; p < u.ncluster; ++p) std::printf("%*d |", iw, u.nlive_p[p]);
        std::printf("\n");
        for (size_t i = 0; i < bar.size(); ++i) std::printf("\xE2\x80\xBE");
        std::printf("\nncluster   =%8d /%8d\nndead      =%20ld\n", u.ncluster, u.ncluster + u.ncluster_dead, u.ndead);
        const int ng = u.ngrade > 0 ? u.ngrade : 1;
        std::printf("nlike      =");
        for (int g = 0; g < ng; ++g) std::printf("%20ld", u.nlike_grade ? u.nlike_grade[g] : u.nlike);
        std::printf("\n<nlike>    =");
        std::vector<double> since((size_t)ng);
        for (int g = 0; g < ng; ++g) since[g] = (double)((u.nlike_grade ? u.nlike_grade[g] : u.nlike) - (g < (int)nlike_last_g.size() ? nlike_last_g[g] : 0L));
        for (int g = 0; g < ng; ++g) std::printf("%15.2f", since[g] / nlive_set);
        std::printf("   (");
        for (int g = 0; g < ng; ++g) std::printf("%15.2f", since[g] / ((double)(u.grade_repeats ? u.grade_repeats[g] : num_repeats) * nlive_set));
        std::printf(" per slice )\n");
        if (std::fabs(u.logZ) < 1e9) std::printf("log(Z)     = %15.2f +/- %5.2f\n", u.logZ, u.logZerr);
        else std::printf("log(Z)     = ?\n");
        std::vector<std::pair<double, int>> ord;
        for (int p = 0; p < u.ncluster; ++p) ord.push_back({-u.logZp[p], p});
        for (int p = 0; p < u.ncluster_dead; ++p) ord.push_back({-u.logZp_dead[p], u.ncluster + p});
        std::stable_sort(ord.begin(), ord.end());
        for (size_t k = 0; k < ord.size() && ord.size() > 1; ++k) {
            const int p = ord[k].second;
            const bool alive = p < u.ncluster;
            const double z = alive ? u.logZp[p] : u.logZp_dead[p - u.ncluster], e = alive ? u.logZperr[p] : u.logZperr_dead[p - u.ncluster];
            if (std::fabs(z) < 1e9) std::printf("log(Z_%zu)%*s= %15.2f +/- %5.2f%s\n", k + 1, k + 1 < 9 ? 3 : (k + 1 < 99 ? 2 : 1), "", z, e, alive ? " (still evaluating)" : "");
            else std::printf("log(Z_%zu)%*s= ?%s\n", k + 1, k + 1 < 9 ? 3 : (k + 1 < 99 ? 2 : 1), "", alive ? " (still evaluating)" : "");
        }
        std::printf("\n\n\n");
        std::fflush(stdout);
    }

    std::string path(const char *suffix) const { return base + "/" + root + suffix; }
    static FILE *open(const std::string &p, const char *mode)
    {
        FILE *f = std::fopen(p.c_str(), mode);
        if (!f) halt_program(("polychord_hip: cannot open " + p).c_str());
        return f;
    }
    void rows(FILE *f, const double *rows_, long i0, long i1, int npars, bool logl_first, bool birth) const
    {
        const int np = nDims + nDer;
        std::string line;
        for (long i = i0; i < i1; ++i) {
            const double *r = rows_ + (size_t)i * npars;
            line.clear();
            if (logl_first) {                     // read_write.F90:698-703: logL, theta, phi
                line += fmt_e24(r[np + 1]);
                for (int k = 0; k < np; ++k) line += fmt_e24(r[k]);
            } else {                              // read_write.F90:707-716: theta, phi, logL[, birth]
                for (int k = 0; k < np; ++k) line += fmt_e24(r[k]);
                line += fmt_e24(r[np + 1]);
                if (birth) line += fmt_e24(r[np]);
            }
            line += '\n';
            std::fwrite(line.data(), 1, line.size(), f);
        }
    }
    void stats(const pchip_update &u) const
    {   // read_write.F90:809-910
        FILE *f = open(path(".stats"), "w");
        std::fprintf(f, "Evidence estimates:\n===================\n");
        std::fprintf(f, "  - The evidence Z is a log-normally distributed, with location and scale parameters mu and sigma.\n");
        std::fprintf(f, "  - We denote this as log(Z) = mu +/- sigma.\n\nGlobal evidence:\n----------------\n\n");
        std::fprintf(f, "log(Z)       = %s +/- %s\n\n\n", fmt_e24(u.logZ).c_str(), fmt_e24(u.logZerr).c_str());
        std::fprintf(f, "Local evidences:\n----------------\n\n");
        char lab[32];
        for (int p = 0; p < u.ncluster; ++p) {
            std::snprintf(lab, sizeof lab, "log(Z_%d)", p + 1);
            std::fprintf(f, "%-13s= %s +/- %s (Still Active)\n", lab, fmt_e24(u.logZp[p]).c_str(), fmt_e24(u.logZperr[p]).c_str());
        }
        for (int p = 0; p < u.ncluster_dead; ++p) {
            std::snprintf(lab, sizeof lab, "log(Z_%d)", p + 1 + u.ncluster);
            std::fprintf(f, "%-13s= %s +/- %s\n", lab, fmt_e24(u.logZp_dead[p]).c_str(), fmt_e24(u.logZperr_dead[p]).c_str());
        }
        std::fprintf(f, "\n\nRun-time information:\n---------------------\n\n");
        std::fprintf(f, " ncluster:   %8d /%8d\n", u.ncluster, u.ncluster + u.ncluster_dead);
        std::fprintf(f, " nposterior: %8d\n", nposterior);
        std::fprintf(f, " nequals:    %8d\n", nequals);
        std::fprintf(f, " ndead:      %8ld\n", u.ndead);
        std::fprintf(f, " nlive:      %8d\n", u.nlive);
        // one column per grade (read_write.F90:880-889)
        const int ng = u.ngrade > 0 ? u.ngrade : 1;
        auto last_g = [&](int g) { return g < (int)nlike_last_g.size() ? nlike_last_g[g] : 0L; };
        std::fprintf(f, " nlike:      ");
        for (int g = 0; g < ng; ++g) std::fprintf(f, "%8ld", u.nlike_grade ? u.nlike_grade[g] : u.nlike);
        std::fprintf(f, "\n <nlike>:    ");
        std::vector<double> per_it(ng, 0.0), per_slice(ng, 0.0);
        if (u.nlive > 0) {                        // likelihood calls per iteration since the last update
            const double upd = -(double)u.nlive * std::log(compression);
            for (int g = 0; g < ng; ++g) {
                per_it[g] = (double)((u.nlike_grade ? u.nlike_grade[g] : u.nlike) - last_g(g)) / upd;
                per_slice[g] = per_it[g] / (double)(u.grade_repeats ? u.grade_repeats[g] : num_repeats);
            }
        }
        for (int g = 0; g < ng; ++g) std::fprintf(f, "%8.2f", per_it[g]);
        std::fprintf(f, "   (");
        for (int g = 0; g < ng; ++g) std::fprintf(f, "%8.2f", per_slice[g]);
        std::fprintf(f, " per slice )\n");
        if (posteriors && !mu.empty()) {
            std::fprintf(f, "\n\nDim No.       Mean        Sigma\n");
            for (int k = 0; k < nDims + nDer; ++k) {
                if (k == nDims) std::fprintf(f, "-------------------------------\n");
                std::fprintf(f, "%3d%s +/- %s\n", k + 1, fmt_e24(mu[k]).c_str(), fmt_e24(sig[k]).c_str());
            }
            if (nDer == 0) std::fprintf(f, "-------------------------------\n");
        }
        // behind everything the reference's readers look at: the one engine-specific sampling rule a clustered run with a nursery of
        // several chains follows (include/polychord_hip.h pchip_settings.epoch_discard); batch = 1 and unclustered runs have none
        if (epoch_note >= 0)
            std::fprintf(f, "\n\npolychord_hip: chains in flight when the list of clusters changes: epoch_discard = %d (%s)\n", epoch_note,
                         epoch_note ? "all discarded: the reference farm's rule, nested_sampling.F90:313" : "only those seeded in the cluster that ended are lost");
        std::fclose(f);
    }
    // weighted / equally weighted posteriors from the dead points (update_posteriors, run_time_info.f90:955-1066;
    // write_posterior_file, read_write.F90:479-617); the equally weighted list comes from thin_equals_round.
    // One round of update_posteriors for the global equal-weight list: survivors of the earlier rounds are re-drawn against
    // the ratio of their weight to the largest weight so far and move up to it, the points that joined the posterior stack
    // since the last round (deaths in death order, then the phantoms boost_posterior kept) are drawn against it.
    // The reference's trials are independent draws of its one generator, and WHICH draw a row gets is an accident of its
    // arrays' order (per-cluster stacks walked cluster by cluster, delete = overwrite-with-last, array_utils.f90:433-458).
    // Here -- and in the oracle's keyed mode, oracle/pc_oracle.c bernoulli_post -- the trial of posterior row r in thinning
    // round k is the draw keyed by (k, r): r = the dead point's index in death order, or the kept phantom's id with the top
    // bit of the round word set.  The rows that survive are then the same whatever order the lists are walked in: with
    // several clusters, and with phantoms in the stack, the file holds the oracle's rows (tests: sorted, row for row).
    void thin_equals_round(const pchip_update &u)
    {
        // The reference makes a round at every update AND whenever a cluster has lost its last point (delete_cluster calls
        // update_posteriors, run_time_info.f90:533-534; also for every cluster that ends in the final kill-off, nested_sampling.F90:381-386).
        // The engine deletes clusters on the device with no host call, but the moments can be read off the records: a cluster ended with
        // the last point that died in it.  Rounds in the reference's order: the cluster ends before this hook's death count, ascending; the
        // update itself (the last cluster's end, when the run is over); then the ends that coincide with it (nested_sampling.F90:321-339: the
        // update's round comes first).  Phantoms kept by boost_posterior join at the hooks only (with several clusters AND boost_posterior the
        // rounds between differ from the reference's: its clean_phantoms runs in them too).
        std::vector<long> ends;
        for (int k = eq_ncd_seen; k < u.ncluster_dead; ++k) {
            long last = -1;
            for (long i = u.ndead - 1; i >= 0 && last < 0; --i) if (u.dead_cluster[i] == u.cluster_uid_dead[k]) last = i;
            if (last >= 0) ends.push_back(last + 1);
        }
        eq_ncd_seen = u.ncluster_dead;
        std::sort(ends.begin(), ends.end());
        for (long b : ends) if (b < u.ndead) thin_equals_to(u, b, eq_xdone);
        thin_equals_to(u, u.ndead, u.n_extra);
        for (long b : ends) if (b >= u.ndead) thin_equals_to(u, u.ndead, u.n_extra);
    }
    // one round: the deaths before `nd` and the kept phantoms before `nx` that no round has seen yet
    void thin_equals_to(const pchip_update &u, long nd, long nx)
    {
        const unsigned round = ++eq_round;
        auto draw = [&](long i, bool extra) {
            const unsigned long long id = extra ? (u.extra_uid ? u.extra_uid[i] : (unsigned long long)i) : (unsigned long long)i;
            return polychord_hip_keyed_uniform(seed, 6u, (round & 0x0FFFFFFFu) | (extra ? 0x80000000u : 0u), (unsigned)((id & 0x7FFFFFFFFFFFFFFFull) >> 32), (unsigned)id);
        };
        // a failed spawn carries logweight = the run's own logzero (run_time_info.f90:781-785): logpost - logL <= logzero
        auto lived = [&](long i) { return u.logpost[i] - u.dead[(size_t)i * u.npars + u.npars - 1] > logzero; };
        if (nd < eq_done) nd = eq_done;
        if (nx < eq_xdone) nx = eq_xdone;
        for (long i = eq_done; i < nd; ++i) if (lived(i)) eq_max = std::max(eq_max, u.logpost[i]);
        for (long i = eq_xdone; i < nx; ++i) eq_max = std::max(eq_max, u.extra_logpost[i]);
        for (size_t i = 0; i < eq_w.size();) {
            if (eq_w[i] < eq_max) {
                if (draw(eq_i[i], eq_x[i] != 0) < std::exp(eq_w[i] - eq_max)) { eq_w[i] = eq_max; ++i; }
                else { eq_w[i] = eq_w.back(); eq_i[i] = eq_i.back(); eq_x[i] = eq_x.back(); eq_w.pop_back(); eq_i.pop_back(); eq_x.pop_back(); }
            } else ++i;
        }
        for (long i = eq_done; i < nd; ++i) {
            if (!lived(i)) continue;                            // failed spawns never entered the stack
            if (draw(i, false) < std::exp(u.logpost[i] - eq_max)) { eq_w.push_back(eq_max); eq_i.push_back(i); eq_x.push_back(0); }
        }
        for (long i = eq_xdone; i < nx; ++i)
            if (draw(i, true) < std::exp(u.extra_logpost[i] - eq_max)) { eq_w.push_back(eq_max); eq_i.push_back(i); eq_x.push_back(1); }
        eq_done = nd; eq_xdone = nx;
    }

    void posterior_files(const pchip_update &u)
    {
        const int np = nDims + nDer, npars = u.npars;
        const long ntot = u.ndead + u.n_extra;                 // dead points, then phantoms kept by boost_posterior
        auto lp = [&](long i) { return i < u.ndead ? u.logpost[i] : u.extra_logpost[i - u.ndead]; };
        auto rowp = [&](long i) { return i < u.ndead ? u.dead + (size_t)i * npars : u.extra + (size_t)(i - u.ndead) * npars; };
        double mx = -1.7e308;
        for (long i = 0; i < ntot; ++i) if (lp(i) > -1e29) mx = std::max(mx, lp(i));
        FILE *fp = posteriors ? open(path(".txt"), "w") : nullptr;
        FILE *fe = equals ? open(path("_equal_weights.txt"), "w") : nullptr;
        mu.assign(np, 0.0); sig.assign(np, 0.0);
        double sw = 0.0;
        nposterior = 0; nequals = 0;
        std::string tail;
        for (long i = 0; i < ntot; ++i) {
            if (!(lp(i) > -1e29)) continue;               // failed spawns carry no weight
            const double *row = rowp(i);
            const double wgt = std::exp(lp(i) - mx);
            sw += wgt;
            for (int k = 0; k < np; ++k) { mu[k] += wgt * row[k]; sig[k] += wgt * row[k] * row[k]; }
            if (!fp) continue;
            tail = fmt_e24(-2 * row[np + 1]);
            for (int k = 0; k < np; ++k) tail += fmt_e24(row[k]);
            if (fp && wgt > 0.0) { std::fprintf(fp, "%s%s\n", fmt_e24(wgt).c_str(), tail.c_str()); nposterior++; }
        }
        if (fe) {                                              // the list the update rounds left, in its own order
            for (size_t k = 0; k < eq_i.size(); ++k) {
                const double *row = eq_x[k] ? u.extra + (size_t)eq_i[k] * npars : u.dead + (size_t)eq_i[k] * npars;
                tail = fmt_e24(-2 * row[np + 1]);
                for (int c = 0; c < np; ++c) tail += fmt_e24(row[c]);
                std::fprintf(fe, "%s%s\n", fmt_e24(1.0).c_str(), tail.c_str()); nequals++;
            }
        }
        for (int k = 0; k < np; ++k) { mu[k] /= sw; sig[k] = std::sqrt(std::fabs(sig[k] / sw - mu[k] * mu[k])); }
        if (fp) std::fclose(fp);
        if (fe) std::fclose(fe);
    }
    // Per-cluster posteriors (write_posterior_file, read_write.F90:521-592; files clusters/<root>_<rank>.txt, rank 1 =
    // largest evidence).  A cluster's posterior = the points that died in it + the points of every ancestor, scaled
    // by the evidence fractions of the splits in between (add_cluster, run_time_info.f90:432-436,499-502); the
    // largest weight of file k is Z_k / Z.
    void cluster_files(const pchip_update &u)
    {
        const int np = nDims + nDer, K = u.ncluster + u.ncluster_dead;
        std::vector<double> lz(K); std::vector<unsigned> uid(K);
        for (int k = 0; k < u.ncluster; ++k) { lz[k] = u.logZp[k]; uid[k] = u.cluster_uid[k]; }
        for (int k = 0; k < u.ncluster_dead; ++k) { lz[u.ncluster + k] = u.logZp_dead[k]; uid[u.ncluster + k] = u.cluster_uid_dead[k]; }
        std::vector<int> order(K);
        for (int k = 0; k < K; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lz[a] > lz[b]; });
        const long ntot = u.ndead + u.n_extra;
        auto lp = [&](long i) { return i < u.ndead ? u.logpost[i] : u.extra_logpost[i - u.ndead]; };
        auto rowp = [&](long i) { return i < u.ndead ? u.dead + (size_t)i * u.npars : u.extra + (size_t)(i - u.ndead) * u.npars; };
        std::map<unsigned, std::vector<long>> pts;
        for (long i = 0; i < ntot; ++i) if (lp(i) > -1e29) pts[i < u.ndead ? u.dead_cluster[i] : u.extra_cluster[i - u.ndead]].push_back(i);
        std::map<unsigned, std::pair<unsigned, double>> parent;
        for (int j = 0; j < u.nsplit; ++j) parent[u.split_child[j]] = {u.split_parent[j], u.split_logfrac[j]};
        std::string tail;
        for (int r = 0; r < K; ++r) {
            const int k = order[r];
            std::vector<std::pair<unsigned, double>> chain;        // (cluster id, log scale of its points), youngest first
            unsigned cur = uid[k]; double cum = 0.0;
            chain.push_back({cur, 0.0});
            for (auto it = parent.find(cur); it != parent.end(); it = parent.find(cur)) { cum += it->second.second; cur = it->second.first; chain.push_back({cur, cum}); }
            double mx = -1.7e308;
            for (auto &c : chain) for (long i : pts[c.first]) mx = std::max(mx, lp(i) + c.second);
            char num[32]; std::snprintf(num, sizeof num, "%d", r + 1);
            const std::string stem = base + "/clusters/" + root + "_" + num;
            FILE *fp = posteriors ? std::fopen((stem + ".txt").c_str(), "w") : nullptr;
            FILE *fe = equals ? std::fopen((stem + "_equal_weights.txt").c_str(), "w") : nullptr;
            if ((posteriors && !fp) || (equals && !fe)) halt_program(("polychord_hip: cannot write " + stem + " (does " + base + "/clusters exist?)").c_str());
            const double frac = std::exp(lz[k] - u.logZ);
            for (auto c = chain.rbegin(); c != chain.rend(); ++c)      // ancestors first, as the copies were made
                for (long i : pts[c->first]) {
                    const double *row = rowp(i);
                    const double rel = std::exp(lp(i) + c->second - mx);
                    if (!(rel > 0.0)) continue;
                    tail = fmt_e24(-2 * row[np + 1]);
                    for (int q = 0; q < np; ++q) tail += fmt_e24(row[q]);
                    if (fp) std::fprintf(fp, "%s%s\n", fmt_e24(rel * frac).c_str(), tail.c_str());
                    if (fe && polychord_hip_keyed_uniform(seed, 7u, (unsigned)r, 0u, (unsigned)i) < rel) std::fprintf(fe, "%s%s\n", fmt_e24(frac).c_str(), tail.c_str());
                }
            if (fp) std::fclose(fp);
            if (fe) std::fclose(fe);
        }
    }
    void update(const pchip_update &u)
    {
        if (u.final_call == 2) {                  // write_prior_file (read_write.F90:721-752) + generate.F90:274-279
            if (!write_prior) return;
            const int np = nDims + nDer;
            FILE *f = open(path("_prior.txt"), "w");
            std::string line;
            for (int i = 0; i < u.nlive; ++i) {
                const double *r = u.live + (size_t)i * u.npars;
                line = fmt_e24(1.0) + fmt_e24(-2 * r[np + 1]);
                for (int k = 0; k < np; ++k) line += fmt_e24(r[k]);
                std::fprintf(f, "%s\n", line.c_str());
            }
            std::fclose(f);
            f = open(path(".prior_info"), "w");
            std::fprintf(f, "nprior = %12d\nndiscarded = %12ld\n", u.nlive, u.ndiscarded);
            std::fclose(f);
            return;
        }
        if (write_dead && (u.ndead > dead_written || dead_written == 0)) {
            FILE *f1 = open(path("_dead.txt"), dead_written ? "a" : "w"), *f2 = open(path("_dead-birth.txt"), dead_written ? "a" : "w");
            rows(f1, u.dead, dead_written, u.ndead, u.npars, true, false);
            rows(f2, u.dead, dead_written, u.ndead, u.npars, false, true);
            std::fclose(f1); std::fclose(f2);
            dead_written = u.ndead;
        }
        if (write_live) {
            FILE *f1 = open(path("_phys_live.txt"), "w"), *f2 = open(path("_phys_live-birth.txt"), "w");
            rows(f1, u.live, 0, u.nlive, u.npars, false, false);
            rows(f2, u.live, 0, u.nlive, u.npars, false, true);
            std::fclose(f1); std::fclose(f2);
        }
        if (equals && u.final_call != 2) thin_equals_round(u);
        if (u.final_call == 1 && (posteriors || equals)) { posterior_files(u); if (cluster_posteriors) cluster_files(u); }
        if (write_stats) stats(u);
        if (feedback >= 1 && u.final_call == 0) progress(u);
        nlike_last = u.nlike;
        nlike_last_g.assign((size_t)(u.ngrade > 0 ? u.ngrade : 1), 0L);
        for (int g = 0; g < (int)nlike_last_g.size(); ++g) nlike_last_g[g] = u.nlike_grade ? u.nlike_grade[g] : u.nlike;
    }
    static void hook(void *user, const pchip_update *u) { ((FileSink *)user)->update(*u); }
};

}  // namespace

extern "C" {

double polychord_hip_gaussian(double *th, int D, double *phi, int nDer)
{   // likelihoods/examples/gaussian.f90:12-41
    double s = 0, r2 = 0;
    for (int d = 0; d < D; ++d) { const double z = (th[d] - G.g_mu) / G.g_sigma; s += z * z; r2 += (th[d] - G.g_mu) * (th[d] - G.g_mu); }
    if (nDer >= 1) phi[0] = std::sqrt(r2);
    if (nDer >= 2) phi[1] = std::log(std::pow(phi[0], (double)D) * std::pow(std::sqrt(3.14159265358979323846), (double)D) / std::tgamma(1.0 + D / 2.0));
    return -(double)D * (std::log(G.g_sigma) + LOG_TWO_PI / 2.0) - s / 2.0;
}
double polychord_hip_rastrigin(double *th, int D, double *, int)
{   // likelihoods/examples/rastrigin.f90:20-35
    double s = 0;
    for (int d = 0; d < D; ++d) s += std::log(4991.21750) + th[d] * th[d] - 10.0 * std::cos(6.283185307179586 * th[d]);
    return -s;
}
double polychord_hip_twin_gaussian(double *th, int D, double *phi, int nDer)
{   // likelihoods/examples/twin_gaussian.f90:14-56
    const double sg = G.g_sigma;
    const double n = -(double)D * (std::log(sg) + LOG_TWO_PI / 2.0);
    double s1 = 0, s2 = 0;
    for (int d = 0; d < D; ++d) {
        const double m1 = d < 2 ? -0.5 : 0.0, m2 = d < 2 ? 0.5 : 0.0, z1 = (th[d] - m1) / sg, z2 = (th[d] - m2) / sg;
        s1 += z1 * z1; s2 += z2 * z2;
    }
    if (nDer >= 1) phi[0] = th[0] > 0.5 ? 1.0 : -1.0;
    const double a = n - s1 / 2, b = n - s2 / 2;
    return (a > b ? a + std::log(std::exp(b - a) + 1) : b + std::log(std::exp(a - b) + 1)) - std::log(2.0);
}
double polychord_hip_corr_gaussian(double *th, int D, double *, int)
{   // likelihoods/examples/random_gaussian.f90:17-30, utils.F90:1028-1048
    if (G.cg_D != D) halt_program("polychord_hip: polychord_hip_set_corr_gaussian was not called for this nDims");
    double q = 0;
    for (int a = 0; a < D; ++a) {
        double t = 0;
        for (int b = 0; b < D; ++b) t += G.cg_invcov[(size_t)a * D + b] * (th[b] - G.cg_mean[b]);
        q += (th[a] - G.cg_mean[a]) * t;
    }
    return -((double)D * LOG_TWO_PI + G.cg_logdet) / 2.0 - q / 2.0;
}
void polychord_hip_set_gaussian(double mu, double sigma) { G.g_mu = mu; G.g_sigma = sigma; }
void polychord_hip_set_corr_gaussian(int D, const double *invcov, const double *mean, double logdet)
{
    G.cg_D = D; G.cg_invcov.assign(invcov, invcov + (size_t)D * D); G.cg_mean.assign(mean, mean + D); G.cg_logdet = logdet;
}
void polychord_hip_uniform_prior(double *cube, double *theta, int D)
{   // priors.f90:40-55
    for (int d = 0; d < D; ++d) {
        const double lo = d < G.up_D ? G.up_lo[d] : 0.0, hi = d < G.up_D ? G.up_hi[d] : 1.0;
        theta[d] = lo + (hi - lo) * cube[d];
    }
}
void polychord_hip_set_uniform_prior(int D, const double *lo, const double *hi)
{
    G.up_D = D; G.up_lo.assign(lo, lo + D); G.up_hi.assign(hi, hi + D);
}
void polychord_hip_set_option(const char *name, double value)
{
    if (!std::strcmp(name, "batch")) G.batch = (int)value;
    else if (!std::strcmp(name, "device")) G.device = (int)value;
    else if (!std::strcmp(name, "inject_fault")) pchip_inject_fault((int)value);
    else if (!std::strcmp(name, "cluster_capacity")) pchip_set_capacity((int)value, -1);
    else if (!std::strcmp(name, "phantom_capacity")) pchip_set_capacity(-1, (int)value);
    else if (!std::strcmp(name, "trim_cache")) pchip_trim_cache();
    else if (!std::strcmp(name, "halt_returns")) G.halt_returns = value != 0.0;
    else if (!std::strcmp(name, "epoch_discard")) G.epoch_discard = value != 0.0 ? 1 : 0;
    else std::fprintf(stderr, "polychord_hip: unknown option %s\n", name);
}

int pchip_abi_version(void) { return PCHIP_ABI_VERSION; }
unsigned long pchip_sizeof(const char *n)
{
    if (!n) return 0;
    if (!std::strcmp(n, "settings")) return sizeof(pchip_settings);
    if (!std::strcmp(n, "result")) return sizeof(pchip_result);
    if (!std::strcmp(n, "merged")) return sizeof(pchip_merged);
    if (!std::strcmp(n, "like")) return sizeof(pchip_like);
    if (!std::strcmp(n, "prior")) return sizeof(pchip_prior);
    if (!std::strcmp(n, "update")) return sizeof(pchip_update);
    return 0;
}

// Files of a merged result (pchip_merge_records / pchip_run_repeats) in the reference's formats: <root>.stats
// (read_write.F90:809-910; no local evidences: the union has one volume), <root>_dead-birth.txt (theta, phi, logL, birth;
// :707-716) and <root>.txt (weight, -2 logL, theta, phi; :479-617, weights relative to the largest).
int pchip_merged_write(const pchip_merged *m, int nDims, int nDerived, const char *base_dir, const char *file_root)
{
    if (!m || !m->rows || m->n <= 0) { std::fprintf(stderr, "polychord_hip: merged result without rows\n"); return 1; }
    const std::string stem = std::string(base_dir ? base_dir : "chains") + "/" + (file_root ? file_root : "test");
    const int np = nDims + nDerived, nT = m->nTotal, p0 = nDims, b0 = 2 * nDims + nDerived, l0 = b0 + 1;
    FILE *fd = std::fopen((stem + "_dead-birth.txt").c_str(), "w"), *fp = std::fopen((stem + ".txt").c_str(), "w");
    FILE *fs = std::fopen((stem + ".stats").c_str(), "w");
    if (!fd || !fp || !fs) { if (fd) std::fclose(fd); if (fp) std::fclose(fp); if (fs) std::fclose(fs); std::fprintf(stderr, "PolyChord Error: cannot write %s.*\n", stem.c_str()); return 2; }
    double mx = -1.7e308;
    for (long i = 0; i < m->n; ++i) mx = std::max(mx, m->logweights[i] + m->rows[(size_t)i * nT + l0]);
    std::string line;
    long nposterior = 0;
    for (long i = 0; i < m->n; ++i) {
        const double *r = m->rows + (size_t)i * nT;
        line.clear();
        for (int k = 0; k < np; ++k) line += fmt_e24(r[p0 + k]);
        line += fmt_e24(r[l0]); line += fmt_e24(r[b0]); line += '\n';
        std::fwrite(line.data(), 1, line.size(), fd);
        const double w = std::exp(m->logweights[i] + r[l0] - mx);
        if (w > 0.0) {
            line = fmt_e24(w) + fmt_e24(-2 * r[l0]);
            for (int k = 0; k < np; ++k) line += fmt_e24(r[p0 + k]);
            line += '\n';
            std::fwrite(line.data(), 1, line.size(), fp);
            nposterior++;
        }
    }
    std::fclose(fd); std::fclose(fp);
    std::fprintf(fs, "Evidence estimates:\n===================\n");
    std::fprintf(fs, "  - The evidence Z is a log-normally distributed, with location and scale parameters mu and sigma.\n");
    std::fprintf(fs, "  - We denote this as log(Z) = mu +/- sigma.\n\nGlobal evidence:\n----------------\n\n");
    std::fprintf(fs, "log(Z)       = %s +/- %s\n\n\n", fmt_e24(m->logZ).c_str(), fmt_e24(std::sqrt(std::fabs(m->varlogZ))).c_str());
    std::fprintf(fs, "Local evidences:\n----------------\n\n");
    std::fprintf(fs, "\n\nRun-time information:\n---------------------\n\n");
    std::fprintf(fs, " ncluster:   %8d /%8d\n", 0, m->nruns);
    std::fprintf(fs, " nposterior: %8ld\n", nposterior);
    std::fprintf(fs, " nequals:    %8d\n", 0);
    std::fprintf(fs, " ndead:      %8ld\n", m->n);
    std::fprintf(fs, " nlive:      %8d\n", 0);
    std::fprintf(fs, " nlike:      %8ld\n", m->nlike);
    std::fprintf(fs, " <nlike>:    %8.2f   (%8.2f per slice )\n", 0.0, 0.0);
    std::fprintf(fs, "\n\nDim No.       Mean        Sigma\n");
    for (int k = 0; k < np; ++k) {
        if (k == nDims) std::fprintf(fs, "-------------------------------\n");
        std::fprintf(fs, "%3d%s +/- %s\n", k + 1, fmt_e24(m->post_mean[k]).c_str(), fmt_e24(std::sqrt(std::fabs(m->post_var[k]))).c_str());
    }
    if (nDerived == 0) std::fprintf(fs, "-------------------------------\n");
    // behind everything the reference's readers look at (pypolychord/output.py:57-99 stops at <nlike>): what kind of evidence this is
    std::fprintf(fs, "\n\nUnion of %d independent runs:\n-----------------------------\n\n", m->nruns);
    if (m->evidence_rule == 1)
        std::fprintf(fs, " evidence rule 1: %d of the runs held more than one cluster; the global evidence above is the mean of the runs' own Z\n"
                         "   (log-normal moments, error = the larger of the propagated one and the scatter between runs), posterior weights are the\n"
                         "   runs' own over the number of runs.  A replay of the union from ranks and live counts does not know the clusters' volumes:\n", m->nclustered);
    else
        std::fprintf(fs, " evidence rule 0: no run ever held more than one cluster; the global evidence above is the replay of the union from ranks and live counts\n");
    std::fprintf(fs, "   replay of the union: %s +/- %s\n", fmt_e24(m->logZ_replay).c_str(), fmt_e24(std::sqrt(std::fabs(m->varlogZ_replay))).c_str());
    std::fprintf(fs, "   mean of the runs' own log Z: %s +/- %s\n", fmt_e24(m->runs_logZ_mean).c_str(), fmt_e24(m->runs_logZ_sem).c_str());
    std::fclose(fs);
    return 0;
}

// .resume files without a run: parse `in` and write it back to `out` (NULL: only parse).  counts[0..5] =
// nDims, nDerived, ndead, ncluster, ncluster_dead, live points in total.  Returns 0 on success.
int polychord_hip_resume_copy(const char *in, const char *out, int *counts)
{
    PcResume r; std::string err;
    if (!pc_resume_read(in, r, err)) { std::fprintf(stderr, "polychord_hip: %s\n", err.c_str()); return 1; }
    if (counts) {
        int nl = 0;
        for (int v : r.nlive) nl += v;
        counts[0] = r.nDims; counts[1] = r.nDerived; counts[2] = r.ndead; counts[3] = r.ncluster; counts[4] = r.ncluster_dead; counts[5] = nl;
    }
    if (out && !pc_resume_write(out, r, -1e30, err)) { std::fprintf(stderr, "polychord_hip: %s\n", err.c_str()); return 2; }
    return 0;
}

// time_speeds (generate.F90:330-455) for this engine: seconds per likelihood call of every grade.  The likelihoods that
// run inside the slice kernel cost the same whichever parameters moved (speed ratio 1).  A host callback is timed:
// grade 1 = a call after all parameters changed, grade g = a call after only the parameters of grades >= g changed
// (callers that cache on the slow parameters, e.g. cobaya, return faster then).
static std::vector<double> time_speeds_host(polychord_loglike_fn like, polychord_prior_fn prior, int D, int nDer, int nGrade,
                                            const int *grade_dims, double logzero, unsigned seed, int feedback)
{
    std::vector<double> speed(nGrade, 1.0);
    const bool device_like = like == polychord_hip_gaussian || like == polychord_hip_rastrigin ||
                             like == polychord_hip_twin_gaussian || like == polychord_hip_corr_gaussian;
    if (device_like) return speed;
    std::vector<double> cube(D), theta(D), phi(std::max(1, nDer));
    unsigned long long n = 0;
    auto draw = [&](int from) { for (int d = from; d < D; ++d) cube[d] = polychord_hip_keyed_uniform(seed, 8u /* timing stream */, 0u, 0u, (unsigned)(n++)); };
    auto eval = [&]() {
        if (prior && prior != polychord_hip_uniform_prior) prior(cube.data(), theta.data(), D); else polychord_hip_uniform_prior(cube.data(), theta.data(), D);
        return like(theta.data(), D, phi.data(), nDer);
    };
    using clk = std::chrono::steady_clock;
    auto timed = [&](int from, int ncalls) {
        double tot = 0.0; int ok = 0;
        for (int tries = 0; ok < ncalls && tries < 50 * ncalls; ++tries) {
            draw(from);
            const auto t0 = clk::now();
            const double l = eval();
            const double dt = std::chrono::duration<double>(clk::now() - t0).count();
            if (l > logzero) { tot += dt; ok++; } else if (from > 0) draw(0);
        }
        return ok > 0 ? tot / ok : 1.0;
    };
    draw(0);
    speed[0] = std::max(1e-9, timed(0, 8));
    for (int g = 1, off = 0; g < nGrade; ++g) {
        off += grade_dims[g - 1];
        speed[g] = std::max(1e-9, timed(off, 16));
        if (feedback >= 1) std::printf("Speed %2d = %10.3E seconds\n", g + 1, speed[g]);
    }
    return speed;
}

static void c_interface_impl(
    polychord_loglike_fn loglikelihood, polychord_prior_fn prior, polychord_dumper_fn dumper,
    int nlive, int num_repeats, int nprior, int nfail, bool do_clustering, int feedback,
    double precision_criterion, double logzero, int max_ndead, double boost_posterior,
    bool posteriors, bool equals, bool cluster_posteriors, bool write_resume, bool write_paramnames,
    bool read_resume, bool write_stats_f, bool write_live, bool write_dead, bool write_prior,
    bool maximise, double compression_factor, bool synchronous, int nDims, int nDerived,
    char *base_dir, char *file_root, int nGrade, double *grade_frac, int *grade_dims, int n_nlives,
    double *loglikes, int *nlives, int seed, int *comm);

const char *polychord_hip_last_error(void) { return G.last_error.empty() ? nullptr : G.last_error.c_str(); }

void polychord_c_interface(
    polychord_loglike_fn loglikelihood, polychord_prior_fn prior, polychord_dumper_fn dumper,
    int nlive, int num_repeats, int nprior, int nfail, bool do_clustering, int feedback,
    double precision_criterion, double logzero, int max_ndead, double boost_posterior,
    bool posteriors, bool equals, bool cluster_posteriors, bool write_resume, bool write_paramnames,
    bool read_resume, bool write_stats_f, bool write_live, bool write_dead, bool write_prior,
    bool maximise, double compression_factor, bool synchronous, int nDims, int nDerived,
    char *base_dir, char *file_root, int nGrade, double *grade_frac, int *grade_dims, int n_nlives,
    double *loglikes, int *nlives, int seed, int *comm)
{
    G.last_error.clear();
    try {
        c_interface_impl(loglikelihood, prior, dumper, nlive, num_repeats, nprior, nfail, do_clustering, feedback,
                         precision_criterion, logzero, max_ndead, boost_posterior, posteriors, equals, cluster_posteriors,
                         write_resume, write_paramnames, read_resume, write_stats_f, write_live, write_dead, write_prior,
                         maximise, compression_factor, synchronous, nDims, nDerived, base_dir, file_root, nGrade, grade_frac,
                         grade_dims, n_nlives, loglikes, nlives, seed, comm);
    } catch (const HaltRequest &h) { G.last_error = h.msg; }      // only in "halt_returns" mode: halt_program exits otherwise
}

static void c_interface_impl(
    polychord_loglike_fn loglikelihood, polychord_prior_fn prior, polychord_dumper_fn dumper,
    int nlive, int num_repeats, int nprior, int nfail, bool do_clustering, int feedback,
    double precision_criterion, double logzero, int max_ndead, double boost_posterior,
    bool posteriors, bool equals, bool cluster_posteriors, bool write_resume, bool write_paramnames,
    bool read_resume, bool write_stats_f, bool write_live, bool write_dead, bool write_prior,
    bool maximise, double compression_factor, bool synchronous, int nDims, int nDerived,
    char *base_dir, char *file_root, int nGrade, double *grade_frac, int *grade_dims, int n_nlives,
    double *loglikes, int *nlives, int seed, int *comm)
{
    (void)synchronous; (void)comm;
    if (num_repeats < 1) halt_program("You need to set num_repeats. Suggestion: 5*nDims");     // settings.f90:216
    pchip_settings s;
    pchip_settings_default(&s, nDims, nDerived);
    // fast/slow parameter grades.  generate.F90:303-309: if every grade_frac exceeds 1 they ARE the repeats per grade;
    // otherwise the first grade keeps num_repeats and grade g gets nint(frac_g / frac_1 * num_repeats * speed_1 / speed_g),
    // the speeds being wall-clock times per likelihood call (time_speeds, generate.F90:330-455).
    std::vector<int> g_reps;
    if (nGrade > 1 && grade_dims && grade_frac) {
        if (nGrade > 8) halt_program("polychord_hip: at most 8 parameter grades");
        int tot = 0;
        for (int g = 0; g < nGrade; ++g) { if (grade_dims[g] < 1) halt_program("polychord_hip: every grade needs at least one parameter"); tot += grade_dims[g]; }
        if (tot != nDims) halt_program("polychord_hip: grade_dims must sum to nDims");
        bool explicit_reps = true;
        for (int g = 0; g < nGrade; ++g) explicit_reps = explicit_reps && grade_frac[g] > 1.0;
        g_reps.assign(nGrade, num_repeats);
        if (explicit_reps) for (int g = 0; g < nGrade; ++g) g_reps[g] = (int)grade_frac[g];
        else {
            std::vector<double> speed = time_speeds_host(loglikelihood, prior, nDims, nDerived, nGrade, grade_dims, logzero, (unsigned)(seed >= 0 ? seed : 1), feedback);
            for (int g = 1; g < nGrade; ++g)
                g_reps[g] = std::max(1, (int)std::lround(grade_frac[g] / grade_frac[0] * num_repeats * speed[0] / speed[g]));
        }
        s.nGrade = nGrade; s.grade_dims = grade_dims; s.grade_repeats = g_reps.data();
    } else if (nGrade == 1 && grade_dims && grade_dims[0] != nDims) halt_program("polychord_hip: grade_dims must sum to nDims");
    // one grade whose grade_frac exceeds 1: the reference takes it as THE number of repeats (generate.F90:303-309,
    // RTI%num_repeats = int(grade_frac) when no entry is <= 1), whatever num_repeats says
    if (nGrade == 1 && grade_frac && grade_frac[0] > 1.0) num_repeats = (int)grade_frac[0];
    s.nlive = nlive; s.num_repeats = num_repeats; s.nprior = nprior; s.nfail = nfail; s.do_clustering = do_clustering;
    s.feedback = feedback; s.precision_criterion = precision_criterion; s.logzero = logzero; s.max_ndead = max_ndead;
    s.boost_posterior = boost_posterior; s.posteriors = posteriors; s.equals = equals; s.cluster_posteriors = cluster_posteriors;
    s.compression_factor = compression_factor; s.n_nlives = n_nlives; s.loglikes = loglikes; s.nlives = nlives;
    s.seed = seed >= 0 ? seed : (int)(std::chrono::system_clock::now().time_since_epoch().count() & 0x7fffffff); // random_utils.F90:62-79
    s.batch = G.batch; s.device = G.device; s.epoch_discard = G.epoch_discard;
    // tests: the engine's sequential-stream mode through the reference's entry point (draw order of the reference binary;
    // with the RNG shim of oracle/ the two programs then write the same files)
    if (const char *e = std::getenv("PC_SEQUENTIAL_RNG")) s.sequential_rng = std::atoi(e) != 0;
    const std::string resume_path = std::string(base_dir ? base_dir : "chains") + "/" + (file_root ? file_root : "test") + ".resume";   // read_write.F90:1040-1062
    if (write_resume) s.resume_write = resume_path.c_str();
    if (read_resume) s.resume_read = resume_path.c_str();
    pchip_like L{}; pchip_prior P{};
    if (loglikelihood == polychord_hip_gaussian) { L.kind = PCHIP_LIKE_GAUSSIAN; L.mu = G.g_mu; L.sigma = G.g_sigma; }
    else if (loglikelihood == polychord_hip_rastrigin) L.kind = PCHIP_LIKE_RASTRIGIN;
    else if (loglikelihood == polychord_hip_twin_gaussian) { L.kind = PCHIP_LIKE_TWIN_GAUSSIAN; L.sigma = G.g_sigma; }
    else if (loglikelihood == polychord_hip_corr_gaussian) {
        if (G.cg_D != nDims) halt_program("polychord_hip: polychord_hip_set_corr_gaussian was not called for this nDims");
        L.kind = PCHIP_LIKE_CORR_GAUSSIAN; L.invcov = G.cg_invcov.data(); L.mean = G.cg_mean.data(); L.logdetcov = G.cg_logdet;
    } else { L.kind = PCHIP_LIKE_CALLBACK; L.fn = loglikelihood; }
    if (prior == polychord_hip_uniform_prior) {
        P.kind = 1;
        if (G.up_D == nDims) { P.lo = G.up_lo.data(); P.hi = G.up_hi.data(); }
    } else { P.kind = 0; P.fn = prior; }
    const std::string base = base_dir ? base_dir : "chains", root = file_root ? file_root : "test";
    if (write_stats_f || write_dead || write_live || posteriors || equals || write_prior || write_resume) {
        struct stat sb;
        if (stat(base.c_str(), &sb) != 0) halt_program(("PolyChord Error: " + base + " does not exist").c_str()); // read_write.F90:28-38
    }
    if (write_paramnames) {
        // interfaces.F90:424-428 + ini.f90:98-121: theta<i> / phi<i> with LaTeX labels; read_write.F90:964-1014
        FILE *f = std::fopen((base + "/" + root + ".paramnames").c_str(), "w");
        if (!f) halt_program(("PolyChord Error: " + base + " does not exist").c_str());
        for (int i = 1; i <= nDims; ++i) std::fprintf(f, "theta%d      \\theta_{%d}\n", i, i);
        for (int i = 1; i <= nDerived; ++i) std::fprintf(f, "phi%d      \\phi_{%d}\n", i, i);
        std::fclose(f);
        f = std::fopen((base + "/" + root + ".properties.ini").c_str(), "w");
        if (f) { std::fprintf(f, "sampler=nested\nlabel=%s\n", root.c_str()); std::fclose(f); }
    }
    if (P.kind == 0 && L.kind != PCHIP_LIKE_CALLBACK) {
        // a user prior with a built-in likelihood: evaluate both on the host (the device still proposes)
        L.kind = PCHIP_LIKE_CALLBACK; L.fn = loglikelihood;
    }
    FileSink sink;
    sink.base = base; sink.root = root; sink.nDims = nDims; sink.nDer = nDerived;
    sink.write_stats = write_stats_f; sink.write_live = write_live; sink.write_dead = write_dead;
    sink.posteriors = posteriors; sink.equals = equals; sink.write_prior = write_prior; sink.cluster_posteriors = cluster_posteriors; sink.seed = (unsigned)s.seed; sink.logzero = logzero;
    sink.compression = compression_factor; sink.num_repeats = num_repeats;
    sink.epoch_note = (do_clustering && s.batch != 1 && !s.sequential_rng) ? s.epoch_discard : -1;
    sink.feedback = feedback; sink.nlive_set = nlive > 0 ? nlive : 1;
    const bool files = write_stats_f || write_dead || write_live || posteriors || equals || write_prior;
    pchip_result r;
    pchip_hooks hooks{dumper, (files || feedback >= 1) ? FileSink::hook : nullptr, &sink};
    if (feedback >= 1) {   // feedback.f90:19-63, 79-91, 189-218 in a few lines
        std::printf("\nPolyChord interface on polychord_hip (MI355X engine)\n");
        std::printf("nlive      : %8d\nnDims      : %8d\nnDerived   : %8d\n", nlive, nDims, nDerived);
        if (do_clustering) std::printf("Doing Clustering\n");
        if (do_clustering && s.batch != 1)
            std::printf("chains in flight when the list of clusters changes: %s\n",
                        s.epoch_discard ? "all discarded (epoch_discard = 1: the reference farm's rule, nested_sampling.F90:313)"
                                        : "only those seeded in the cluster that ended are lost (epoch_discard = 0, this engine's default; option \"epoch_discard\" = 1: the reference farm's rule)");
        if (write_resume || read_resume) std::printf("Resume file: %s/%s.resume\n", base.c_str(), root.c_str());
        std::printf("\nnum_repeats:");
        if (g_reps.size()) for (int v : g_reps) std::printf("%8d", v); else std::printf("%8d", num_repeats);
        std::printf("\nstarted sampling\n\n");
        std::fflush(stdout);
    }
    const int rc = pchip_run_hooks(&s, &L, &P, &hooks, &r);
    if (rc == 5) { pchip_result_free(&r); return; }           // stopped by a binding (callback raised)
    if (rc != 0) { pchip_result_free(&r); halt_program(("polychord_hip: engine failure (code " + std::to_string(rc) + ", message above)").c_str()); }
    if (maximise && r.nlive_final > 0) {
        // nested_sampling.F90:379: polish the best live points of the set at termination (host side, pc_maximise.hip)
        const std::string mpath = base + "/" + root + ".maximum";
        if (pchip_maximise(loglikelihood, prior, nDims, nDerived, logzero, r.live, r.live_cluster, r.nlive_final,
                           posteriors ? r.post_mean : nullptr, mpath.c_str()) == 2)
            halt_program(("PolyChord Error: " + base + " does not exist").c_str());
    }
    if (feedback >= 0) {   // feedback.f90:320-339
        std::printf(" ____________________________________________________ \n|                                                    |\n");
        std::printf("| ndead  = %12ld                              |\n", r.ndead);
        std::printf("| log(Z) = %18.5f +/- %18.5f |\n", r.logZ, std::sqrt(std::fabs(r.varlogZ)));
        std::printf("|____________________________________________________|\n");
        // (behind the reference's box, for the two things a caller of the drop-in cannot see from its numbers)
        if (r.ncluster_peak > 1) {
            std::printf("polychord_hip: up to %d clusters were alive at once.  The error above is this run's own estimate (run_time_info.f90:652-678);\n"
                        "               WHICH modes a run finds adds run-to-run scatter it does not contain -- measured 4 x the reported error on a 10-D\n"
                        "               Rastrigin, for the reference binary as for this engine.  Repeats with different seeds, combined, carry it\n"
                        "               (pchip_run_repeats / polychordlite_amd.repeats.run_repeats).\n", r.ncluster_peak);
            if (r.batch != 1 && !s.sequential_rng)
                std::printf("polychord_hip: chains in flight when the list of clusters changed: %s\n",
                            r.epoch_discard ? "all discarded -- the reference farm's rule (nested_sampling.F90:313)"
                                            : "only those seeded in the cluster that ended were lost -- THIS ENGINE's rule, not the reference farm's\n"
                                              "               (nested_sampling.F90:313 discards them all; polychord_hip_set_option(\"epoch_discard\", 1) selects that: same\n"
                                              "               distribution of results, 1.7 x the likelihood evaluations at 10-D Rastrigin)");
        }
    }
    if (feedback >= 1) {
        std::printf("polychord_hip: nlike = %ld  (%.3f s on the device, %d chains per nursery)\n", r.nlike, r.t_total, r.batch);
    }
    std::fflush(stdout);
    pchip_result_free(&r);
}

}  // extern "C"
