// pc_resume.hip -- reader / writer of the reference's .resume grammar (see pc_resume.h).  Host code only.
#include "pc_resume.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <fstream>
#include <sstream>

namespace {

// Fortran E24.15E3 (utils.F90:19-21), e.g. "   0.626931681801488E-001"
void put_e24(std::string &out, double v)
{
    char buf[64];
    if (v == 0.0) { out += "   0.000000000000000E+000"; return; }
    if (!std::isfinite(v)) { std::snprintf(buf, sizeof buf, "%24s", std::isnan(v) ? "NaN" : (v > 0 ? "Infinity" : "-Infinity")); out += buf; return; }
    char t[64];
    std::snprintf(t, sizeof t, "%.14E", std::fabs(v));
    const char *e = std::strchr(t, 'E');
    const int ex = std::atoi(e + 1) + 1;
    char m[32];
    m[0] = t[0]; std::memcpy(m + 1, t + 2, 14); m[15] = 0;
    std::snprintf(buf, sizeof buf, "%s0.%sE%c%03d", v < 0 ? "-" : "", m, ex < 0 ? '-' : '+', std::abs(ex));
    char pad[64];
    std::snprintf(pad, sizeof pad, "%24s", buf);
    out += pad;
}

struct Writer {
    FILE *f;
    std::string line;
    void head(const char *s) { std::fprintf(f, "%s\n", s); }
    void sep() { std::fprintf(f, "---------------------------------------\n"); }
    template <class T> void ints(const T *a, size_t n)
    {
        if (!n) return;
        for (size_t i = 0; i < n; ++i) std::fprintf(f, "%12lld", (long long)a[i]);
        std::fputc('\n', f);
    }
    void reals(const double *a, size_t n)
    {
        if (!n) return;
        line.clear();
        for (size_t i = 0; i < n; ++i) put_e24(line, a[i]);
        line += '\n';
        std::fwrite(line.data(), 1, line.size(), f);
    }
};

struct Section { std::string name; std::vector<std::string> lines; };   // separator lines are kept as "---"

bool is_head(const std::string &l) { return l.compare(0, 3, "===") == 0; }
bool is_sep(const std::string &l) { return l.compare(0, 3, "---") == 0; }

void numbers(const std::string &l, std::vector<double> &out)
{
    const char *p = l.c_str();
    char *e;
    for (;;) {
        const double v = std::strtod(p, &e);
        if (e == p) break;
        out.push_back(v); p = e;
    }
}

struct Reader {
    std::vector<Section> secs;
    size_t at = 0;
    std::string err;
    const Section *next(const char *key)
    {
        if (at >= secs.size()) { err = std::string("missing section: ") + key; return nullptr; }
        if (secs[at].name.find(key) == std::string::npos) { err = "expected section '" + std::string(key) + "', found '" + secs[at].name + "'"; return nullptr; }
        return &secs[at++];
    }
    bool flat(const char *key, std::vector<double> &v)
    {
        const Section *s = next(key);
        if (!s) return false;
        v.clear();
        for (auto &l : s->lines) if (!is_sep(l)) numbers(l, v);
        return true;
    }
    bool ints(const char *key, std::vector<int> &v)
    {
        std::vector<double> d;
        if (!flat(key, d)) return false;
        v.assign(d.size(), 0);
        for (size_t i = 0; i < d.size(); ++i) v[i] = (int)std::llround(d[i]);
        return true;
    }
    bool one_int(const char *key, int &v)
    {
        std::vector<int> t;
        if (!ints(key, t)) return false;
        if (t.size() != 1) { err = std::string("one integer expected in ") + key; return false; }
        v = t[0]; return true;
    }
    bool one_real(const char *key, double &v)
    {
        std::vector<double> t;
        if (!flat(key, t)) return false;
        if (t.size() != 1) { err = std::string("one real expected in ") + key; return false; }
        v = t[0]; return true;
    }
    // blocks introduced by separator lines: block b holds the numbers of its lines, concatenated
    bool blocks(const char *key, std::vector<std::vector<double>> &b)
    {
        const Section *s = next(key);
        if (!s) return false;
        b.clear();
        for (auto &l : s->lines) {
            if (is_sep(l)) { b.emplace_back(); continue; }
            if (b.empty()) b.emplace_back();
            numbers(l, b.back());
        }
        return true;
    }
};

}  // namespace

bool pc_resume_write(const std::string &path, const PcResume &r, double logzero, std::string &err)
{
    const std::string tmp = path + "_new";                     // the reference writes to a temporary, then renames
    FILE *f = std::fopen(tmp.c_str(), "w");
    if (!f) { err = "cannot open " + tmp; return false; }
    Writer w{f, {}};
    const int nc = r.ncluster, ncd = r.ncluster_dead, D = r.nDims, nT = r.nTotal(), zero = 0;
    std::vector<int> zeros_nc(nc, 0), zeros_ncd(ncd, 0);
    std::vector<double> lz_nc(nc, logzero), lz_ncd(ncd, logzero);
    w.head("=== Number of dimensions ===");                                     w.ints(&r.nDims, 1);
    w.head("=== Number of derived parameters ===");                             w.ints(&r.nDerived, 1);
    w.head("=== Number of dead points/iterations ===");                         w.ints(&r.ndead, 1);
    w.head("=== Number of clusters ===");                                       w.ints(&nc, 1);
    w.head("=== Number of dead clusters ===");                                  w.ints(&ncd, 1);
    w.head("=== Number of global weighted posterior points ===");               w.ints(&zero, 1);
    w.head("=== Number of global equally weighted posterior points ===");       w.ints(&zero, 1);
    const int ng = (int)r.grade_dims.size();
    w.head("=== Number of grades ===");                                         w.ints(&ng, 1);
    w.head("=== positions of grades ===");                                      w.ints(r.grade_dims.data(), r.grade_dims.size());
    w.head("=== Number of repeats ===");                                        w.ints(r.num_repeats.data(), r.num_repeats.size());
    w.head("=== Number of likelihood calls ===");                               w.ints(r.nlike.data(), r.nlike.size());
    w.head("=== Number of live points in each cluster ===");                    w.ints(r.nlive.data(), nc);
    w.head("=== Number of phantom points in each cluster ===");                 w.ints(r.nphantom.data(), nc);
    w.head("=== Number of weighted posterior points in each cluster ===");      w.ints(zeros_nc.data(), nc);
    w.head("=== Number of equally weighted posterior points in each cluster ==="); w.ints(zeros_nc.data(), nc);
    w.head("=== Minimum loglikelihood positions ===");                          w.ints(r.imin.data(), nc);
    w.head("=== Number of weighted posterior points in each dead cluster ==="); w.ints(zeros_ncd.data(), ncd);
    w.head("=== Number of equally weighted posterior points in each dead cluster ==="); w.ints(zeros_ncd.data(), ncd);
    w.head("=== global evidence -- log(<Z>) ===");                              w.reals(&r.logZ, 1);
    w.head("=== global evidence^2 -- log(<Z^2>) ===");                          w.reals(&r.logZ2, 1);
    w.head("=== posterior thin factor ===");                                    w.reals(&r.thin_posterior, 1);
    w.head("=== local loglikelihood bounds ===");                               w.reals(r.logLp.data(), nc);
    w.head("=== local volume -- log(<X_p>) ===");                               w.reals(r.logXp.data(), nc);
    w.head("=== last update volume ===");                                       w.reals(&r.logX_last_update, 1);
    w.head("=== global evidence volume cross correlation -- log(<ZX_p>) ===");  w.reals(r.logZXp.data(), nc);
    w.head("=== local evidence -- log(<Z_p>) ===");                             w.reals(r.logZp.data(), nc);
    w.head("=== local evidence^2 -- log(<Z_p^2>) ===");                         w.reals(r.logZp2.data(), nc);
    w.head("=== local evidence volume cross correlation -- log(<Z_pX_p>) ==="); w.reals(r.logZpXp.data(), nc);
    w.head("=== local volume cross correlation -- log(<X_pX_q>) ===");
    for (int q = 0; q < nc; ++q) w.reals(r.logXpXq.data() + (size_t)q * nc, nc);
    w.head("=== maximum log weights -- log(w_p) ===");                          w.reals(lz_nc.data(), nc);
    w.head("=== local dead evidence -- log(<Z_p>) ===");                        w.reals(r.logZp_dead.data(), ncd);
    w.head("=== local dead evidence^2 -- log(<Z_p^2>) ===");                    w.reals(r.logZp2_dead.data(), ncd);
    w.head("=== maximum dead log weights -- log(w_p) ===");                     w.reals(lz_ncd.data(), ncd);
    w.head("=== covariance matrices ===");
    for (int c = 0; c < nc; ++c) { w.sep(); for (int j = 0; j < D; ++j) w.reals(r.covmat.data() + ((size_t)c * D + j) * D, D); }
    w.head("=== cholesky decompositions ===");
    for (int c = 0; c < nc; ++c) { w.sep(); for (int j = 0; j < D; ++j) w.reals(r.cholesky.data() + ((size_t)c * D + j) * D, D); }
    w.head("=== live points ===");
    for (int c = 0; c < nc; ++c) { w.sep(); for (int i = 0; i < r.nlive[c]; ++i) w.reals(r.live[c].data() + (size_t)i * nT, nT); }
    w.head("=== dead points ===");
    for (int i = 0; i < r.ndead; ++i) w.reals(r.dead.data() + (size_t)i * nT, nT);
    w.head("=== logweights of dead points ===");                                w.reals(r.logweights.data(), r.ndead);
    w.head("=== phantom points ===");
    for (int c = 0; c < nc; ++c) { w.sep(); for (int i = 0; i < r.nphantom[c]; ++i) w.reals(r.phantom[c].data() + (size_t)i * nT, nT); }
    w.head("=== weighted posterior points ===");
    for (int c = 0; c < nc; ++c) w.sep();
    w.head("=== dead weighted posterior points ===");
    for (int c = 0; c < ncd; ++c) w.sep();
    w.head("=== global weighted posterior points ===");
    w.head("=== equally weighted posterior points ===");
    for (int c = 0; c < nc; ++c) w.sep();
    w.head("=== dead equally weighted posterior points ===");
    for (int c = 0; c < ncd; ++c) w.sep();
    w.head("=== global equally weighted posterior points ===");
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { err = "cannot rename " + tmp; return false; }
    return true;
}

bool pc_resume_read(const std::string &path, PcResume &r, std::string &err)
{
    std::ifstream in(path);
    if (!in) { err = "cannot open " + path; return false; }
    Reader R;
    std::string l;
    while (std::getline(in, l)) {
        if (is_head(l)) { R.secs.push_back({l, {}}); continue; }
        if (R.secs.empty()) continue;
        R.secs.back().lines.push_back(is_sep(l) ? std::string("---") : l);
    }
    int ngrades = 0, np_global = 0, ne_global = 0;
    std::vector<int> tmp_i;
    std::vector<double> tmp_d, nlk;
    std::vector<std::vector<double>> bl;
    bool ok = R.one_int("Number of dimensions", r.nDims) && R.one_int("Number of derived", r.nDerived) &&
              R.one_int("Number of dead points", r.ndead) && R.one_int("Number of clusters", r.ncluster) &&
              R.one_int("Number of dead clusters", r.ncluster_dead) && R.one_int("global weighted posterior", np_global) &&
              R.one_int("global equally weighted", ne_global) && R.one_int("Number of grades", ngrades) &&
              R.ints("positions of grades", r.grade_dims) && R.ints("Number of repeats", r.num_repeats) &&
              R.flat("Number of likelihood calls", nlk) && R.ints("live points in each cluster", r.nlive) &&
              R.ints("phantom points in each cluster", r.nphantom) && R.ints("weighted posterior points in each cluster", tmp_i) &&
              R.ints("equally weighted posterior points in each cluster", tmp_i) && R.ints("Minimum loglikelihood positions", r.imin) &&
              R.ints("weighted posterior points in each dead cluster", tmp_i) && R.ints("equally weighted posterior points in each dead cluster", tmp_i) &&
              R.one_real("global evidence --", r.logZ) && R.one_real("global evidence^2", r.logZ2) &&
              R.one_real("posterior thin factor", r.thin_posterior) && R.flat("local loglikelihood bounds", r.logLp) &&
              R.flat("local volume --", r.logXp) && R.one_real("last update volume", r.logX_last_update) &&
              R.flat("global evidence volume cross", r.logZXp) && R.flat("local evidence --", r.logZp) &&
              R.flat("local evidence^2", r.logZp2) && R.flat("local evidence volume cross", r.logZpXp) &&
              R.flat("local volume cross correlation", r.logXpXq) && R.flat("maximum log weights", tmp_d) &&
              R.flat("local dead evidence --", r.logZp_dead) && R.flat("local dead evidence^2", r.logZp2_dead) &&
              R.flat("maximum dead log weights", tmp_d) && R.flat("covariance matrices", r.covmat) &&
              R.flat("cholesky decompositions", r.cholesky) && R.blocks("=== live points", bl);
    if (!ok) { err = path + ": " + R.err; return false; }
    r.nlike.assign(nlk.size(), 0);
    for (size_t i = 0; i < nlk.size(); ++i) r.nlike[i] = std::llround(nlk[i]);
    const int nc = r.ncluster, nT = r.nTotal(), D = r.nDims;
    auto bad = [&](const char *what) { err = path + ": inconsistent " + what; return false; };
    if ((int)r.nlive.size() != nc || (int)r.nphantom.size() != nc || (int)r.logXp.size() != nc || (int)r.logZp.size() != nc ||
        (int)r.logZXp.size() != nc || (int)r.logZp2.size() != nc || (int)r.logZpXp.size() != nc || (int)r.logLp.size() != nc)
        return bad("per-cluster arrays");
    if ((int)r.logXpXq.size() != nc * nc) return bad("log(<X_pX_q>)");
    if ((int)r.covmat.size() != nc * D * D || (int)r.cholesky.size() != nc * D * D) return bad("covariance matrices");
    if ((int)r.logZp_dead.size() != r.ncluster_dead || (int)r.logZp2_dead.size() != r.ncluster_dead) return bad("dead cluster evidences");
    if ((int)bl.size() < nc) bl.resize(nc);
    r.live.assign(nc, {});
    for (int c = 0; c < nc; ++c) {
        if ((long)bl[c].size() != (long)r.nlive[c] * nT) return bad("live points");
        r.live[c] = bl[c];
    }
    if (!R.flat("=== dead points", r.dead) || !R.flat("logweights of dead points", r.logweights) || !R.blocks("=== phantom points", bl)) { err = path + ": " + R.err; return false; }
    if ((long)r.dead.size() != (long)r.ndead * nT || (int)r.logweights.size() != r.ndead) return bad("dead points");
    if ((int)bl.size() < nc) bl.resize(nc);
    r.phantom.assign(nc, {});
    for (int c = 0; c < nc; ++c) {
        if ((long)bl[c].size() != (long)r.nphantom[c] * nT) return bad("phantom points");
        r.phantom[c] = bl[c];
    }
    // the posterior stacks that follow are rebuilt from the dead points at the end of the run
    return true;
}
