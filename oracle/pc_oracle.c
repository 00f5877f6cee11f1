/*
 * pc_oracle.c -- CPU restatement of the PolyChordLite nested-sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pc_oracle.h).  Written from SURVEY.md section 8(a)
 * and a reading of the reference's behaviour; every routine names the reference
 * file:line (under /root/reference) it restates.  No reference source is copied:
 * the reference is Fortran 2003, this is C99 with different data structures.
 *
 * Two RNG modes:
 *   keyed      : every uniform is Philox4x32-10(counter=(idx>>1, stream_lo, stream_hi, domain))
 *                -- the layout the HIP engine uses, so engine and oracle consume identical
 *                numbers whatever their execution order;
 *   sequential : one running stream consumed in the reference's own program order,
 *                used with batch=1 to pin this restatement against the reference
 *                binary whose `random_number` is fed the same stream (oracle/ref_rng_shim.c).
 */
#include "pc_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define HUGE_D DBL_MAX

/* ========================================================================== */
/* RNG                                                                          */
/* ========================================================================== */
void pc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    /* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11) */
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double pc_uniform_keyed(const uint32_t key[2], uint32_t dom, uint32_t shi, uint32_t slo, uint32_t idx)
{
    uint32_t ctr[4] = { idx >> 1, slo, shi, dom }, o[4];
    pc_philox4x32_10(ctr, key, o);
    uint64_t w = (idx & 1u) ? (((uint64_t)o[2] << 32) | o[3]) : (((uint64_t)o[0] << 32) | o[1]);
    /* 53 random bits, centred: never 0 or 1 */
    return ((double)(w >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

double pc_rng_u(pc_rng *r, uint32_t dom, uint32_t shi, uint32_t slo, uint32_t idx)
{
    if (r->sequential) {
        uint64_t n = r->seq++;
        return pc_uniform_keyed(r->key, PC_DOM_SEQ, (uint32_t)(n >> 32), 0u, (uint32_t)n);
    }
    return pc_uniform_keyed(r->key, dom, shi, slo, idx);
}

static double rng_post(pc_rng *r)
{
    if (r->sequential) return pc_rng_u(r, 0, 0, 0, 0);
    uint64_t n = r->post++;
    return pc_uniform_keyed(r->key, PC_DOM_POST, (uint32_t)(n >> 32), 0u, (uint32_t)n);
}

/* ========================================================================== */
/* numerics units                                                               */
/* ========================================================================== */
static double poly8(const double *a, double x)
{   /* Horner, highest coefficient first (utils.F90:1015-1021) */
    double v = 0.0;
    for (int i = 7; i >= 0; --i) v = v * x + a[i];
    return v;
}

/* Wichura (1988) Algorithm AS 241, PPND16.  Restates utils.F90:806-966. */
double pc_inv_normal_cdf(double p)
{
    static const double a[8] = { 3.3871328727963666080e+00, 1.3314166789178437745e+02,
        1.9715909503065514427e+03, 1.3731693765509461125e+04, 4.5921953931549871457e+04,
        6.7265770927008700853e+04, 3.3430575583588128105e+04, 2.5090809287301226727e+03 };
    static const double b[8] = { 1.0, 4.2313330701600911252e+01, 6.8718700749205790830e+02,
        5.3941960214247511077e+03, 2.1213794301586595867e+04, 3.9307895800092710610e+04,
        2.8729085735721942674e+04, 5.2264952788528545610e+03 };
    static const double c[8] = { 1.42343711074968357734e+00, 4.63033784615654529590e+00,
        5.76949722146069140550e+00, 3.64784832476320460504e+00, 1.27045825245236838258e+00,
        2.41780725177450611770e-01, 2.27238449892691845833e-02, 7.74545014278341407640e-04 };
    static const double d[8] = { 1.0, 2.05319162663775882187e+00, 1.67638483018380384940e+00,
        6.89767334985100004550e-01, 1.48103976427480074590e-01, 1.51986665636164571966e-02,
        5.47593808499534494600e-04, 1.05075007164441684324e-09 };
    static const double e[8] = { 6.65790464350110377720e+00, 5.46378491116411436990e+00,
        1.78482653991729133580e+00, 2.96560571828504891230e-01, 2.65321895265761230930e-02,
        1.24266094738807843860e-03, 2.71155556874348757815e-05, 2.01033439929228813265e-07 };
    static const double f[8] = { 1.0, 5.99832206555887937690e-01, 1.36929880922735805310e-01,
        1.48753612908506148525e-02, 7.86869131145613259100e-04, 1.84631831751005468180e-05,
        1.42151175831644588870e-07, 2.04426310338993978564e-15 };
    if (p <= 0.0) return -HUGE_D;
    if (p >= 1.0) return HUGE_D;
    double q = p - 0.5, r, v;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        return q * poly8(a, r) / poly8(b, r);
    }
    r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) { r -= 1.6; v = poly8(c, r) / poly8(d, r); }
    else          { r -= 5.0; v = poly8(e, r) / poly8(f, r); }
    return (q < 0.0) ? -v : v;
}

double pc_logsumexp(const double *v, int n)
{   /* utils.F90:362-374 */
    double m = v[0];
    for (int i = 1; i < n; ++i) if (v[i] > m) m = v[i];
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(v[i] - m);
    return m + log(s);
}

double pc_logaddexp(double a, double b)
{   /* utils.F90:377-389 */
    if (a > b) return a + log(exp(b - a) + 1.0);
    return b + log(exp(a - b) + 1.0);
}

void pc_logincexp(double *a, double b) { *a = pc_logaddexp(*a, b); } /* utils.F90:417-439 */

static void logincexp2(double *a, double b, double c) { pc_logincexp(a, b); pc_logincexp(a, c); }

void pc_cholesky(const double *a, int n, double *L)
{   /* utils.F90:621-649; row-major, L[j*n+i] with j>=i */
    memset(L, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < i; ++k) s += L[i * n + k] * L[i * n + k];
        double dii = a[i * n + i] - s;
        if (dii <= 0.0) {
            if (getenv("PC_ORACLE_TRACE_CHOL")) fprintf(stderr, "oracle chol fallback: n=%d column %d pivot %.3e\n", n, i, dii);
            double tr = 0.0;
            for (int k = 0; k < n; ++k) tr += a[k * n + k];
            memset(L, 0, sizeof(double) * n * n);
            for (int k = 0; k < n; ++k) L[k * n + k] = sqrt(tr);
            return;
        }
        L[i * n + i] = sqrt(dii);
        for (int j = i + 1; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < i; ++k) t += L[i * n + k] * L[j * n + k];
            L[j * n + i] = (a[i * n + j] - t) / L[i * n + i];
        }
    }
}

void pc_covmat(const double *live, int nlive, const double *ph, int nph, int w, int D, double *cov)
{   /* run_time_info.f90:613-634: population covariance of live U phantom cube coordinates */
    double *mean = (double *)calloc(D, sizeof(double));
    for (int i = 0; i < nlive; ++i) for (int d = 0; d < D; ++d) mean[d] += live[(size_t)i * w + d];
    double *m2 = (double *)calloc(D, sizeof(double));
    for (int i = 0; i < nph; ++i) for (int d = 0; d < D; ++d) m2[d] += ph[(size_t)i * w + d];
    for (int d = 0; d < D; ++d) mean[d] = (mean[d] + m2[d]) / (double)(nlive + nph);
    for (int i = 0; i < D * D; ++i) cov[i] = 0.0;
    double *acc = (double *)calloc((size_t)D * D, sizeof(double));
    for (int i = 0; i < nlive; ++i)
        for (int a = 0; a < D; ++a) {
            double xa = live[(size_t)i * w + a] - mean[a];
            for (int b = 0; b < D; ++b) cov[a * D + b] += xa * (live[(size_t)i * w + b] - mean[b]);
        }
    for (int i = 0; i < nph; ++i)
        for (int a = 0; a < D; ++a) {
            double xa = ph[(size_t)i * w + a] - mean[a];
            for (int b = 0; b < D; ++b) acc[a * D + b] += xa * (ph[(size_t)i * w + b] - mean[b]);
        }
    for (int i = 0; i < D * D; ++i) cov[i] = (cov[i] + acc[i]) / (double)(nlive + nph);
    free(mean); free(m2); free(acc);
}

void pc_similarity(const double *x, int n, int stride, int D, double *S)
{   /* calculate.f90:94-109:  S_ij = x_i.x_i + x_j.x_j - 2 x_i.x_j */
    double *r = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += x[(size_t)i * stride + d] * x[(size_t)i * stride + d];
        r[i] = s;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += x[(size_t)i * stride + d] * x[(size_t)j * stride + d];
            S[(size_t)i * n + j] = (r[i] + r[j]) - 2.0 * s;
        }
    free(r);
}

/* ---- kNN clustering ------------------------------------------------------ */
void pc_compute_knn(const double *S, int n, int k, int *knn)
{   /* clustering.f90:134-174: knn[i*k + m] = m-th nearest (0-based ids), stable insertion */
    double *d2 = (double *)malloc(sizeof(double) * k);
    for (int i = 0; i < n; ++i) {
        int *row = knn + (size_t)i * k;
        for (int m = 0; m < k; ++m) { d2[m] = HUGE_D; row[m] = -1; }
        for (int j = 0; j < n; ++j) {
            double s = S[(size_t)i * n + j];
            /* first slot whose stored distance is strictly greater (list is ascending) */
            int pos = -1;
            double best = 0.0;
            for (int m = 0; m < k; ++m)
                if (d2[m] > s && (pos < 0 || d2[m] < best)) { pos = m; best = d2[m]; }
            if (pos >= 0) {
                for (int m = k - 1; m > pos; --m) { d2[m] = d2[m - 1]; row[m] = row[m - 1]; }
                d2[pos] = s; row[pos] = j;
            }
        }
    }
    free(d2);
}

static int relabel(int *lab, int n)
{   /* utils.F90:713-749: labels renamed 1,2,.. in order of first appearance */
    int *map = (int *)malloc(sizeof(int) * n), nl = 0;
    int *out = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) {
        int f = -1;
        for (int m = 0; m < nl; ++m) if (map[m] == lab[i]) { f = m; break; }
        if (f < 0) { map[nl] = lab[i]; f = nl++; }
        out[i] = f + 1;
    }
    memcpy(lab, out, sizeof(int) * n);
    free(map); free(out);
    return nl;
}

static void clustering_k(const int *knn, int kstride, int nn, int n, int *c)
{   /* clustering.f90:100-130 with neighbours() :178-188 -- connected components of the
       "i in j's nn-list or j in i's" graph, label = smallest member (1-based) */
    for (int i = 0; i < n; ++i) c[i] = i + 1;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            if (c[i] == c[j]) continue;
            const int *ki = knn + (size_t)i * kstride, *kj = knn + (size_t)j * kstride;
            int nb = 0;
            for (int m = 0; m < nn && !nb; ++m) if (ki[m] == kj[0]) nb = 1;
            for (int m = 0; m < nn && !nb; ++m) if (kj[m] == ki[0]) nb = 1;
            if (nb) {
                int ci = c[i], cj = c[j], lo = ci < cj ? ci : cj;
                for (int m = 0; m < n; ++m) if (c[m] == ci || c[m] == cj) c[m] = lo;
            }
        }
}

int pc_nn_clustering(const double *S, int n, int *labels)
{   /* clustering.f90:15-97 */
    if (n <= 1) { for (int i = 0; i < n; ++i) labels[i] = 1; return 1; }
    int k = n < 10 ? n : 10;
    int kcap = n;                       /* storage stride for knn lists */
    int *knn = (int *)malloc(sizeof(int) * (size_t)n * kcap);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)n * kcap);
    int *old = (int *)malloc(sizeof(int) * n);
    int num = n;
    /* knn stored with stride kcap so that growing k keeps rows in place */
    pc_compute_knn(S, n, k, tmp);
    for (int i = 0; i < n; ++i) memcpy(knn + (size_t)i * kcap, tmp + (size_t)i * k, sizeof(int) * k);
    for (int i = 0; i < n; ++i) { old[i] = i + 1; labels[i] = i + 1; }
    int k0 = k;                         /* Fortran DO trip count is fixed at loop entry */
    for (int nn = 2; nn <= k0; ++nn) {
        clustering_k(knn, kcap, nn, n, labels);
        num = relabel(labels, n);
        if (num == 1) { free(knn); free(tmp); free(old); return 1; }
        int same = 1;
        for (int i = 0; i < n; ++i) if (labels[i] != old[i]) { same = 0; break; }
        if (same) break;
        if (nn == k) {
            k = (2 * k < n) ? 2 * k : n;
            pc_compute_knn(S, n, k, tmp);
            for (int i = 0; i < n; ++i) memcpy(knn + (size_t)i * kcap, tmp + (size_t)i * k, sizeof(int) * k);
        }
        memcpy(old, labels, sizeof(int) * n);
    }
    free(knn); free(tmp); free(old);
    if (num > 1) {
        int ic = 1;
        while (ic <= num) {
            int m = 0;
            int *pts = (int *)malloc(sizeof(int) * n);
            for (int j = 0; j < n; ++j) if (labels[j] == ic) pts[m++] = j;
            double *sub = (double *)malloc(sizeof(double) * (size_t)m * m);
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) sub[(size_t)a * m + b] = S[(size_t)pts[a] * n + pts[b]];
            int *sl = (int *)malloc(sizeof(int) * (m > 0 ? m : 1));
            int nnew = pc_nn_clustering(sub, m, sl);
            for (int a = 0; a < m; ++a) labels[pts[a]] = num + sl[a];
            if (nnew == 1) ic++;
            num = relabel(labels, n);
            free(pts); free(sub); free(sl);
        }
    }
    return num;
}

/* ========================================================================== */
/* likelihoods / priors                                                         */
/* ========================================================================== */
#define LOG_TWO_PI 1.8378770664093453   /* log(2*pi) */
static const double TWO_PI = 6.283185307179586;

double pc_like_eval(const pc_like *L, const double *th, int D, double *phi, int nDerived)
{
    switch (L->kind) {
    case PC_LIKE_GAUSSIAN: {      /* likelihoods/examples/gaussian.f90:12-41 */
        double norm = -(double)D * (log(L->sigma) + LOG_TWO_PI / 2.0), s = 0.0, r2 = 0.0;
        for (int d = 0; d < D; ++d) {
            double z = (th[d] - L->mu) / L->sigma;
            s += z * z;
            r2 += (th[d] - L->mu) * (th[d] - L->mu);
        }
        if (nDerived >= 1) phi[0] = sqrt(r2);
        if (nDerived >= 2) {
            /* log( r^D * V_D ),  V_D = pi^(D/2)/Gamma(1+D/2)  (utils.F90:754-760) */
            double Vn = pow(sqrt(3.14159265358979323846), (double)D) / tgamma(1.0 + D / 2.0);
            phi[1] = log(pow(phi[0], (double)D) * Vn);
        }
        return norm - s / 2.0;
    }
    case PC_LIKE_RASTRIGIN: {     /* likelihoods/examples/rastrigin.f90:20-35 */
        double s = 0.0;
        for (int d = 0; d < D; ++d)
            s += log(4991.21750) + th[d] * th[d] - 10.0 * cos(TWO_PI * th[d]);
        return -s;
    }
    case PC_LIKE_TWIN_GAUSSIAN: { /* likelihoods/examples/twin_gaussian.f90:14-56 */
        double norm = -(double)D * (log(L->sigma) + LOG_TWO_PI / 2.0), s1 = 0.0, s2 = 0.0;
        for (int d = 0; d < D; ++d) {
            double m1 = (d < 2) ? -0.5 : 0.0, m2 = (d < 2) ? 0.5 : 0.0;
            double z1 = (th[d] - m1) / L->sigma, z2 = (th[d] - m2) / L->sigma;
            s1 += z1 * z1; s2 += z2 * z2;
        }
        if (nDerived >= 1) phi[0] = (th[0] > 0.5) ? 1.0 : -1.0;
        return pc_logaddexp(norm - s1 / 2.0, norm - s2 / 2.0) - log(2.0);
    }
    case PC_LIKE_CORR_GAUSSIAN: { /* random_gaussian.f90:17-30 + utils.F90:1028-1048 */
        double q = 0.0;
        for (int a = 0; a < D; ++a) {
            double t = 0.0;
            for (int b = 0; b < D; ++b) t += L->invcov[(size_t)a * D + b] * (th[b] - L->mean[b]);
            q += (th[a] - L->mean[a]) * t;
        }
        return -((double)D * LOG_TWO_PI + L->logdetcov) / 2.0 - q / 2.0;
    }
    default:
        return L->fn(th, D, phi, nDerived, L->ctx);
    }
}

static void prior_eval(const pc_prior *P, const double *cube, double *theta, int D)
{   /* priors.f90:40-55 (uniform_htp): theta = lo + (hi-lo)*cube */
    if (P->kind == 1) {
        for (int d = 0; d < D; ++d) {
            double lo = P->lo ? P->lo[d] : 0.0, hi = P->hi ? P->hi[d] : 1.0;
            theta[d] = lo + (hi - lo) * cube[d];
        }
    } else P->fn(cube, theta, D, P->ctx);
}

/* ========================================================================== */
/* sampler state                                                                */
/* ========================================================================== */
typedef struct { double *a; uint64_t *uid; int n, cap, w; } ptarr;

static void pa_init(ptarr *p, int w) { p->a = NULL; p->uid = NULL; p->n = 0; p->cap = 0; p->w = w; }
static void pa_free(ptarr *p) { free(p->a); free(p->uid); p->a = NULL; p->uid = NULL; p->n = p->cap = 0; }
static void pa_add(ptarr *p, const double *row, uint64_t uid)
{   /* array_utils.f90:396-428 (append, capacity doubling) */
    if (p->n == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 64;
        p->a = (double *)realloc(p->a, sizeof(double) * (size_t)p->cap * p->w);
        p->uid = (uint64_t *)realloc(p->uid, sizeof(uint64_t) * (size_t)p->cap);
    }
    memcpy(p->a + (size_t)p->n * p->w, row, sizeof(double) * p->w);
    p->uid[p->n] = uid;
    p->n++;
}
static void pa_del(ptarr *p, int i, double *out, uint64_t *uid)
{   /* array_utils.f90:433-458 (overwrite with last) */
    if (out) memcpy(out, p->a + (size_t)i * p->w, sizeof(double) * p->w);
    if (uid) *uid = p->uid[i];
    if (i != p->n - 1) {
        memcpy(p->a + (size_t)i * p->w, p->a + (size_t)(p->n - 1) * p->w, sizeof(double) * p->w);
        p->uid[i] = p->uid[p->n - 1];
    }
    p->n--;
}
static void pa_copy(ptarr *dst, const ptarr *src)
{
    pa_init(dst, src->w);
    for (int i = 0; i < src->n; ++i) pa_add(dst, src->a + (size_t)i * src->w, src->uid[i]);
}

typedef struct {
    ptarr live, phantom, pstack, posterior, equals;
    int npstack0;           /* scratch */
    double logZp, logXp, logZXp, logZp2, logZpXp, logLp, maxlogweight;
    int imin;               /* position of the lowest live point (-1 if empty) */
    double *chol, *cov;
} cluster_t;

typedef struct {
    const pc_settings *s; const pc_like *like; const pc_prior *prior; pc_rng rng;
    pc_settings s_local;                     /* settings with num_repeats = total over the grades */
    int ngrade, g_off[8], g_nr[8];           /* grades: first parameter and repeats of each (chordal_sampling.f90:119-130) */
    long nlike_g[8];
    int D, nDer, nTotal, npost, np;
    int h0, p0, d0, b0, l0;                 /* 0-based offsets (settings.f90:163-182) */
    int pos_X, pos_l, pos_w, pos_Z, pos_p0; /* posterior layout (settings.f90:186-203) */
    int ncluster, ccap;
    cluster_t *cl;
    double *XpXq;                           /* ccap x ccap */
    double logZ, logZ2;
    ptarr dead; double *logweights; int lwcap;
    long nlike;
    double logX_last_update, thin_posterior, maxlogweight_global;
    uint32_t post_round;      /* update_posteriors calls so far (keyed mode: the thinning round of bernoulli_post) */
    ptarr posterior_global, equals_global;
    /* dead clusters */
    int ncluster_dead; double *logZp_dead, *logZp2_dead;
    long nposterior_dead_tot, nequals_dead_tot;
    int nlive_target_static;
    /* chains in flight (keyed mode, B > 1): cluster index and epoch of every chain still in the nursery; see remap_chains */
    int *w_cluster, *w_epoch, w_n;
} rti_t;

#define XQ(R, p, q) ((R)->XpXq[(size_t)(p) * (R)->ccap + (q)])

static void cluster_init(rti_t *R, cluster_t *c)
{
    pa_init(&c->live, R->nTotal); pa_init(&c->phantom, R->nTotal);
    pa_init(&c->pstack, R->npost); pa_init(&c->posterior, R->npost); pa_init(&c->equals, R->np);
    c->logZp = c->logZXp = c->logZp2 = c->logZpXp = R->s->logzero;
    c->logXp = 0.0; c->logLp = R->s->logzero; c->imin = -1; c->maxlogweight = R->s->logzero;
    c->chol = (double *)calloc((size_t)R->D * R->D, sizeof(double));
    c->cov = (double *)calloc((size_t)R->D * R->D, sizeof(double));
    for (int d = 0; d < R->D; ++d) c->chol[d * R->D + d] = c->cov[d * R->D + d] = 1.0;
}
static void cluster_free(cluster_t *c)
{
    pa_free(&c->live); pa_free(&c->phantom); pa_free(&c->pstack); pa_free(&c->posterior); pa_free(&c->equals);
    free(c->chol); free(c->cov);
}

static void ensure_ccap(rti_t *R, int need)
{
    if (need <= R->ccap) return;
    int nc = R->ccap ? R->ccap : 4;
    while (nc < need) nc *= 2;
    double *nx = (double *)malloc(sizeof(double) * (size_t)nc * nc);
    for (int i = 0; i < nc * nc; ++i) nx[i] = 0.0;
    for (int p = 0; p < R->ccap; ++p) for (int q = 0; q < R->ccap; ++q) nx[(size_t)p * nc + q] = XQ(R, p, q);
    free(R->XpXq); R->XpXq = nx;
    R->cl = (cluster_t *)realloc(R->cl, sizeof(cluster_t) * nc);
    R->ccap = nc;
}

static void find_min_loglikelihoods(rti_t *R)
{   /* run_time_info.f90:883-909 */
    for (int c = 0; c < R->ncluster; ++c) {
        cluster_t *C = &R->cl[c];
        if (C->live.n == 0) { C->imin = -1; C->logLp = HUGE_D; continue; }
        int im = 0; double m = C->live.a[R->l0];
        for (int i = 1; i < C->live.n; ++i) {
            double v = C->live.a[(size_t)i * R->nTotal + R->l0];
            if (v < m) { m = v; im = i; }
        }
        C->imin = im; C->logLp = m;
    }
}

static double logsumexp_Xp(const rti_t *R)
{
    double m = R->cl[0].logXp;
    for (int c = 1; c < R->ncluster; ++c) if (R->cl[c].logXp > m) m = R->cl[c].logXp;
    double s = 0.0;
    for (int c = 0; c < R->ncluster; ++c) s += exp(R->cl[c].logXp - m);
    return m + log(s);
}

static double update_evidence(rti_t *R, int p)
{   /* run_time_info.f90:211-296 */
    cluster_t *P = &R->cl[p];
    const double log2 = log(2.0);
    double logL = P->logLp;
    double lognp = log((double)P->live.n + 0.0), lognp1 = log((double)P->live.n + 1.0),
           lognp2 = log((double)P->live.n + 2.0);
    double logweight = P->logXp - lognp1;
    pc_logincexp(&R->logZ, P->logXp + logL - lognp1);
    pc_logincexp(&P->logZp, P->logXp + logL - lognp1);
    P->logXp = P->logXp + lognp - lognp1;
    logincexp2(&R->logZ2, log2 + P->logZXp + logL - lognp1,
               log2 + XQ(R, p, p) + 2 * logL - lognp1 - lognp2);
    P->logZXp = P->logZXp + lognp - lognp1;
    pc_logincexp(&P->logZXp, XQ(R, p, p) + logL + lognp - lognp1 - lognp2);
    for (int q = 0; q < R->ncluster; ++q)
        if (q != p) pc_logincexp(&R->cl[q].logZXp, XQ(R, p, q) + logL - lognp1);
    logincexp2(&P->logZp2, log2 + P->logZpXp + logL - lognp1,
               log2 + XQ(R, p, p) + 2 * logL - lognp1 - lognp2);
    P->logZpXp = P->logZpXp + lognp - lognp1;
    pc_logincexp(&P->logZpXp, XQ(R, p, p) + logL + lognp - lognp1 - lognp2);
    XQ(R, p, p) = XQ(R, p, p) + lognp - lognp2;
    for (int q = 0; q < R->ncluster; ++q)
        if (q != p) {
            XQ(R, p, q) = XQ(R, p, q) + lognp - lognp1;
            XQ(R, q, p) = XQ(R, q, p) + lognp - lognp1;
        }
    return logweight;
}

static void add_dead(rti_t *R, const double *pt, double logweight)
{
    pa_add(&R->dead, pt, 0);
    if (R->dead.n > R->lwcap) {
        R->lwcap = R->lwcap ? R->lwcap * 2 : 1024;
        while (R->lwcap < R->dead.n) R->lwcap *= 2;
        R->logweights = (double *)realloc(R->logweights, sizeof(double) * R->lwcap);
    }
    R->logweights[R->dead.n - 1] = logweight;
}

static int total_live(const rti_t *R)
{
    int n = 0;
    for (int c = 0; c < R->ncluster; ++c) n += R->cl[c].live.n;
    return n;
}

static void delete_outermost_point(rti_t *R)
{   /* run_time_info.f90:789-817 */
    int cd = 0;
    for (int c = 1; c < R->ncluster; ++c) if (R->cl[c].logLp < R->cl[cd].logLp) cd = c; /* minpos: first min */
    cluster_t *C = &R->cl[cd];
    double logweight = update_evidence(R, cd);
    double *pt = (double *)malloc(sizeof(double) * R->nTotal);
    pa_del(&C->live, C->imin, pt, NULL);
    find_min_loglikelihoods(R);
    add_dead(R, pt, logweight);
    /* posterior stack row: calculate.f90:53-79 */
    double *pp = (double *)malloc(sizeof(double) * R->npost);
    pp[R->pos_X] = logsumexp_Xp(R);
    pp[R->pos_l] = pt[R->l0];
    pp[R->pos_w] = logweight;
    pp[R->pos_Z] = R->logZ;
    memcpy(pp + R->pos_p0, pt + R->p0, sizeof(double) * (R->D + R->nDer));
    pa_add(&C->pstack, pp, (uint64_t)(R->dead.n - 1));      /* (keyed mode: the row's id = its index in death order, see bernoulli_post) */
    double lw = pp[R->pos_w] + pp[R->pos_l];
    if (lw > C->maxlogweight) C->maxlogweight = lw;
    if (C->maxlogweight > R->maxlogweight_global) R->maxlogweight_global = C->maxlogweight;
    free(pt); free(pp);
}

static int identify_cluster(const rti_t *R, const double *pt)
{   /* run_time_info.f90:913-949: cluster of the nearest live point in cube coordinates */
    if (R->ncluster == 1) return 0;
    double best = HUGE_D; int cb = 0;
    for (int c = 0; c < R->ncluster; ++c) {
        const cluster_t *C = &R->cl[c];
        for (int i = 0; i < C->live.n; ++i) {
            const double *q = C->live.a + (size_t)i * R->nTotal;
            double d2 = 0.0;
            for (int d = 0; d < R->D; ++d) { double t = pt[d] - q[d]; d2 += t * t; }
            if (d2 < best) { best = d2; cb = c; }
        }
    }
    return cb;
}

static int nlive_target(const rti_t *R, double logL)
{   /* run_time_info.f90:766-771: maxloc(loglikes, mask = logL > loglikes) */
    const pc_settings *s = R->s;
    if (s->n_nlives <= 0) return (logL > s->logzero) ? s->nlive : s->nlive;
    int best = -1;
    for (int i = 0; i < s->n_nlives; ++i)
        if (logL > s->loglikes[i] && (best < 0 || s->loglikes[i] > s->loglikes[best])) best = i;
    return best < 0 ? s->nlive : s->nlives[best];
}

static int replace_point(rti_t *R, const double *babies, const uint64_t *uids, int nb, int cluster_add)
{   /* run_time_info.f90:716-787 */
    double logL = R->cl[0].logLp;
    for (int c = 1; c < R->ncluster; ++c) if (R->cl[c].logLp < logL) logL = R->cl[c].logLp;
    for (int i = 0; i < nb - 1; ++i) {
        const double *pt = babies + (size_t)i * R->nTotal;
        if (pt[R->l0] > logL && identify_cluster(R, pt) == cluster_add)
            pa_add(&R->cl[cluster_add].phantom, pt, uids[i]);
    }
    const double *pt = babies + (size_t)(nb - 1) * R->nTotal;
    int replaced = 0;
    if (pt[R->l0] > logL) {
        if (identify_cluster(R, pt) == cluster_add) {
            int nl = nlive_target(R, logL);
            int cdel = -1, idel = -1;
            if (total_live(R) >= (nl > 1 ? nl : 1)) {
                cdel = 0;
                for (int c = 1; c < R->ncluster; ++c) if (R->cl[c].logLp < R->cl[cdel].logLp) cdel = c;
                idel = R->cl[cdel].imin;
                delete_outermost_point(R); replaced = 1;
            }
            if (total_live(R) < nl) {
                ptarr *lv = &R->cl[cluster_add].live;
                pa_add(lv, pt, uids[nb - 1]);
                if (!R->rng.sequential && replaced && cdel == cluster_add && idel < lv->n - 1) {
                    /* KEYED MODE ONLY (the HIP engine's rule): the newcomer takes the list position of the
                     * point it replaces instead of array_utils.f90:396-458's swap-with-last + append.
                     * Same set of live points, different list order => which point a seed index selects;
                     * the sequential mode used for pinning keeps the reference rule. */
                    double *a = lv->a + (size_t)idel * R->nTotal, *b = lv->a + (size_t)(lv->n - 1) * R->nTotal;
                    for (int e = 0; e < R->nTotal; ++e) { double t = a[e]; a[e] = b[e]; b[e] = t; }
                    uint64_t tu = lv->uid[idel]; lv->uid[idel] = lv->uid[lv->n - 1]; lv->uid[lv->n - 1] = tu;
                }
                find_min_loglikelihoods(R);
            }
        }
    } else {
        add_dead(R, pt, R->s->logzero);
    }
    return replaced;
}

/* A Bernoulli trial of update_posteriors (run_time_info.f90:975-1062).  Sequential mode: the next draw of the ONE stream, in the
 * reference's program order (what the reference binary pins).  Keyed mode (the HIP engine's): the reference's trials are independent
 * draws, and WHICH draw a row gets is an accident of its arrays' order (per-cluster stacks, delete = overwrite-with-last) -- so the
 * trial of posterior row `uid` (a dead point: its index in death order; a phantom kept by boost_posterior: its phantom id, top bit set)
 * in thinning round `post_round` (the number of the update_posteriors call) is the draw keyed by (round, row): the same rows survive
 * whatever order the lists are walked in, with several clusters and with phantoms in the stack.  list: 0 the global list, 1 a
 * cluster's own (cluster_posteriors). */
static int bernoulli_post(rti_t *R, double p, uint64_t uid, uint32_t list)
{
    if (R->rng.sequential) return rng_post(&R->rng) < p;
    const uint32_t a = (R->post_round & 0x0FFFFFFFu) | ((uid >> 63) ? 0x80000000u : 0u) | (list ? 0x40000000u : 0u);
    return pc_uniform_keyed(R->rng.key, PC_DOM_POST, a, (uint32_t)((uid & 0x7FFFFFFFFFFFFFFFull) >> 32), (uint32_t)uid) < p;
}

static void clean_phantoms(rti_t *R)
{   /* run_time_info.f90:820-877.  The posterior stack of a cluster holds that cluster's
       deaths since the last update in ascending logL, so "first stack entry above the
       phantom" (minloc with mask) is an upper-bound search. */
    const pc_settings *s = R->s;
    for (int c = 0; c < R->ncluster; ++c) {
        cluster_t *C = &R->cl[c];
        int ns0 = C->pstack.n;
        int i = 0;
        double *pt = (double *)malloc(sizeof(double) * R->nTotal);
        double *pp = (double *)malloc(sizeof(double) * R->npost);
        while (i < C->phantom.n) {
            double pl = C->phantom.a[(size_t)i * R->nTotal + R->l0];
            int lo = 0, hi = ns0;           /* first index with stack logL > pl */
            while (lo < hi) {
                int mid = (lo + hi) / 2;
                if (C->pstack.a[(size_t)mid * R->npost + R->pos_l] > pl) hi = mid; else lo = mid + 1;
            }
            if (lo >= ns0) { i++; continue; }
            uint64_t uid;
            pa_del(&C->phantom, i, pt, &uid);
            if (s->equals || s->posteriors) {
                double u = R->rng.sequential ? pc_rng_u(&R->rng, 0, 0, 0, 0)
                           : pc_uniform_keyed(R->rng.key, PC_DOM_PHANTOM, (uint32_t)(uid >> 32), (uint32_t)uid, 0);
                if (u < R->thin_posterior) {
                    const double *st = C->pstack.a + (size_t)lo * R->npost;
                    pp[R->pos_X] = st[R->pos_X];
                    pp[R->pos_l] = pt[R->l0];
                    pp[R->pos_w] = st[R->pos_w];
                    pp[R->pos_Z] = st[R->pos_Z];
                    memcpy(pp + R->pos_p0, pt + R->p0, sizeof(double) * (R->D + R->nDer));
                    pa_add(&C->pstack, pp, uid | (1ull << 63));
                    double lw = pp[R->pos_w] + pp[R->pos_l];
                    if (lw > C->maxlogweight) C->maxlogweight = lw;
                    if (C->maxlogweight > R->maxlogweight_global) R->maxlogweight_global = C->maxlogweight;
                }
            }
        }
        free(pt); free(pp);
    }
}

static void thin_equals(rti_t *R, ptarr *eq, double maxlw, uint32_t list)
{   /* run_time_info.f90:976-997 */
    int i = 0;
    while (i < eq->n) {
        double *row = eq->a + (size_t)i * R->np;
        if (row[0] < maxlw) {
            if (bernoulli_post(R, exp(row[0] - maxlw), eq->uid[i], list)) { row[0] = maxlw; i++; }
            else pa_del(eq, i, NULL, NULL);
        } else i++;
    }
}

static void update_posteriors(rti_t *R)
{   /* run_time_info.f90:955-1066 */
    const pc_settings *s = R->s;
    R->post_round++;
    clean_phantoms(R);
    if (s->equals) {
        thin_equals(R, &R->equals_global, R->maxlogweight_global, 0);
        if (s->cluster_posteriors)
            for (int c = 0; c < R->ncluster; ++c) thin_equals(R, &R->cl[c].equals, R->cl[c].maxlogweight, 1);
    }
    double *ep = (double *)malloc(sizeof(double) * R->np);
    for (int c = 0; c < R->ncluster; ++c) {
        cluster_t *C = &R->cl[c];
        for (int i = 0; i < C->pstack.n; ++i) {
            const double *st = C->pstack.a + (size_t)i * R->npost;
            if (s->equals) {
                if (bernoulli_post(R, exp(st[R->pos_w] + st[R->pos_l] - R->maxlogweight_global), C->pstack.uid[i], 0)) {
                    ep[0] = R->maxlogweight_global; ep[1] = -2 * st[R->pos_l];
                    memcpy(ep + 2, st + R->pos_p0, sizeof(double) * (R->D + R->nDer));
                    pa_add(&R->equals_global, ep, C->pstack.uid[i]);
                }
                if (s->cluster_posteriors &&
                    bernoulli_post(R, exp(st[R->pos_w] + st[R->pos_l] - C->maxlogweight), C->pstack.uid[i], 1)) {
                    ep[0] = C->maxlogweight; ep[1] = -2 * st[R->pos_l];
                    memcpy(ep + 2, st + R->pos_p0, sizeof(double) * (R->D + R->nDer));
                    pa_add(&C->equals, ep, C->pstack.uid[i]);
                }
            }
            if (s->posteriors) {
                pa_add(&R->posterior_global, st, 0);
                if (s->cluster_posteriors) pa_add(&C->posterior, st, 0);
            }
        }
        C->pstack.n = 0;
    }
    free(ep);
}

/* ENGINE RULE (settings.epoch_discard = 0, the default; B > 1 only -- at B = 1 no chain is in flight when a cluster ends).
 * The reference's farm discards every baby seeded before a change of the cluster list (nested_sampling.F90:313: the workers'
 * messages carry positional cluster indices).  A chain seeded in a cluster the change did not touch is as good a sample after it
 * as before: when cluster p ends -- it died (delete_cluster) or was split (add_cluster) -- only the chains seeded in p are lost;
 * the clusters behind p move up one place and the chains seeded in them follow.  epoch_discard = 1 restores the reference's rule. */
static void remap_chains(rti_t *R, int p)
{
    if (!R->w_cluster || R->s->epoch_discard || R->s->farm) return;
    for (int w = 0; w < R->w_n; ++w) {
        const int c = R->w_cluster[w];
        if (c == p) { R->w_cluster[w] = -1; R->w_epoch[w] = -1; }
        else if (c > p) R->w_cluster[w] = c - 1;
    }
}

static int delete_cluster(rti_t *R)
{   /* run_time_info.f90:507-598 */
    int p = -1;
    for (int c = 0; c < R->ncluster; ++c) if (R->cl[c].live.n == 0) { p = c; break; }
    if (p < 0) return 0;
    update_posteriors(R);
    R->ncluster_dead++;
    R->logZp_dead = (double *)realloc(R->logZp_dead, sizeof(double) * R->ncluster_dead);
    R->logZp2_dead = (double *)realloc(R->logZp2_dead, sizeof(double) * R->ncluster_dead);
    R->logZp_dead[R->ncluster_dead - 1] = R->cl[p].logZp;
    R->logZp2_dead[R->ncluster_dead - 1] = R->cl[p].logZp2;
    R->nposterior_dead_tot += R->cl[p].posterior.n;
    R->nequals_dead_tot += R->cl[p].equals.n;
    cluster_free(&R->cl[p]);
    for (int c = p; c < R->ncluster - 1; ++c) R->cl[c] = R->cl[c + 1];
    /* compact XpXq */
    int n = R->ncluster;
    for (int a = 0, na = 0; a < n; ++a) {
        if (a == p) continue;
        for (int b = 0, nb = 0; b < n; ++b) {
            if (b == p) continue;
            double v = XQ(R, a, b);
            XQ(R, na, nb) = v;
            nb++;
        }
        na++;
    }
    R->ncluster--;
    remap_chains(R, p);
    return 1;
}

static void calculate_covmats(rti_t *R)
{   /* run_time_info.f90:601-641 */
    for (int c = 0; c < R->ncluster; ++c) {
        cluster_t *C = &R->cl[c];
        pc_covmat(C->live.a, C->live.n, C->phantom.a, C->phantom.n, R->nTotal, R->D, C->cov);
        pc_cholesky(C->cov, R->D, C->chol);
    }
}

static void add_cluster(rti_t *R, int p, const int *labels, int nnew)
{   /* run_time_info.f90:303-505: split cluster p into nnew clusters appended at the end */
    const pc_settings *s = R->s;
    int nold = R->ncluster - 1, ncl_new = R->ncluster + nnew - 1;
    ensure_ccap(R, ncl_new + 1);
    cluster_t oldp = R->cl[p];
    /* save every cluster's phantoms in the OLD numbering (incl. p) */
    int nc_old = R->ncluster;
    ptarr *oldph = (ptarr *)malloc(sizeof(ptarr) * nc_old);
    for (int c = 0; c < nc_old; ++c) { oldph[c] = R->cl[c].phantom; pa_init(&R->cl[c].phantom, R->nTotal); }
    oldp.phantom = oldph[p];
    double logXp = oldp.logXp, logXp2 = XQ(R, p, p), logZp = oldp.logZp, logZp2 = oldp.logZp2,
           logZXp = oldp.logZXp, logZpXp = oldp.logZpXp;
    double *rowpq = (double *)malloc(sizeof(double) * (nold > 0 ? nold : 1));
    for (int q = 0, k = 0; q < R->ncluster; ++q) if (q != p) rowpq[k++] = XQ(R, p, q);
    /* old clusters keep their order at 0..nold-1 */
    for (int c = p; c < R->ncluster - 1; ++c) R->cl[c] = R->cl[c + 1];
    remap_chains(R, p);
    {
        int n = R->ncluster;
        double *tmp = (double *)malloc(sizeof(double) * (size_t)n * n);
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) tmp[(size_t)a * n + b] = XQ(R, a, b);
        for (int a = 0, na = 0; a < n; ++a) {
            if (a == p) continue;
            for (int b = 0, nb = 0; b < n; ++b) { if (b == p) continue; XQ(R, na, nb) = tmp[(size_t)a * n + b]; nb++; }
            na++;
        }
        free(tmp);
    }
    R->ncluster = ncl_new;
    for (int k = 0; k < nnew; ++k) {
        cluster_t *C = &R->cl[nold + k];
        cluster_init(R, C);
        memcpy(C->chol, oldp.chol, sizeof(double) * R->D * R->D);  /* reallocate keeps nothing; set below */
        memcpy(C->cov, oldp.cov, sizeof(double) * R->D * R->D);
    }
    /* 3) distribute live points */
    for (int i = 0; i < oldp.live.n; ++i)
        pa_add(&R->cl[nold + labels[i] - 1].live, oldp.live.a + (size_t)i * R->nTotal, oldp.live.uid[i]);
    find_min_loglikelihoods(R);
    /* 4) posterior points copied to each new cluster */
    for (int k = 0; k < nnew; ++k) {
        cluster_t *C = &R->cl[nold + k];
        for (int i = 0; i < oldp.posterior.n; ++i) pa_add(&C->posterior, oldp.posterior.a + (size_t)i * R->npost, 0);
        for (int i = 0; i < oldp.equals.n; ++i) pa_add(&C->equals, oldp.equals.a + (size_t)i * R->np, oldp.equals.uid[i]);
        C->maxlogweight = oldp.maxlogweight;
    }
    /* 4b) re-home ALL phantoms (old numbering order) by nearest live point */
    for (int c = 0; c < nc_old; ++c) {
        for (int i = 0; i < oldph[c].n; ++i) {
            const double *pt = oldph[c].a + (size_t)i * R->nTotal;
            int j = identify_cluster(R, pt);
            if (pt[R->l0] > R->cl[j].logLp) pa_add(&R->cl[j].phantom, pt, oldph[c].uid[i]);
        }
    }
    /* 5) evidences and volumes split in proportion to (nlive+nphantom) */
    double *logni = (double *)malloc(sizeof(double) * nnew), *logni1 = (double *)malloc(sizeof(double) * nnew);
    for (int k = 0; k < nnew; ++k) {
        cluster_t *C = &R->cl[nold + k];
        logni[k] = log((double)(C->live.n + C->phantom.n) + 0.0);
        logni1[k] = log((double)(C->live.n + C->phantom.n) + 1.0);
    }
    double logn = pc_logsumexp(logni, nnew), logn1 = pc_logaddexp(logn, 0.0);
    for (int k = 0; k < nnew; ++k) {
        cluster_t *C = &R->cl[nold + k];
        C->logXp = logXp + logni[k] - logn;
        C->logZXp = logZXp + logni[k] - logn;
        C->logZp = logZp + logni[k] - logn;
        C->logZp2 = logZp2 + logni[k] + logni1[k] - logn - logn1;
        C->logZpXp = logZpXp + logni[k] + logni1[k] - logn - logn1;
    }
    for (int k = 0; k < nnew; ++k)
        for (int q = 0; q < nold; ++q) {
            XQ(R, nold + k, q) = rowpq[q] + logni[k] - logn;
            XQ(R, q, nold + k) = XQ(R, nold + k, q);
        }
    for (int a = 0; a < nnew; ++a)
        for (int b = 0; b < nnew; ++b)
            XQ(R, nold + a, nold + b) = (a == b) ? logXp2 + logni[a] + logni1[a] - logn - logn1
                                                  : logXp2 + logni[a] + logni[b] - logn - logn1;
    for (int k = 0; k < nnew; ++k) {
        cluster_t *C = &R->cl[nold + k];
        for (int i = 0; i < C->equals.n; ++i) C->equals.a[(size_t)i * R->np + 1] += 0.0; /* p_2l untouched */
        /* run_time_info.f90:500-503 shifts column pos_l (index 2 in Fortran = p_2l for equals) */
        for (int i = 0; i < C->equals.n; ++i) C->equals.a[(size_t)i * R->np + R->pos_l] += C->logZp - logZp;
        for (int i = 0; i < C->posterior.n; ++i) C->posterior.a[(size_t)i * R->npost + R->pos_l] += C->logZp - logZp;
    }
    (void)s;
    free(logni); free(logni1); free(rowpq);
    for (int c = 0; c < nc_old; ++c) pa_free(&oldph[c]);
    free(oldph);
    oldp.phantom.a = NULL; oldp.phantom.uid = NULL;
    pa_free(&oldp.live); pa_free(&oldp.pstack); pa_free(&oldp.posterior); pa_free(&oldp.equals);
    free(oldp.chol); free(oldp.cov);
}

static int do_clustering(rti_t *R)
{   /* clustering.f90:253-324 */
    int found = 0, nold = R->ncluster, ic = 0;
    while (ic < nold) {
        if (ic >= R->ncluster) break;
        int n = R->cl[ic].live.n;
        if (n > 2) {
            double *S = (double *)malloc(sizeof(double) * (size_t)n * n);
            int *lab = (int *)malloc(sizeof(int) * n);
            pc_similarity(R->cl[ic].live.a, n, R->nTotal, R->D, S);
            int num = pc_nn_clustering(S, n, lab);
            if (num > 1) { found = 1; add_cluster(R, ic, lab, num); }
            else ic++;
            free(S); free(lab);
        } else ic++;
    }
    return found;
}

static double live_logZ(const rti_t *R)
{   /* run_time_info.f90:683-709 */
    double v = R->s->logzero;
    for (int c = 0; c < R->ncluster; ++c) {
        const cluster_t *C = &R->cl[c];
        if (C->live.n > 0) {
            double m = C->live.a[R->l0];
            for (int i = 1; i < C->live.n; ++i) { double t = C->live.a[(size_t)i * R->nTotal + R->l0]; if (t > m) m = t; }
            double sum = 0.0;
            for (int i = 0; i < C->live.n; ++i) sum += exp(C->live.a[(size_t)i * R->nTotal + R->l0] - m);
            pc_logincexp(&v, m + log(sum) - log((double)C->live.n + 0.0) + C->logXp);
        }
    }
    return v;
}

static int more_samples_needed(const rti_t *R)
{   /* nested_sampling.F90:514-543 */
    const pc_settings *s = R->s;
    if (s->max_ndead == 0) return 0;
    if (s->max_ndead > 0 && R->dead.n >= s->max_ndead) return 0;
    if (s->precision_criterion > 0 && live_logZ(R) < log(s->precision_criterion) + R->logZ) return 0;
    return 1;
}

static void calculate_point(rti_t *R, double *pt, long *nlike)
{   /* calculate.f90:6-50 */
    int D = R->D, outside = 0;
    for (int d = 0; d < D; ++d) if (pt[d] < 0.0 || pt[d] > 1.0) outside = 1;
    double logL;
    if (outside) {
        for (int d = 0; d < D; ++d) pt[R->p0 + d] = 0.0;
        logL = R->s->logzero;
    } else {
        prior_eval(R->prior, pt, pt + R->p0, D);
        logL = pc_like_eval(R->like, pt + R->p0, D, pt + R->d0, R->nDer);
    }
    if (logL > R->s->logzero) (*nlike)++;
    pt[R->l0] = logL;
}

/* ---- directions ------------------------------------------------------------ */
static void generate_nhats(rti_t *R, uint32_t batch, uint32_t chain, double *nh /* [nr][D] */, int *speeds /* [nr] grade of each */)
{   /* chordal_sampling.f90:94-145 (single grade) + random_utils.F90:381-437, 276-298, 505-532 */
    int D = R->D, nr = R->s->num_repeats;
    double *basis = (double *)malloc(sizeof(double) * (size_t)D * D);
    double *raw = (double *)calloc((size_t)nr * D, sizeof(double));
    uint32_t e = 0;                 /* running index inside the (batch, chain) stream of PC_DOM_NHAT */
    int col0 = 0;
    for (int g = 0; g < R->ngrade; ++g) {
        /* grade g moves the parameters from its first one to the last: an orthonormal basis of that subspace */
        const int off = R->g_off[g], Dg = D - off, nrg = R->g_nr[g];
        const int nbases = (nrg + Dg - 1) / Dg;
        for (int b = 0; b < nbases; ++b) {
            for (int i = 0; i < Dg; ++i) {
                double *v = basis + (size_t)i * Dg, n2 = 0.0;
                do {    /* random_direction: gaussian deviates by inverse CDF, retry while |v|=0 */
                    n2 = 0.0;
                    for (int d = 0; d < Dg; ++d) {
                        v[d] = pc_inv_normal_cdf(pc_rng_u(&R->rng, PC_DOM_NHAT, batch, chain, e++));
                        n2 += v[d] * v[d];
                    }
                } while (n2 <= 0.0);
                double nrm = sqrt(n2);
                for (int d = 0; d < Dg; ++d) v[d] = v[d] / nrm;
                for (int j = 0; j < i; ++j) {   /* Gram-Schmidt against the finished vectors, in order */
                    const double *q = basis + (size_t)j * Dg;
                    double dot = 0.0;
                    for (int d = 0; d < Dg; ++d) dot += v[d] * q[d];
                    for (int d = 0; d < Dg; ++d) v[d] = v[d] - dot * q[d];
                }
                n2 = 0.0;
                for (int d = 0; d < Dg; ++d) n2 += v[d] * v[d];
                nrm = sqrt(n2);
                for (int d = 0; d < Dg; ++d) v[d] = v[d] / nrm;
            }
            for (int i = 0; i < Dg; ++i) {
                int col = b * Dg + i;
                if (col < nrg) memcpy(raw + (size_t)(col0 + col) * D + off, basis + (size_t)i * Dg, sizeof(double) * Dg);
            }
        }
        col0 += nrg;
    }
    /* deck = 1..nr, first stays, the rest Fisher-Yates shuffled from the top */
    int *deck = (int *)malloc(sizeof(int) * nr);
    for (int i = 0; i < nr; ++i) deck[i] = i;
    int n = nr - 1;
    for (int i = n; i >= 1; --i) {
        double u = pc_rng_u(&R->rng, PC_DOM_SHUFFLE, batch, chain, (uint32_t)i);
        int j = (int)ceil(u * i);
        if (j < 1) j = 1;
        if (j > i) j = i;
        int t = deck[i]; deck[i] = deck[j]; deck[j] = t;   /* deck(2:)(i) <-> deck(2:)(j) */
    }
    for (int i = 0; i < nr; ++i) {
        memcpy(nh + (size_t)i * D, raw + (size_t)deck[i] * D, sizeof(double) * D);
        if (speeds) { int g = 0, c = R->g_nr[0]; while (deck[i] >= c) c += R->g_nr[++g]; speeds[i] = g; }
    }
    free(basis); free(raw); free(deck);
}

static void slice_sample(rti_t *R, uint32_t batch, uint32_t chain, int islice, double logLb,
                         const double *nhat, const double *x0, double w, long *nlike, double *baby)
{   /* chordal_sampling.f90:163-273 */
    int D = R->D, nT = R->nTotal;
    double *Rt = (double *)malloc(sizeof(double) * nT), *Lf = (double *)malloc(sizeof(double) * nT);
    memcpy(Rt, x0, sizeof(double) * nT); memcpy(Lf, x0, sizeof(double) * nT); memcpy(baby, x0, sizeof(double) * nT);
    uint32_t base = (uint32_t)islice * PC_SLICE_STRIDE, k = 0;
    double u = pc_rng_u(&R->rng, PC_DOM_SLICE, batch, chain, base + k++);
    for (int d = 0; d < D; ++d) {
        Lf[d] = x0[d] - u * w * nhat[d];
        Rt[d] = x0[d] + (1 - u) * w * nhat[d];
    }
    calculate_point(R, Rt, nlike);
    calculate_point(R, Lf, nlike);
    int istep = 0;
    while (Rt[R->l0] >= logLb && Rt[R->l0] > R->s->logzero) {
        istep++;
        for (int d = 0; d < D; ++d) Rt[d] = x0[d] + nhat[d] * w * istep;
        calculate_point(R, Rt, nlike);
    }
    istep = 0;
    while (Lf[R->l0] >= logLb && Lf[R->l0] > R->s->logzero) {
        istep++;
        for (int d = 0; d < D; ++d) Lf[d] = x0[d] - nhat[d] * w * istep;
        calculate_point(R, Lf, nlike);
    }
    int done = 0;
    for (istep = 0; istep <= 100; ++istep) {
        double dl = 0.0, dr = 0.0;
        for (int d = 0; d < D; ++d) { double t = x0[d] - Lf[d]; dl += t * t; }
        for (int d = 0; d < D; ++d) { double t = x0[d] - Rt[d]; dr += t * t; }
        dl = sqrt(dl); dr = sqrt(dr);
        u = pc_rng_u(&R->rng, PC_DOM_SLICE, batch, chain, base + k++);
        double t = u * (dr + dl) - dl;
        for (int d = 0; d < D; ++d) baby[d] = x0[d] + t * nhat[d];
        calculate_point(R, baby, nlike);
        if (baby[R->l0] < logLb || baby[R->l0] <= R->s->logzero) {
            double dot = 0.0;
            for (int d = 0; d < D; ++d) dot += (baby[d] - x0[d]) * nhat[d];
            if (dot > 0.0) memcpy(Rt, baby, sizeof(double) * nT);
            else memcpy(Lf, baby, sizeof(double) * nT);
        } else { done = 1; break; }
    }
    if (!done) baby[R->l0] = R->s->logzero;   /* "Non deterministic loglikelihood" */
    free(Rt); free(Lf);
}

static long slice_sampling(rti_t *R, uint32_t batch, uint32_t chain, const double *seed_point,
                           const double *chol, double logLb, double *babies, double *nhats_out, long *nlike_g)
{   /* chordal_sampling.f90:7-92 */
    int D = R->D, nr = R->s->num_repeats, nT = R->nTotal;
    long nlike = 0;
    double *nh = (double *)malloc(sizeof(double) * (size_t)nr * D);
    double *v = (double *)malloc(sizeof(double) * D);
    double *prev = (double *)malloc(sizeof(double) * nT);
    memcpy(prev, seed_point, sizeof(double) * nT);
    int *speeds = (int *)malloc(sizeof(int) * nr);
    generate_nhats(R, batch, chain, nh, speeds);
    for (int i = 0; i < nr; ++i) {
        const long nl0 = nlike;
        /* nhat = L . nhat_i  (matmul(cholesky,nhats), chordal_sampling.f90:73) */
        for (int a = 0; a < D; ++a) {
            double t = 0.0;
            for (int b = 0; b < D; ++b) t += chol[(size_t)a * D + b] * nh[(size_t)i * D + b];
            v[a] = t;
        }
        double w = 0.0;
        for (int d = 0; d < D; ++d) w += v[d] * v[d];
        w = sqrt(w);
        for (int d = 0; d < D; ++d) v[d] = v[d] / w;
        if (nhats_out) memcpy(nhats_out + (size_t)i * D, v, sizeof(double) * D);
        w = w * 3.0;
        slice_sample(R, batch, chain, i, logLb, v, prev, w, &nlike, babies + (size_t)i * nT);
        memcpy(prev, babies + (size_t)i * nT, sizeof(double) * nT);
        if (nlike_g) nlike_g[speeds[i]] += nlike - nl0;     /* chordal_sampling.f90:84 */
    }
    free(speeds);
    for (int i = 0; i < nr; ++i) babies[(size_t)i * nT + R->b0] = logLb;   /* nested_sampling.F90:260 */
    free(nh); free(v); free(prev);
    return nlike;
}

static void rti_setup(rti_t *R, const pc_settings *s, const pc_like *like, const pc_prior *prior)
{
    memset(R, 0, sizeof(*R));
    R->s_local = *s; R->s = &R->s_local; R->like = like; R->prior = prior;
    /* grades (generate.F90:303-309, deterministic branch: repeats = int(grade_frac)) */
    R->ngrade = 1; R->g_off[0] = 0; R->g_nr[0] = s->num_repeats;
    if (s->nGrade > 1) {
        R->ngrade = s->nGrade > 8 ? 8 : s->nGrade;
        int off = 0, tot = 0;
        for (int g = 0; g < R->ngrade; ++g) { R->g_off[g] = off; off += s->grade_dims[g]; R->g_nr[g] = (int)s->grade_frac[g]; tot += R->g_nr[g]; }
        R->s_local.num_repeats = tot;
    }
    s = R->s;
    R->D = s->nDims; R->nDer = s->nDerived;
    R->h0 = 0; R->p0 = R->D; R->d0 = 2 * R->D; R->b0 = 2 * R->D + R->nDer; R->l0 = R->b0 + 1;
    R->nTotal = R->l0 + 1;
    R->pos_X = 0; R->pos_l = 1; R->pos_w = 2; R->pos_Z = 3; R->pos_p0 = 4;
    R->npost = 4 + R->D + R->nDer; R->np = 2 + R->D + R->nDer;
    R->rng.key[0] = (uint32_t)s->seed; R->rng.key[1] = 0x504F4C59u; /* 'POLY' */
    R->rng.sequential = s->sequential_rng; R->rng.seq = 0; R->rng.post = 0;
    ensure_ccap(R, 4);
    R->ncluster = 1;
    cluster_init(R, &R->cl[0]);
    XQ(R, 0, 0) = 0.0;
    R->logZ = R->logZ2 = s->logzero;
    pa_init(&R->dead, R->nTotal);
    pa_init(&R->posterior_global, R->npost); pa_init(&R->equals_global, R->np);
    R->maxlogweight_global = s->logzero;
    R->logX_last_update = 0.0;
}

long pc_slice_chain(const pc_settings *s, const pc_like *like, const pc_prior *prior, pc_rng *rng,
                    uint32_t batch, uint32_t chain, const double *seed_point, const double *chol,
                    double logL, double *babies, double *nhats_out)
{
    rti_t R;
    rti_setup(&R, s, like, prior);
    R.rng = *rng;
    long n = slice_sampling(&R, batch, chain, seed_point, chol, logL, babies, nhats_out, NULL);
    *rng = R.rng;
    cluster_free(&R.cl[0]); free(R.cl); free(R.XpXq);
    return n;
}

void pc_settings_default(pc_settings *s, int nDims, int nDerived)
{   /* c_interface.cpp:6-39 defaults */
    memset(s, 0, sizeof(*s));
    s->nDims = nDims; s->nDerived = nDerived; s->nlive = 500; s->num_repeats = 5 * nDims;
    s->nprior = -1; s->nfail = -1; s->do_clustering = 0; s->precision_criterion = 0.001;
    s->logzero = -1e30; s->max_ndead = -1; s->boost_posterior = 0.0;
    s->compression_factor = 0.36787944117144233; s->seed = -1; s->batch = 1;
}

static void generate_seed(rti_t *R, uint32_t batch, uint32_t chain, int *cluster, const double **pt)
{   /* generate.F90:19-55, random_utils.F90:548-576, :215-228 */
    int nc = R->ncluster;
    double *probs = (double *)malloc(sizeof(double) * nc);
    double lse = logsumexp_Xp(R), norm = 0.0;
    for (int c = 0; c < nc; ++c) { probs[c] = exp(R->cl[c].logXp - lse); norm += probs[c]; }
    double u = pc_rng_u(&R->rng, PC_DOM_SEED, batch, chain, 0);
    double cdf = 0.0; int sel = nc - 1;
    for (int c = 0; c < nc; ++c) { cdf += probs[c] / norm; if (u < cdf) { sel = c; break; } }
    double u2 = pc_rng_u(&R->rng, PC_DOM_SEED, batch, chain, 1);
    int n = R->cl[sel].live.n;
    int idx = (int)ceil(u2 * n);
    if (idx < 1) idx = 1;
    if (idx > n) idx = n;
    *cluster = sel;
    *pt = R->cl[sel].live.a + (size_t)(idx - 1) * R->nTotal;
    free(probs);
}

int pc_oracle_run(const pc_settings *s, const pc_like *like, const pc_prior *prior, pc_result *out)
{
    rti_t Rs, *R = &Rs;
    rti_setup(R, s, like, prior);
    s = R->s;                                   /* num_repeats is the total over the grades from here on */
    int D = R->D, nT = R->nTotal, nr = s->num_repeats;
    int B = s->batch > 1 ? s->batch : 1;
    memset(out, 0, sizeof(*out));
    if (nr < 1) { fprintf(stderr, "pc_oracle: You need to set num_repeats\n"); return 1; } /* settings.f90:216 */

    /* ---- GenerateLivePoints, linear mode (generate.F90:150-183) ---- */
    int nprior = s->nprior <= 0 ? s->nlive : s->nprior;
    const int farm = s->farm && s->sequential_rng && B > 1;
    /* The farm's administrator (generate.F90:187-252) draws the coordinates itself and has one request out to every worker: when the
     * nprior-th point comes back B - 1 more are on their way, and they are added as well (the points are put in the order they were
     * drawn in, :236; what exceeds nlive dies before the first chain, nested_sampling.F90:201-205).  Restated for likelihoods that
     * accept every prior sample (all of the built-in ones): with rejections the count would depend on the order of arrival. */
    if (farm) nprior += B - 1;
    long nlike = 0; uint32_t attempt = 0;
    double *pt = (double *)calloc(nT, sizeof(double));
    while (R->cl[0].live.n < nprior) {
        for (int d = 0; d < D; ++d) pt[d] = pc_rng_u(&R->rng, PC_DOM_LIVEGEN, 0, attempt, (uint32_t)d);
        calculate_point(R, pt, &nlike);
        pt[R->b0] = s->logzero;
        if (pt[R->l0] > s->logzero) pa_add(&R->cl[0].live, pt, ((uint64_t)0xFFFFFFFFu << 32) | attempt);
        attempt++;
    }
    if (s->time_speeds_draw) {   /* generate.F90:388-393: one more prior sample is drawn and evaluated */
        long dummy = 0;
        do {
            for (int d = 0; d < D; ++d) pt[d] = pc_rng_u(&R->rng, PC_DOM_LIVEGEN, 1, attempt, (uint32_t)d);
            calculate_point(R, pt, &dummy);
            attempt++;
        } while (!(pt[R->l0] > s->logzero));
    }
    R->nlike = nlike;
    find_min_loglikelihoods(R);
    /* thin_posterior: generate.F90:311-316 */
    R->thin_posterior = (s->boost_posterior < 0.0) ? 1.0 : (s->boost_posterior + 0.0) / ((double)nr + 0.0);
    while (R->cl[0].live.n > s->nlive) delete_outermost_point(R);   /* nested_sampling.F90:201-205 */

    int nfail = s->nfail <= 0 ? s->nlive : s->nfail, failures = 0;
    double *nursery = (double *)malloc(sizeof(double) * (size_t)B * nr * nT);
    uint64_t *uids = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)B * nr);
    int *wcluster = (int *)malloc(sizeof(int) * B), *wepoch = (int *)malloc(sizeof(int) * B);
    long *wnlike = (long *)malloc(sizeof(long) * B), *wnlike_g = (long *)calloc((size_t)B * 8, sizeof(long));
    int i_nursery = 0, admin_epoch = 0;
    uint32_t batch = 0;
    long niter = 0;
    pc_rng *wrng = NULL;
    if (farm) {
        /* worker w = rank w + 1 of the farm: its own running stream (ref_rng_shim.c pc_shim_set_rank), of which time_speeds has taken
         * one prior sample on every rank before the first chain (generate.F90:388-393) */
        wrng = (pc_rng *)malloc(sizeof(pc_rng) * B);
        for (int w = 0; w < B; ++w) {
            wrng[w] = R->rng; wrng[w].key[1] = 0x504F4C59u ^ ((uint32_t)(w + 1) * 0x9E3779B9u); wrng[w].seq = 0;
            if (s->time_speeds_draw) {
                pc_rng keep_rng = R->rng; long dummy = 0;
                R->rng = wrng[w];
                do { for (int d = 0; d < D; ++d) pt[d] = pc_rng_u(&R->rng, PC_DOM_LIVEGEN, 1, 0, (uint32_t)d); calculate_point(R, pt, &dummy); } while (!(pt[R->l0] > s->logzero));
                wrng[w] = R->rng; R->rng = keep_rng;
            }
        }
    }

    while (more_samples_needed(R) && failures <= nfail) {
        int cluster_id; const double *seedpt;
        if (R->rng.sequential) {   /* nested_sampling.F90:245 draws a seed at the top of every iteration */
            generate_seed(R, batch, 0, &cluster_id, &seedpt);
        }
        if (i_nursery == 0) {      /* nested_sampling.F90:259 (linear) / :266-281 (synchronous farm) */
            for (int w = 0; w < B; ++w) {
                if (!(R->rng.sequential && B == 1)) generate_seed(R, batch, (uint32_t)w, &cluster_id, &seedpt);
                wcluster[w] = cluster_id; wepoch[w] = admin_epoch;
                for (int g = 0; g < 8; ++g) wnlike_g[(size_t)w * 8 + g] = 0;
                pc_rng admin_rng = R->rng;
                if (farm) R->rng = wrng[w];                 /* (the chain is worker w's: its draws come from its stream) */
                wnlike[w] = slice_sampling(R, batch, (uint32_t)w, seedpt, R->cl[cluster_id].chol,
                                           R->cl[cluster_id].logLp, nursery + (size_t)w * nr * nT, NULL, wnlike_g + (size_t)w * 8);
                if (farm) { wrng[w] = R->rng; R->rng = admin_rng; }
                for (int i = 0; i < nr; ++i) uids[(size_t)w * nr + i] = ((uint64_t)batch << 32) | (uint32_t)(w * nr + i);
            }
            i_nursery = B; batch++; out->nbatches++;
        }
        int w = i_nursery - 1; i_nursery--;
        R->w_cluster = wcluster; R->w_epoch = wepoch; R->w_n = i_nursery;
        R->nlike += wnlike[w];
        for (int g = 0; g < 8; ++g) R->nlike_g[g] += wnlike_g[(size_t)w * 8 + g];
        niter++;
        if (wepoch[w] == admin_epoch) {
            if (replace_point(R, nursery + (size_t)w * nr * nT, uids + (size_t)w * nr, nr, wcluster[w])) failures = 0;
            else failures++;
            double lx = logsumexp_Xp(R);
            int update = lx <= R->logX_last_update + log(s->compression_factor);
            if (update) {
                R->logX_last_update = lx;
                update_posteriors(R);
            }
            if (delete_cluster(R) && (s->epoch_discard || s->farm)) admin_epoch++;
            if (R->ncluster == 0) break;
            if (update) {
                if (s->do_clustering && do_clustering(R) && (s->epoch_discard || s->farm)) admin_epoch++;
                calculate_covmats(R);
            }
        }
    }
    R->w_cluster = NULL; R->w_epoch = NULL; R->w_n = 0;
    /* final live points (before kill-off) */
    out->nlive_final = total_live(R);
    out->live = (double *)malloc(sizeof(double) * (size_t)(out->nlive_final > 0 ? out->nlive_final : 1) * nT);
    for (int c = 0, k = 0; c < R->ncluster; ++c)
        for (int i = 0; i < R->cl[c].live.n; ++i, ++k)
            memcpy(out->live + (size_t)k * nT, R->cl[c].live.a + (size_t)i * nT, sizeof(double) * nT);
    int ncluster_at_end = R->ncluster;
    /* per-cluster evidences in stats order: live clusters at end + dead clusters */
    /* nested_sampling.F90:381-384 */
    while (R->ncluster > 0) { delete_outermost_point(R); delete_cluster(R); }
    update_posteriors(R);
    /* run_time_info.f90:652-678 */
    out->logZ = 2 * R->logZ - 0.5 * R->logZ2;
    if (out->logZ < -HUGE_D) out->logZ = -HUGE_D;
    out->varlogZ = R->logZ2 - 2 * R->logZ;
    out->ndead = R->dead.n; out->nlike = R->nlike; out->ncluster = ncluster_at_end;
    for (int g = 0; g < 8; ++g) out->nlike_grade[g] = R->nlike_g[g];
    if (R->ngrade == 1) out->nlike_grade[0] = R->nlike;       /* the prior samples belong to grade 1 (generate.F90:294) */
    else out->nlike_grade[0] += nprior;
    out->ncluster_dead = R->ncluster_dead; out->niter = niter; out->nTotal = nT;
    out->dead = R->dead.a; out->logweights = R->logweights;
    out->nZp = R->ncluster_dead;
    out->logZp = (double *)malloc(sizeof(double) * (out->nZp > 0 ? out->nZp : 1));
    out->varlogZp = (double *)malloc(sizeof(double) * (out->nZp > 0 ? out->nZp : 1));
    for (int c = 0; c < R->ncluster_dead; ++c) {
        out->logZp[c] = 2 * R->logZp_dead[c] - 0.5 * R->logZp2_dead[c];
        out->varlogZp[c] = R->logZp2_dead[c] - 2 * R->logZp_dead[c];
    }
    out->nposterior_global = R->posterior_global.n; out->nequals_global = R->equals_global.n;
    {   /* rows as write_posterior_file prints them, before its normalisation by the largest weight */
        const int npar = R->D + R->nDer;
        out->post_rows = (double *)malloc(sizeof(double) * (size_t)(R->posterior_global.n + 1) * (2 + npar));
        for (int i = 0; i < R->posterior_global.n; ++i) {
            const double *st = R->posterior_global.a + (size_t)i * R->npost;
            double *o = out->post_rows + (size_t)i * (2 + npar);
            o[0] = st[R->pos_w] + st[R->pos_l]; o[1] = st[R->pos_l];
            memcpy(o + 2, st + R->pos_p0, sizeof(double) * npar);
        }
        out->equal_rows = (double *)malloc(sizeof(double) * (size_t)(R->equals_global.n + 1) * (1 + npar));
        for (int i = 0; i < R->equals_global.n; ++i)
            memcpy(out->equal_rows + (size_t)i * (1 + npar), R->equals_global.a + (size_t)i * R->np + 1, sizeof(double) * (1 + npar));
        out->maxlogweight = R->maxlogweight_global;
    }
    /* posterior mean / variance of theta from dead points: w_i = logweight_i + logL_i */
    out->post_mean = (double *)calloc(D, sizeof(double)); out->post_var = (double *)calloc(D, sizeof(double));
    {
        double m = -HUGE_D;
        for (long i = 0; i < out->ndead; ++i) {
            double lw = out->logweights[i] + out->dead[(size_t)i * nT + R->l0];
            if (out->logweights[i] > s->logzero && lw > m) m = lw;
        }
        double sw = 0.0;
        for (long i = 0; i < out->ndead; ++i) {
            if (!(out->logweights[i] > s->logzero)) continue;
            double wgt = exp(out->logweights[i] + out->dead[(size_t)i * nT + R->l0] - m);
            sw += wgt;
            for (int d = 0; d < D; ++d) {
                double th = out->dead[(size_t)i * nT + R->p0 + d];
                out->post_mean[d] += wgt * th; out->post_var[d] += wgt * th * th;
            }
        }
        for (int d = 0; d < D; ++d) {
            out->post_mean[d] /= sw;
            out->post_var[d] = out->post_var[d] / sw - out->post_mean[d] * out->post_mean[d];
        }
    }
    free(pt); free(nursery); free(uids); free(wcluster); free(wepoch); free(wnlike);
    pa_free(&R->posterior_global); pa_free(&R->equals_global);
    free(R->cl); free(R->XpXq); free(R->logZp_dead); free(R->logZp2_dead);
    return 0;
}

void pc_result_free(pc_result *r)
{
    free(r->dead); free(r->logweights); free(r->live); free(r->logZp); free(r->varlogZp);
    free(r->post_mean); free(r->post_var); free(r->post_rows); free(r->equal_rows);
    memset(r, 0, sizeof(*r));
}

/* ---- evidence replay: static-nlive recursion from (logL, birth) ---- */
static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
void pc_evidence_replay(const double *logL, const double *birth, long n, double *logZ, double *varlogZ)
{
    /* n(L_i) = #{birth < L_i} - #{death < L_i} over all recorded points, death order = ascending logL */
    double *d = (double *)malloc(sizeof(double) * n), *b = (double *)malloc(sizeof(double) * n);
    memcpy(d, logL, sizeof(double) * n); memcpy(b, birth, sizeof(double) * n);
    qsort(d, n, sizeof(double), cmp_d); qsort(b, n, sizeof(double), cmp_d);
    double Z = -1e30, Z2 = -1e30, X = 0.0, ZX = -1e30, XX = 0.0;
    const double log2 = log(2.0);
    long jb = 0;
    for (long i = 0; i < n; ++i) {
        while (jb < n && b[jb] < d[i]) jb++;
        double np = (double)(jb - i), L = d[i];
        double lognp = log(np), lognp1 = log(np + 1.0), lognp2 = log(np + 2.0);
        pc_logincexp(&Z, X + L - lognp1);
        X = X + lognp - lognp1;
        logincexp2(&Z2, log2 + ZX + L - lognp1, log2 + XX + 2 * L - lognp1 - lognp2);
        ZX = ZX + lognp - lognp1;
        pc_logincexp(&ZX, XX + L + lognp - lognp1 - lognp2);
        XX = XX + lognp - lognp2;
    }
    *logZ = 2 * Z - 0.5 * Z2; *varlogZ = Z2 - 2 * Z;
    free(d); free(b);
}

/* random_inverse_covmat (random_utils.F90:581-614): random orthonormal eigenbasis,
 * eigen-sigmas  sigma * 1e-2^((j-1)/(D-1)).  Keyed stream so every rank / the engine
 * can rebuild the same matrix from (seed, D). */
void pc_random_invcov(uint32_t seed, int D, double sigma, double *invcov, double *logdet)
{
    uint32_t key[2] = { seed, 0x434F5652u };   /* 'COVR' */
    double *E = (double *)malloc(sizeof(double) * (size_t)D * D);
    for (int i = 0; i < D; ++i) {
        double *v = E + (size_t)i * D, n2 = 0.0;
        for (int d = 0; d < D; ++d) {
            v[d] = pc_inv_normal_cdf(pc_uniform_keyed(key, PC_DOM_NHAT, 0, 0, (uint32_t)(i * D + d)));
            n2 += v[d] * v[d];
        }
        double nrm = sqrt(n2);
        for (int d = 0; d < D; ++d) v[d] /= nrm;
        for (int j = 0; j < i; ++j) {
            const double *q = E + (size_t)j * D; double dot = 0.0;
            for (int d = 0; d < D; ++d) dot += v[d] * q[d];
            for (int d = 0; d < D; ++d) v[d] -= dot * q[d];
        }
        n2 = 0.0; for (int d = 0; d < D; ++d) n2 += v[d] * v[d];
        nrm = sqrt(n2);
        for (int d = 0; d < D; ++d) v[d] /= nrm;
    }
    *logdet = 0.0;
    for (int a = 0; a < D * D; ++a) invcov[a] = 0.0;
    for (int j = 0; j < D; ++j) {
        double ev = (D > 1) ? sigma * pow(1e-2, ((double)j) / ((double)D - 1.0)) : sigma;
        *logdet += 2.0 * log(ev);
        const double *v = E + (size_t)j * D;   /* eigenvector j = basis(:,j) */
        for (int a = 0; a < D; ++a)
            for (int b = 0; b < D; ++b) invcov[(size_t)a * D + b] += v[a] * v[b] / (ev * ev);
    }
    free(E);
}
