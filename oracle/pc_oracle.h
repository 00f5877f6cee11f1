/*
 * pc_oracle.h -- CPU restatement (plain C) of the PolyChordLite slice-sampling
 * nested-sampling hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 * The shipped engine (polychordlite_amd/csrc) never links or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Pinning status: see oracle/README.md --
 *  (1) deterministic unit vectors produced by the reference's own Fortran
 *      modules (tests/golden/ref_units.json),
 *  (2) evidence replay of the reference's dead-birth output,
 *  (3) FULL-RUN BIT-LEVEL PINNING: the reference (built by oracle/Makefile into
 *      oracle/_ref) is run with its compiler RNG (`random_number`) fed from the
 *      same sequential Philox stream this oracle uses in `sequential` mode;
 *      ndead / nlike / ncluster agree exactly and logZ to ~1e-12.
 */
#ifndef PC_ORACLE_H
#define PC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG: Philox4x32-10 counter based; see pc_oracle.c ------------------- */
enum { PC_DOM_LIVEGEN = 0, PC_DOM_SEED = 1, PC_DOM_NHAT = 2, PC_DOM_SHUFFLE = 3,
       PC_DOM_SLICE = 4, PC_DOM_PHANTOM = 5, PC_DOM_POST = 6, PC_DOM_SEQ = 0xFFFF };
#define PC_SLICE_STRIDE 128 /* uniforms reserved per slice in PC_DOM_SLICE */

typedef struct {
    uint32_t key[2];
    int sequential;   /* 1: ignore (domain,stream,index): one running stream (reference order) */
    uint64_t seq;     /* running index of the sequential stream */
    uint64_t post;    /* running index inside PC_DOM_POST in keyed mode */
} pc_rng;

void pc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* uniform in (0,1): index idx of stream (shi,slo) of domain dom */
double pc_uniform_keyed(const uint32_t key[2], uint32_t dom, uint32_t shi, uint32_t slo, uint32_t idx);
double pc_rng_u(pc_rng *r, uint32_t dom, uint32_t shi, uint32_t slo, uint32_t idx);

/* ---- numerics units (utils.F90) ------------------------------------------ */
double pc_inv_normal_cdf(double p);                       /* utils.F90:806-966 (AS241 PPND16) */
double pc_logsumexp(const double *v, int n);              /* utils.F90:362-374 */
double pc_logaddexp(double a, double b);                  /* utils.F90:377-389 */
void   pc_logincexp(double *a, double b);                 /* utils.F90:417-439 */
/* lower Cholesky, row-major L[j*n+i] (j>=i); fallback identity*sqrt(trace) */
void   pc_cholesky(const double *a, int n, double *L);    /* utils.F90:621-649 */
/* covariance of the union of two row-major point sets (cube coords, stride w) */
void   pc_covmat(const double *live, int nlive, const double *ph, int nph, int w, int D,
                 double *cov);                            /* run_time_info.f90:601-641 */
/* similarity matrix of n points of dimension D (row-major x[n][stride]) */
void   pc_similarity(const double *x, int n, int stride, int D, double *S); /* calculate.f90:94-109 */
/* recursive kNN clustering on an n x n similarity matrix; labels 1..num (returned) */
int    pc_nn_clustering(const double *S, int n, int *labels);              /* clustering.f90:15-97 */
void   pc_compute_knn(const double *S, int n, int k, int *knn);            /* clustering.f90:134-174 */

/* evidence replay from (logL, birth) of every dead point in death order, static nlive.
 * Returns logZ, var via the reference recursion (run_time_info.f90:211-296, 652-678). */
void   pc_evidence_replay(const double *logL, const double *birth, long n, double *logZ, double *varlogZ);

/* ---- built-in likelihoods / priors (likelihoods/examples, priors.f90:40-55) */
enum { PC_LIKE_CALLBACK = 0, PC_LIKE_GAUSSIAN = 1, PC_LIKE_RASTRIGIN = 2,
       PC_LIKE_TWIN_GAUSSIAN = 3, PC_LIKE_CORR_GAUSSIAN = 4 };

typedef double (*pc_logl_fn)(const double *theta, int nDims, double *phi, int nDerived, void *ctx);
typedef void   (*pc_prior_fn)(const double *cube, double *theta, int nDims, void *ctx);

typedef struct {
    int kind;            /* PC_LIKE_* */
    double mu, sigma;    /* gaussian: mean / width (gaussian.f90:25-26); twin: sigma */
    const double *invcov;/* corr gaussian: row-major D x D inverse covariance */
    double logdetcov;    /* corr gaussian */
    const double *mean;  /* corr gaussian mean vector (D) */
    pc_logl_fn fn; void *ctx;   /* callback kind */
} pc_like;

typedef struct {
    int kind;            /* 0: callback, 1: uniform box */
    const double *lo, *hi;      /* uniform box bounds per dimension (NULL => [0,1]) */
    pc_prior_fn fn; void *ctx;
} pc_prior;

double pc_like_eval(const pc_like *L, const double *theta, int D, double *phi, int nDerived);
/* random_inverse_covmat (random_utils.F90:581-614) with a keyed RNG: fills invcov (row-major) */
void   pc_random_invcov(uint32_t seed, int D, double sigma, double *invcov, double *logdet);

/* ---- the sampler ------------------------------------------------------------ */
typedef struct {
    int nDims, nDerived;
    int nlive, num_repeats, nprior, nfail;
    int do_clustering;
    double precision_criterion, logzero;
    int max_ndead;
    double boost_posterior;
    int posteriors, equals, cluster_posteriors;
    double compression_factor;
    int n_nlives; const double *loglikes; const int *nlives;   /* dynamic nlive (optional) */
    int seed;
    int batch;          /* B: chains per synchronous nursery (reference nprocs-1); <=1 => linear mode */
    int sequential_rng; /* 1: one running stream consumed in the reference's program order */
    int time_speeds_draw; /* 1: mimic generate.F90:388-393 extra prior draw (sequential pinning) */
    /* fast/slow parameter grades (chordal_sampling.f90:94-145): nGrade <= 1 means one grade of nDims parameters.
     * Only the deterministic branch of generate.F90:303-309 is restated: every grade_frac > 1 is the number of
     * repeats of that grade (the other branch times the likelihood with the wall clock). */
    int nGrade; const int *grade_dims; const double *grade_frac;
    int farm;           /* 1 (with sequential_rng = 1, batch = B): the reference's synchronous FARM, nprocs - 1 = B (nested_sampling.F90:262-286,
                           generate.F90:187-252, :388-393): the administrator's running stream for the prior samples, every seed, the
                           posterior thinning; worker w's own running stream for its chains -- what oracle/_ref/ref_driver_mpi_inject
                           consumes under mpiexec -n B+1 (ref_rng_shim.c pc_shim_set_rank).  Forces epoch_discard = 1 (its rule). */
    int epoch_discard;  /* 1: nested_sampling.F90:313 as written (every chain in flight is lost when the cluster list changes); 0: the engine's
                           rule (pc_oracle.c remap_chains: only the chains of the cluster that ended are lost) */
} pc_settings;

typedef struct {
    double logZ, varlogZ;
    long ndead, nlike;
    int ncluster, ncluster_dead;
    long niter;                /* consumed nursery entries */
    long nbatches;
    int nTotal;
    /* dead points: rows of nTotal doubles [cube|theta|phi|birth|logL], and log-weights */
    double *dead; double *logweights;
    /* final live points at termination (before kill-off), rows of nTotal */
    double *live; int nlive_final;
    /* per-cluster evidences of dead clusters then live clusters at the end (stats file order) */
    double *logZp; double *varlogZp; int nZp;
    /* weighted posterior mean of theta (from dead points + logweights) */
    double *post_mean; double *post_var;
    long nposterior_global, nequals_global;
    long nlike_grade[8];       /* likelihood calls per grade (RTI%nlike) */
    /* the posterior arrays behind <root>.txt / <root>_equal_weights.txt (read_write.F90:479-617):
     * post_rows [nposterior_global][2 + nDims + nDerived] = log posterior weight (logweight + logL), logL, theta, phi;
     * equal_rows [nequals_global][1 + nDims + nDerived] = -2 logL, theta, phi; maxlogweight = RTI%maxlogweight_global */
    double *post_rows, *equal_rows; double maxlogweight;
} pc_result;

void pc_settings_default(pc_settings *s, int nDims, int nDerived);
int  pc_oracle_run(const pc_settings *s, const pc_like *like, const pc_prior *prior, pc_result *out);
void pc_result_free(pc_result *r);

/* exposed for kernel-level parity tests ---------------------------------------- */
/* One slice-sampling chain (chordal_sampling.f90:7-92).  seed/babies are rows of nTotal.
 * chol row-major lower.  Returns nlike.  (batch,chain) select the keyed streams. */
long pc_slice_chain(const pc_settings *s, const pc_like *like, const pc_prior *prior, pc_rng *rng,
                    uint32_t batch, uint32_t chain, const double *seed_point, const double *chol,
                    double logL, double *babies /* [num_repeats][nTotal] */, double *nhats_out /* [nr][D] or NULL */);

#ifdef __cplusplus
}
#endif
#endif
