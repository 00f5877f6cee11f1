/*
 * polychord_hip.h -- C ABI of libpolychord_hip.so, the MI355X-native nested-sampling engine
 * that is a drop-in behind PolyChordLite's own C boundary.
 *
 * Part 1 re-declares, symbol for symbol, what the reference exports from its Fortran
 * ISO_C_BINDING shim (reference src/polychord/interfaces.h:2-56, implemented in
 * src/polychord/interfaces.F90:285-436 and :496-519).  A caller linked against the
 * reference's libchord.so links against libpolychord_hip.so unchanged.
 *
 * Part 2 is the engine's own plain-C surface (no torch types, plain pointers and sizes): device
 * likelihood selectors, the batched engine entry used by bench.py / tests, and kernel-level entry
 * points for parity tests.
 */
#ifndef POLYCHORD_HIP_H
#define POLYCHORD_HIP_H
#include <stdbool.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ===== Part 1: the reference's C boundary ================================================== */

/* loglikelihood(theta, nDims, phi, nDerived) -> logL      (interfaces.h:3, interfaces.F90:291-300) */
typedef double (*polychord_loglike_fn)(double *theta, int nDims, double *phi, int nDerived);
/* prior(cube, theta, nDims): hypercube -> physical        (interfaces.h:4, interfaces.F90:302-309) */
typedef void (*polychord_prior_fn)(double *cube, double *theta, int nDims);
/* dumper(ndead, nlive, npars, live, dead, logweights, logZ, logZerr)
 *                                                          (interfaces.h:5, interfaces.F90:311-322) */
typedef void (*polychord_dumper_fn)(int ndead, int nlive, int npars, double *live, double *dead,
                                    double *logweights, double logZ, double logZerr);

/* replaces polychord_c_interface (interfaces.h:2-45, interfaces.F90:285-436).  Scalars by value,
 * `comm` by reference (an MPI_Fint in the reference's MPI build; ignored here, ranks are processes
 * of torch.distributed / one GPU each).  Strings are NUL-terminated. */
void polychord_c_interface(
    polychord_loglike_fn loglikelihood, polychord_prior_fn prior, polychord_dumper_fn dumper,
    int nlive, int num_repeats, int nprior, int nfail, bool do_clustering, int feedback,
    double precision_criterion, double logzero, int max_ndead, double boost_posterior,
    bool posteriors, bool equals, bool cluster_posteriors, bool write_resume, bool write_paramnames,
    bool read_resume, bool write_stats, bool write_live, bool write_dead, bool write_prior,
    bool maximise, double compression_factor, bool synchronous, int nDims, int nDerived,
    char *base_dir, char *file_root, int nGrade, double *grade_frac, int *grade_dims, int n_nlives,
    double *loglikes, int *nlives, int seed, int *comm);

/* replaces polychord_c_interface_ini (interfaces.h:47-56, interfaces.F90:496-519) */
void polychord_c_interface_ini(polychord_loglike_fn loglikelihood, void (*setup_loglikelihood)(void),
                               char *inifile, int *comm);

/* ===== Part 2: engine surface =============================================================== */

enum { PCHIP_LIKE_CALLBACK = 0, PCHIP_LIKE_GAUSSIAN = 1, PCHIP_LIKE_RASTRIGIN = 2,
       PCHIP_LIKE_TWIN_GAUSSIAN = 3, PCHIP_LIKE_CORR_GAUSSIAN = 4 };

/* Built-in likelihoods that run fused on the device.  These are ordinary host functions with the
 * reference's callback signature (they evaluate the same formula on the host); when one of them
 * is passed as `loglikelihood` to polychord_c_interface the engine recognises the pointer and
 * evaluates the likelihood inside the slice-sampling kernel instead of calling back.
 * (likelihoods/examples/gaussian.f90:12-41, rastrigin.f90:20-35, twin_gaussian.f90:14-56,
 *  random_gaussian.f90:17-30) */
double polychord_hip_gaussian(double *theta, int nDims, double *phi, int nDerived);
double polychord_hip_rastrigin(double *theta, int nDims, double *phi, int nDerived);
double polychord_hip_twin_gaussian(double *theta, int nDims, double *phi, int nDerived);
double polychord_hip_corr_gaussian(double *theta, int nDims, double *phi, int nDerived);
/* parameters of the built-ins (process-global like the reference's setup_loglikelihood state) */
void polychord_hip_set_gaussian(double mu, double sigma);
void polychord_hip_set_corr_gaussian(int nDims, const double *invcov_rowmajor, const double *mean, double logdetcov);
/* uniform box prior evaluated on the device; pass polychord_hip_uniform_prior as `prior` */
void polychord_hip_uniform_prior(double *cube, double *theta, int nDims);
void polychord_hip_set_uniform_prior(int nDims, const double *lo, const double *hi);
/* Batched host evaluation.  In host-callback mode the engine parks the proposals of all chains of a nursery and hands them
 * to the host together; with a batch callback registered it makes ONE call per round instead of one loglikelihood call
 * per proposal: prior + likelihood for n hypercube points (row-major, host memory; logL[i] <= logzero marks an invalid
 * point).  This is the hook for vectorised likelihoods, for likelihoods that run on an accelerator themselves, and for
 * farming the evaluations out to MPI ranks (what the reference's MPI workers did, nested_sampling.F90:426-498).  The scalar
 * callbacks passed to polychord_c_interface must still be valid (they serve the maximiser and single evaluations).
 * Process-global like the other setters; NULL removes it. */
typedef void (*polychord_batch_fn)(void *user, int n, int nDims, int nDerived, const double *cube, double *theta, double *phi,
                                   double *logL);
void polychord_hip_set_batch_callback(polychord_batch_fn fn, void *user);
/* asks a running engine to stop at the next host callback boundary (used by language bindings when a
 * user callback raised: the reference throws through the Fortran frames, _pypolychord.cpp:219-224) */
void polychord_hip_request_stop(void);
/* .resume files (reference grammar) without a run: parse `in`, optionally write it back to `out`;
 * counts[0..5] = nDims, nDerived, ndead, ncluster, ncluster_dead, total live points.  0 on success. */
int polychord_hip_resume_copy(const char *in, const char *out, int *counts);
/* the prior block of an ini file (ini.f90:354-458, priors.f90: uniform, log_uniform, power_uniform, gaussian,
 * half_gaussian, exponential and their sorted_ variants) evaluated at one hypercube point; returns nDims, -1 if n < nDims */
int polychord_hip_ini_prior(const char *inifile, const double *cube, double *theta, int n);
/* inverse normal CDF, AS241 PPND16 (utils.F90:806-966) */
double polychord_hip_inv_normal_cdf(double p);
/* engine options by name: "batch" (chains per nursery), "device" (HIP device ordinal), "halt_returns" (1: a fatal
 * condition inside polychord_c_interface -- which the reference answers with a message and `stop 1`, abort.F90:19-29,
 * and so does this library by default -- returns to the caller instead, with the message kept for
 * polychord_hip_last_error(): what a language binding needs to raise an exception rather than lose its interpreter),
 * "inject_fault" (tests: the next run fails once -- 1: a device allocation, 2: cluster capacity at the next split,
 * 3: growth of the phantom array) */
void polychord_hip_set_option(const char *name, double value);
/* message of the fatal condition that ended the last polychord_c_interface call in "halt_returns" mode, else NULL */
const char *polychord_hip_last_error(void);
void pchip_inject_fault(int kind);
/* initial capacities of the per-cluster arrays (default 128) and of the phantom array (rows; 0 = estimate); both grow on
   demand like the reference's reallocating arrays (run_time_info.f90:392-418), options "cluster_capacity" / "phantom_capacity" */
void pchip_set_capacity(int clusters, int phantom_rows);
/* The engine keeps the device and pinned blocks of finished runs for the next ones (up to 64 GB / 12 GB); this gives them back
 * to the driver -- e.g. after many runs in flight, before a run that sizes its buffers by the memory that is free.
 * polychord_hip_set_option("trim_cache", 0) does the same. */
void pchip_trim_cache(void);

typedef struct {
    int nDims, nDerived;
    int nlive, num_repeats, nprior, nfail;
    int do_clustering;
    double precision_criterion, logzero;
    int max_ndead;
    double boost_posterior;
    int posteriors, equals, cluster_posteriors;
    double compression_factor;
    int n_nlives; const double *loglikes; const int *nlives;
    int seed;
    int batch;          /* chains per synchronous nursery (the reference's nprocs-1); 0 = auto */
    int device;         /* HIP device ordinal, -1 = current/0 */
    int feedback;
    int profile;        /* 0: off; 1: HIP-event stopwatch around every kernel class; otherwise a bit mask,
                           bit k+1 = time kernel class k (0 nhats, 1 slice, 2 consume, 3 apply, 4 clean, 5 covmats, 6 side-stream bases):
                           a few hundred event records per run instead of a few thousand; bits 8..15 = n: only every
                           n-th launch of a class is timed (k_launches then counts the timed ones) */
    int force_general;  /* 1: always use the general contraction kernel (tests) */
    int ablate;         /* developer / test switches, 0 = product: bit 0 = built-in quadratic likelihoods evaluated like a general functor;
                           bits 1, 2, 3 = no pool mode, no deferred update, no fused update (identical results by the older kernels);
                           bit 4 = evidence prefixes of the parallel contraction by pair scans only; bit 5 = several clusters: every
                           launch by the general contraction kernel (no one-wave kernel); bits 6, 7 = a run on its own by the kernels it uses
                           next to other runs of its device (bit 6: lane-per-chain sampling, pc_slice_t.hip; bit 7: lane-per-vector bases); bit 8 = the one-wave
                           contraction of clustered runs ends its launch at a cluster's death (the host relaunches for the rest of the nursery) instead
                           of sorting the live set itself and going on; bit 9 = the kill-off of a run that ends with several clusters by the general
                           contraction kernel instead of the one-wave kernel k_killoff_cl (the same bits); bit 10 = several clusters: the contraction whose ONE
                           wavefront decides chain after chain (k_consume_cl) instead of the one with parallel decisions (k_consume_clp): the same run; bit 11 = k_consume_clp with
                           update_evidence and the live evidence as sums over all of a pass's deaths (a walk per cluster, pair sums, prefix sums: phase C' of
                           pc_consume_clp_body.inc) instead of death after death on two wavefronts -- the same run to rounding in <Z^2>; an experiment that is
                           NOT the default: its walks are as long as the largest cluster's events, and at the BASELINE shapes that is no shorter; bit 12 = runs in step
                           (and bit 7): the deviates of a basis made in the registers of the Gram-Schmidt kernel (k_bases_own) instead of
                           passing through HBM from a kernel of their own -- the same bases bit for bit, a third of the round's bytes less, NOT faster; bit 13 = the
                           fused sampling kernel with ONE wavefront a workgroup (a chain) instead of four chains and their four helper wavefronts
                           (deck shuffle and whitening next to the seed choice instead of in front of it): the same numbers; bit 14 = with the helper, the closed
                           form's s.M.s of a direction reduced by the chain at the head of its slice instead of taken from the table the helper made with the
                           whitening (the same bits: the table's sums follow the wave butterfly's order) */
    const char *resume_write;  /* path of a .resume file (reference grammar, read_write.F90:219-288) rewritten at every
                                  update and at the end; NULL = off */
    int sequential_rng; /* tests: ONE Philox stream consumed in the reference's program order (forces batch = 1 and the
                           general contraction kernel with the reference's list rule): with the generator shim of
                           oracle/ref_rng_shim.c the reference binary then produces the same run, draw for draw */
    const char *resume_read;   /* start from this .resume file if it exists (read_write.F90:384-476; also the file
                                  pypolychord writes for `cube_samples`); NULL = off */
    /* fast/slow parameter grades (chordal_sampling.f90:94-145, generate.F90:303-309): nGrade <= 1 = one grade of nDims
       parameters with num_repeats repeats.  Otherwise grade g owns grade_dims[g] parameters (sum = nDims) and
       grade_repeats[g] directions per chain, drawn in the subspace of its own and all faster parameters;
       num_repeats is then ignored (the total is the sum).  At most 8 grades. */
    int nGrade; const int *grade_dims; const int *grade_repeats;
    int epoch_discard;  /* chains in flight when the list of clusters changes (batch > 1 only).  The reference's farm discards every baby
                           seeded before the change (nested_sampling.F90:313, :339-341: its workers' messages carry positional cluster
                           indices).  0 (default): only the chains seeded in the cluster that ended -- it died or was split -- are lost;
                           the others are samples as good after the change as before and stay in the nursery, their cluster index following
                           the list (BASELINE configs[2] at batch = nlive/2: 199 instead of 338 evaluations per dead point, the reference's
                           linear mode 155).  1: the reference's rule.  Option "epoch_discard" for polychord_c_interface callers. */
    int device_records; /* 1: the run also leaves the records of its points that entered the live set ON THE DEVICE (pchip_result.d_records:
                           what the exchange step of repeat-sharded runs sends; pchip_run_repeats sets it itself).  The dead points of a run
                           are made on the device; without this the merge uploads them again from the host arrays of the result */
} pchip_settings;

typedef struct {
    int kind;                      /* PCHIP_LIKE_* */
    double mu, sigma;
    const double *invcov;          /* host, row-major D x D (corr gaussian) */
    const double *mean;            /* host, D */
    double logdetcov;
    polychord_loglike_fn fn;       /* callback kind */
} pchip_like;

typedef struct {
    int kind;                      /* 0 callback, 1 uniform box */
    const double *lo, *hi;         /* host, D each; NULL => [0,1] */
    polychord_prior_fn fn;
} pchip_prior;

typedef struct {
    double logZ, varlogZ;
    long ndead, nlike, niter, nbatches, nrounds, nupdates;
    int ncluster, ncluster_dead, nTotal, batch;
    double t_generate, t_loop, t_final, t_total;   /* host wall-clock of the phases, seconds */
    double t_setup, t_results, t_teardown;         /* allocation / result download / free */
    double k_time_s[8]; long k_launches[8];        /* HIP-event time per kernel class: nhats, slice, consume, apply,
                                                      clean, covmats, bases drawn ahead on the side stream (profile=1) */
    double *dead, *logweights;     /* [ndead][nTotal] rows [cube|theta|phi|birth|logL], [ndead] */
    double *entry;                 /* [ndead] global contour when the point entered the live set
                                      (== birth column when batch = 1); used by the multi-run merge */
    double *live; int nlive_final; /* live set at termination, before the final kill-off */
    double *logZp, *varlogZp; int nZp;
    double *post_mean, *post_var;  /* [nDims + nDerived] weighted posterior moments of theta, phi */
    long nlike_grade[8];           /* likelihood evaluations per grade (RTI%nlike; the prior samples count for grade 1) */
    int *live_cluster;             /* [nlive_final] 0-based cluster of each row of `live` */
    long nlike_failed;             /* of nlike: evaluations spent on chains whose spawn failed (their last point fell below the
                                      contour that had risen since the nursery was seeded; batch = 1 has almost none) */
    int ncluster_peak;             /* largest number of clusters alive at the same time */
    int epoch_discard;             /* the rule this run followed for chains in flight when the cluster list changed (pchip_settings.epoch_discard:
                                      0 = the engine's, 1 = the reference farm's); it only matters with batch > 1 and several clusters */
    /* settings.device_records: the lived records (logweight > logzero) in death order, in DEVICE memory of device `records_device`, owned
       by the result (pchip_result_free): rows [n_records][nTotal] at d_records, their entry contours at d_records + records_cap * nTotal,
       their own log weights at d_records + records_cap * (nTotal + 1).  NULL when not asked for. */
    double *d_records; long n_records, records_cap; int records_device;
    /* Which kernels the run went through: launches per variant, counted where the host chooses (PCHIP_PATH_* below).  A shape that
       silently leaves a fast path -- an LDS layout that no longer fits, a guard that no longer holds -- shows up here and nowhere
       else: its numbers are the same.  tests/test_baseline_configs.py holds the BASELINE shapes to their paths. */
    long path[24];
} pchip_result;
enum { PCHIP_PATH_CONSUME_PAR = 0,      /* one cluster: the parallel contraction k_consume_par (pc_par.hip) */
       PCHIP_PATH_CONSUME_CL = 1,       /* several clusters: k_consume_clp, decisions in parallel (pc_clus.hip, pc_consume_clp_body.inc) */
       PCHIP_PATH_CONSUME_GENERAL = 2,  /* the general serial kernel k_consume (pc_contract.hip): dynamic nlive, sequential test mode, shapes beyond the LDS */
       PCHIP_PATH_CONSUME_FAST = 3,     /* one cluster, one wavefront: k_consume_fast (B > 1024) */
       PCHIP_PATH_KILLOFF_PAR = 4, PCHIP_PATH_KILLOFF_CL = 5, PCHIP_PATH_KILLOFF_GENERAL = 6, PCHIP_PATH_KILLOFF_FAST = 7,   /* the final kill-off's kernel */
       PCHIP_PATH_UPDATE_FUSED = 8,     /* updates by the fused kernels of pc_update.hip */
       PCHIP_PATH_UPDATE_STEPS = 9,     /* updates by clean + covariance + Cholesky launches (clustered runs, posteriors with boost, nDims beyond the fused kernel) */
       PCHIP_PATH_SLICE_WAVE = 10,      /* nurseries sampled by k_slice (wavefront = chain) */
       PCHIP_PATH_SLICE_LANE = 11,      /* nurseries sampled by k_slice_t (lane = chain: runs in step) */
       PCHIP_PATH_NN_LISTS = 12,        /* candidate-list launches (k_nn_lists) */
       PCHIP_PATH_NN_FALLBACKS = 13,    /* chains whose candidate lists held no living entry: the full search inside the contraction */
       PCHIP_PATH_POOL_MODE = 14,       /* 1: babies written straight into the phantom array */
       PCHIP_PATH_DEFER_UPDATE = 15,    /* 1: the contraction runs past update triggers */
       PCHIP_PATH_CONSUME_CL_SERIAL = 16,   /* several clusters: k_consume_cl, one wavefront deciding chain after chain (settings.ablate bit 10, or an LDS
                                               block the parallel kernel's tables push over the limit) */
       PCHIP_PATH_COUNT = 24 };

/* snapshot handed to the update hook: what the reference's file writers see at every update
   (nested_sampling.F90:323-340, read_write.F90) -- host memory owned by the engine, valid during the call */
typedef struct {
    int final_call;                 /* 0: an update; 1: the call after the kill-off (nested_sampling.F90:386-398);
                                       2: the initial live points, before any death (write_prior_file, :197) */
    long ndiscarded;                /* prior samples rejected while the live points were generated */
    long ndead; int nlive, npars;   /* npars = nDims + nDerived + 2 */
    const double *dead;             /* [ndead][npars]  theta, phi, birth contour, logL -- in death order */
    const double *logpost;          /* [ndead] logweight + logL; failed spawns carry logzero + logL */
    const double *live;             /* [nlive][npars], ordered by cluster, then position in the cluster's list */
    const int *live_cluster;        /* [nlive] 0-based cluster of each live row */
    double logZ, logZerr;
    long nlike;
    int ngrade; const long *nlike_grade; const int *grade_dims, *grade_repeats;   /* [ngrade] RTI%nlike / num_repeats per grade */
    int ncluster, ncluster_dead;
    const int *nlive_p;             /* [ncluster] */
    const double *logZp, *logZperr;            /* [ncluster] clusters still active */
    const double *logZp_dead, *logZperr_dead;  /* [ncluster_dead] */
    /* cluster bookkeeping for the per-cluster posterior files (run_time_info.f90:303-505: a split hands every
       child a copy of the parent's posterior points, weights scaled by the evidence fraction it received) */
    const unsigned *dead_cluster;              /* [ndead] id of the cluster each point died in (failed spawns: 0xFFFFFFFF) */
    const unsigned *cluster_uid, *cluster_uid_dead;   /* ids of the active / dead clusters, same order as logZp* */
    int nsplit;                                /* children created by splits so far */
    const unsigned *split_child, *split_parent;
    const double *split_logfrac;               /* log(evidence of child / evidence of parent) at the split */
    /* boost_posterior: phantom points kept as posterior samples (run_time_info.f90:845-870), same row layout */
    int n_extra; const double *extra, *extra_logpost; const unsigned *extra_cluster;
    const unsigned long long *extra_uid;       /* [n_extra] the phantoms' ids (nursery << 32 | chain * num_repeats + baby): what keys their trials */
} pchip_update;
typedef void (*pchip_update_fn)(void *user, const pchip_update *u);

/* optional host hooks of a run */
typedef struct {
    polychord_dumper_fn dumper;   /* called at every update and at the end, like nested_sampling.F90:335,392 */
    pchip_update_fn on_update;    /* same moments; may be NULL */
    void *user;
} pchip_hooks;

/* The structs of this header grow at their END from release to release and carry no size member (their first members are what existing
 * bindings rely on).  A binding that mirrors them (ctypes, ISO_C_BINDING, cgo ...) checks itself against the library it loaded:
 * pchip_abi_version() == PCHIP_ABI_VERSION of the header it was written against, and pchip_sizeof("settings" | "result" | "merged" |
 * "like" | "prior" | "update") == the size of its own mirror (0 for an unknown name). */
#define PCHIP_ABI_VERSION 7
int  pchip_abi_version(void);
unsigned long pchip_sizeof(const char *struct_name);
void pchip_settings_default(pchip_settings *s, int nDims, int nDerived);
int  pchip_device_count(void);
/* full run; returns 0 on success, else (message on stderr, nothing to free, the process goes on): 1 settings, 2 HIP /
   device error, 3 nDims unsupported, 4 a working set beyond the LDS, 5 stopped on request, 6 resume file, 7 out of
   memory, 8 a capacity limit */
int  pchip_run(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, pchip_result *out);
int  pchip_run_hooks(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior,
                     const pchip_hooks *hooks, pchip_result *out);
void pchip_result_free(pchip_result *r);
/* `maximise = T` (maximiser.F90:32-224, nelder_mead.f90, write_max_file read_write.F90:754-807): Nelder-Mead polish of
   the best live points of a finished run into the maximum-likelihood and maximum-posterior points, written to `path`
   (<base_dir>/<file_root>.maximum).  Host code; live rows as in pchip_result.  0 on success. */
int  pchip_maximise(polychord_loglike_fn loglikelihood, polychord_prior_fn prior, int nDims, int nDerived, double logzero,
                    const double *live, const int *live_cluster, int nlive, const double *post_mean, const char *path);
/* ---- repeat-sharded runs: independent runs of one problem, merged ------------------------------------
 * The reference's way to use more hardware is its MPI farm (nested_sampling.F90:262-301, mpi_utils.F90:376-463: workers'
 * babies gathered into one run).  Here every GPU carries a complete run of its own; the dead points of all runs are
 * gathered and their union -- a nested-sampling run with n(L) = sum of the runs' live points -- gives one evidence
 * (error ~ 1/sqrt(runs)) and one posterior. */
typedef struct {
    double logZ, varlogZ;          /* evidence of the union (log-normal location and variance, run_time_info.f90:652-678) */
    long n; int nTotal, nruns;     /* points that entered a live set, over all runs */
    double *rows;                  /* [n][nTotal] merged dead points ascending in logL (host; only if asked for); the birth
                                      column holds the contour at which the point ENTERED its live set */
    double *logweights;            /* [n] log prior-volume weight logX_i - log(n_i + 1) of each merged point */
    int *nlive;                    /* [n] live points over all runs just before each death */
    double *post_mean, *post_var;  /* [nDims + nDerived] posterior moments of theta, phi */
    double t_merge_s, t_runs_s;    /* wall clock of the merge / of the runs (pchip_run_repeats) */
    long nlike, ndead_all;         /* totals over the runs (pchip_run_repeats) */
    double runs_logZ_mean, runs_logZ_sem;   /* mean of the runs' OWN log Z and its standard error (one run: that run's own error) */
    /* Which evidence `logZ, varlogZ` (and `logweights`, `post_mean`, `post_var`) are: evidence_rule
     *   0 -- no run ever held more than ONE cluster: the replay of the union from ranks and live counts (the sharper estimator, and what
     *        anesthetic computes from <root>_dead-birth.txt); logZ_replay == logZ;
     *   1 -- at least one run held several clusters at some time (pchip_result.ncluster_peak > 1; `nclustered` of them).  Such a run weighs a dead point by
     *        its CLUSTER's volume over the cluster's live count (run_time_info.f90:211-296, volumes apportioned at a split :458-503), which
     *        the replay does not know (10-D Rastrigin, dozens of clusters: the replay sits 0.46 below the runs' own log Z, twenty of its own
     *        error bars).  Then logZ, varlogZ = the runs' own evidences combined in linear space -- log-normal moments of each run,
     *        mean of the Z_r, variance the larger of the propagated one and the scatter between the runs -- and record i of a run keeps its
     *        OWN log weight minus log(nruns): sum_i w_i L_i over the union = the mean of the runs' Z_r, so the union's posterior is the
     *        Z_r-WEIGHTED mixture of the runs' posteriors (run r carries Z_r / sum Z), not an equal-weight one.  The replay stays in
     *        logZ_replay, varlogZ_replay; `nlive` is the replay's live count either way -- and <root>_dead-birth.txt of such a union, read
     *        by anesthetic, reproduces logZ_replay, not the log Z written in <root>.stats. */
    double logZ_replay, varlogZ_replay;
    int evidence_rule, nclustered;
} pchip_merged;
/* Merge `nruns` runs on the device.  rows = the lived records (logweight > logzero) of run 0, then run 1, ...: counts[q]
 * rows of nTotal doubles each, every run ascending in logL (the order in which they died); entry[i] = contour at which
 * record i entered its live set (pchip_result.entry).  on_device: the two arrays are device pointers (e.g. the buffer an
 * RCCL all-gather filled), else host memory that is uploaded first.  Returns 0, or a pchip_run code; no CPU path. */
int  pchip_merge_records(int nDims, int nDerived, int nruns, const long *counts, const double *rows, const double *entry,
                         int on_device, int want_rows, pchip_merged *out);
/* The same with what each run knows about itself: ownw[i] = the log prior-volume weight record i had in its own run
 * (pchip_result.logweights of the lived records, laid out like `entry`), run_logZ / run_varlogZ / run_clustered[q] = the run's own
 * evidence and whether it ever held more than one cluster (pchip_result.ncluster_peak > 1).  If any run did, the union's evidence and
 * weights follow pchip_merged.evidence_rule 1; all four NULL = pchip_merge_records. */
int  pchip_merge_records_ex(int nDims, int nDerived, int nruns, const long *counts, const double *rows, const double *entry,
                            const double *ownw, const double *run_logZ, const double *run_varlogZ, const int *run_clustered,
                            int on_device, int want_rows, pchip_merged *out);
void pchip_merged_free(pchip_merged *m);
/* <root>.stats (global evidence, counters, posterior means), <root>_dead-birth.txt and <root>.txt of a merged result in
 * the reference's formats (read_write.F90:809-910, :707-716, :479-617), so that PolyChordOutput / anesthetic read the
 * union like a single run.  0 on success. */
int  pchip_merged_write(const pchip_merged *m, int nDims, int nDerived, const char *base_dir, const char *file_root);
/* nseeds independent runs (settings `s` with seed = seeds[k]) spread over `devices` (ordinals of this process; NULL / 0 =
 * the device of `s`), at most max_in_flight at a time per device, one host thread each; results[k] as from pchip_run
 * (free each with pchip_result_free), merged (may be NULL) = their union with rows.  0 on success. */
int  pchip_run_repeats(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds,
                       int ndevices, const int *devices, int max_in_flight, pchip_result *results, pchip_merged *merged);
/* the same; want_rows = 0: the union without its merged rows (evidence, weights, live counts, moments only: 370 MB less to bring back
   for sixteen runs of the metric configuration) */
int  pchip_run_repeats_ex(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds,
                          int ndevices, const int *devices, int max_in_flight, int want_rows, pchip_result *results, pchip_merged *merged);

/* ---- between processes: one rank per GPU, RCCL inside the library (dlopen of librccl.so at first use, none at link time).
 * Replaces the reference's MPI exchange (mpi_utils.F90:376-463 throw_baby / catch_babies, nested_sampling.F90:262-301) for
 * the repeat-sharded mode: the only message of a job is ONE all-gather of the runs' lived records at its end.
 *   pchip_comm_get_id: 128 opaque bytes (an ncclUniqueId) made by ONE rank and handed to all others by the caller -- a
 *       file, MPI_Bcast, a torch.distributed store: the library does not care;
 *   pchip_comm_create: collective over the nranks processes (ncclCommInitRank), `device` = this rank's HIP ordinal;
 *   pchip_comm_merge: picks the points of `run` that entered a live set (logweight > logzero) on the device, all-gathers
 *       the counts and then one padded block [nmax][nTotal] | [nmax] | [nmax] per rank (ncclAllGather over xGMI), and merges the
 *       union on every rank with pchip_merge_records; nlike / ndead_all of the result are the totals over the ranks.
 *       c = NULL: this process alone (no collective library is touched).
 * All return 0 or a pchip_run code; nothing has a CPU path. */
typedef struct pchip_comm pchip_comm;
#define PCHIP_COMM_ID_BYTES 128
int  pchip_comm_get_id(char *id128);
int  pchip_comm_create(const char *id128, int nranks, int rank, int device, pchip_comm **out);
void pchip_comm_destroy(pchip_comm *c);
int  pchip_comm_merge(pchip_comm *c, const pchip_result *run, double logzero, int nDims, int nDerived, int want_rows,
                      pchip_merged *out);
/* the same for a rank that holds `nruns` runs (pchip_run_repeats with merged = NULL: the runs of a GPU in step): the ranks' run
 * counts travel first, then six words per run (count, nlike, ndead, log Z, var log Z, clustered?), then ONE padded block per rank
 * [nmax][nTotal] | entry [nmax] | own log weight [nmax] with the rank's runs one after the other; the union of all ranks' runs is
 * merged on every rank (pchip_merge_records_ex).  Ranks may hold different numbers of runs. */
int  pchip_comm_merge_many(pchip_comm *c, const pchip_result *runs, int nruns, double logzero, int nDims, int nDerived, int want_rows,
                           pchip_merged *out);
/* The same exchange over the CALLER's collective instead of RCCL: all_gather(user, send, recv, bytes) places every rank's `bytes` at `send`
 * (device memory of `device`) into `recv` (device memory, nranks * bytes, rank after rank) on all ranks; it is called with the library's stream
 * drained and returns 0 once `recv` is complete.  For a host that brings its own transport (a GPU-aware MPI_Allgather under the reference's
 * MPI launcher, mpi_utils.F90:376-463; torch.distributed) and for tests with several ranks on one GPU, where RCCL forms no communicator.
 * pchip_comm_merge / _merge_many / _destroy take the communicator like one made by pchip_comm_create. */
typedef int (*pchip_allgather_fn)(void *user, const void *send, void *recv, unsigned long bytes);
int  pchip_comm_create_with(pchip_allgather_fn all_gather, void *user, int nranks, int rank, int device, pchip_comm **out);
const char *pchip_comm_library(void);   /* the RCCL that was resolved (path or soname), NULL if none could be loaded */

/* kernel-level: directions + slice chains only (parity tests against oracle pc_slice_chain) */
int  pchip_slice_chains(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior,
                        unsigned batch, int nchains, const double *seeds, const double *chol,
                        double contour, double *babies_out, double *nhats_out, int *nlike_out);

#ifdef __cplusplus
}
#endif
#endif
