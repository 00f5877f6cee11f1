// pc_state.h -- device-resident sampler state of the MI355X nested-sampling engine.
//
// Replaces the reference's `run_time_info` type (src/polychord/run_time_info.f90:10-107):
// instead of per-cluster 3-D Fortran arrays that are re-packed on every split/death, the
// live set is a flat array of slots with (cluster, position-in-cluster-list) labels, so a
// cluster split or death is a relabel, and phantoms / dead points are append-only arrays.
// Everything below lives in HBM; the host only sees the small PcCtl block.
#pragma once
#include "pc_dev.h"

enum { PC_ST_RUNNING = 0, PC_ST_DONE = 1, PC_ST_UPDATE = 2, PC_ST_ERROR = 4 };
enum { PC_ERR_NONE = 0, PC_ERR_PHANTOM_CAP = 1, PC_ERR_DEAD_CAP = 2, PC_ERR_CLUSTER_CAP = 3, PC_ERR_NOSLOT = 4 };

#define PC_MASK_WORDS 16         /* phantom mask words per chain: num_repeats <= 1024 */

struct PcCtl {                   // written by the consume kernel, read by the host after each round
    int status;                  // PC_ST_*
    int error;
    int i_nursery;               // nursery entries still to be consumed (nested_sampling.F90:262-286)
    int admin_epoch;             // administrator epoch (nested_sampling.F90:313)
    int failures;                // consecutive failed spawns (nested_sampling.F90:315-319)
    int ncluster;
    int ncluster_dead;
    int ndead;
    int nphantom;                // rows used in the phantom array (incl. not yet cleaned ones)
    int seg_hi, seg_lo;          // chains [seg_lo, seg_hi] (descending) consumed by the last segment
    int cluster_deleted;         // a cluster died in the last segment (host bookkeeping hint)
    unsigned batch_id;           // id of the batch in the nursery
    unsigned next_cluster_uid;
    long long nlike;             // RTI%nlike(1)
    long long niter;             // consumed nursery entries
    long long nlike_device;      // evaluations executed on the device (incl. dropped nursery)
    long long nlike_failed;      // of nlike: evaluations of chains whose spawn failed (the price of seeding a whole nursery from one snapshot)
    double logZ, logZ2;          // run_time_info.f90:165-166 (log <Z>, log <Z^2>)
    double logX_last_update;
    double live_logZ;            // last evaluated termination estimate
    unsigned long long seq;      // seq_mode: uniforms consumed so far
    long long dbg[8];            // developer cycle counters of the contraction kernel
    long long gen_cyc[4];        // general contraction kernel: cycles in termination test / identify / kill+add / tail
    long long nn_walks, nn_fallbacks;   // chains identified from the candidate lists / of those, chains that needed the full search
    // deferred update (parallel contraction, S.defer_update): the launch ran past its update trigger(s); the update is
    // made afterwards for the state at the LAST mark of the launch (only that covariance is ever sampled from)
    int upd_pending, upd_marks;         // an update is due / triggers passed in this launch
    int upd_tmark;                      // steps of the launch before the mark (the step that caused the triggering death included)
    int upd_T, upd_ts;                  // steps in the nursery at launch / consumed by the launch
    int upd_nph0;                       // phantom rows in use when the launch started (its regions begin there)
    int upd_keep_thr, upd_pad;          // deaths after the mark: death_thr stays the last death's logL
    double upd_thr;                     // logL of the death that triggered the mark (clean_phantoms' threshold)
    int chol_suspect, chol_pad;         // the blocked factorisation met a pivot it does not trust: the reference-order kernel behind it decides
    int spec_ok, upd_in;                // upd_in: lived deaths until the next update trigger, as the state stands after this launch (0 = not known); spec_ok: parallel contraction: number of the nursery that may be sampled at once (this one was consumed whole, no
                                        // update is due, the run goes on), or -1: what a speculatively enqueued k_slice asks first
    long long wave_cyc[4];              // developer counters of k_consume_clp: cycles waves 1, 2, 3 spend in their loops over a pass's deaths, the phantom waves in theirs
};

#define PC_MAX_GRADE 8
#define PC_NN_K 8
#define PC_NN_NONE (-2147483647 - 1)
#define PC_CUID_NONE 0xFFFFFFFEu   /* ph_cuid of a row of the phantom array that holds no phantom (no cluster has this id; the first cluster's id is 0) */

// one record per nursery chain, written by the consume kernel.  The scalar fields are the first 64 bytes, in 16-byte groups:
// the parallel contraction writes them as four 16-byte stores per chain (PcPlanHead), not eleven scattered ones
struct alignas(16) PcPlanHead {
    int dead_idx;                // index in dead[] or -1
    int dead_src;                // >=0: live slot; <0: -(1+chain) whose last baby is the row
    int ph_base;                 // first phantom row of the chain
    unsigned dead_cuid;
    unsigned ph_cuid;
    int ph_count;                // -1: the apply side derives mask/count/base from `contour` (one cluster)
    double logw;
    double postX, postZ;
    double postXs;               // the volume column is postX + log(postXs): the serial kernel leaves the log to the apply side
    double contour;              // global contour when the chain was consumed (phantom test, entry contour)
};
struct alignas(16) PcPlan : PcPlanHead {
    unsigned long long ph_mask[PC_MASK_WORDS];
};
static_assert(sizeof(PcPlanHead) == 64 && sizeof(PcPlan) == 64 + 8 * PC_MASK_WORDS, "plan record layout");

struct PcState {
    // ---- geometry / settings
    int D, nDer, nT, nr, N, Ncap, B, maxc, Pcap, Dcap;
    int b0, l0, p0, d0;          // 0-based offsets in a point row [cube|theta|phi|birth|logL]
    uint32_t k0, k1;             // Philox key
    double logzero, log_prec, log_cf;
    int use_prec, max_ndead, nfail;
    int n_nlives; const double *dyn_loglikes; const int *dyn_nlives;
    PcLike like; PcPrior prior;
    // ---- live set: slots
    double *live;                // [Ncap][nT]
    double *live_logL;           // [Ncap]
    int *live_cluster;           // [Ncap] cluster index, -1 = free slot
    int *live_pos;               // [Ncap] position in the cluster's list (reference ordering)
    double *live_entry;          // [Ncap] global contour at the moment the point entered the live set
    int *cl_list;                // [maxc][Ncap] slot ids in list order (rebuilt after every segment)
    int *cl_n;                   // [maxc]
    // ---- per-cluster evidence state (run_time_info.f90:60-100)
    double *logZp, *logXp, *logZXp, *logZp2, *logZpXp, *logLp, *XpXq /* [maxc*maxc] */;
    int *imin_slot;              // [maxc] slot of the lowest live point
    double *lse_ref, *lse_sum;   // incremental logsumexp of the live logL of each cluster
    double *death_thr;           // [maxc] logL of the last death in the cluster since the last clean
    unsigned *cl_uid;            // [maxc] stable ids (phantoms carry these)
    double *chol, *cov;          // [maxc][D*D] row-major lower Cholesky / covariance
    double *logZp_dead, *logZp2_dead;   // [maxc_dead]
    unsigned *cl_uid_dead;              // [maxc_dead] uid of every dead cluster (cluster posterior files)
    int maxc_dead;
    // ---- phantoms (append-only between cleans)
    double *phantom;             // [Pcap][nT]
    double *ph_logL;             // [Pcap]
    unsigned *ph_cuid;           // [Pcap] cluster uid
    unsigned long long *ph_uid;  // [Pcap] (batch<<32 | chain*nr+baby)
    // ---- dead points
    double *dead;                // [Dcap][nT]
    double *dead_logw;           // [Dcap]
    double *dead_postX, *dead_postZ;   // posterior-stack columns (calculate.f90:53-79)
    unsigned *dead_cuid;
    double *dead_entry;          // [Dcap] entry contour of the dead point (== birth when B = 1)
    // ---- nursery (one synchronous batch of B chains)
    double *babies;              // [B][nr][nT]
    double *baby_logL;           // [B][nr]
    double *baby_logL_T;         // [nr][B] the same values, slice-major (read by the parallel contraction, lane = chain)
    int *ch_cluster, *ch_epoch, *ch_nlike, *ch_seed_slot;
    double *ch_contour;          // [B] contour each chain sampled under
    double *nhat;                // [B][nr][D] whitened, normalised directions (generation order)
    double *nhat_w;              // [B][nr] 3*|L n|
    double *nhat_raw;            // [B][nb_total][D][D] orthonormal bases of the next nursery, before whitening (or null)
    // correlated Gaussian, 64 < nDims <= 128 (or null): the products the chord needs, formed on the matrix cores by the
    // kernel that makes the directions instead of one matrix-vector product per slice on the chain's critical path
    double *nhat_Ms;             // [B][nr][D] M.(span o n^) for every direction
    double *ch_My;               // [B][D] M.(theta_seed - mean) of every chain's start point
    // ---- plan written by the consume kernel for the apply kernels
    PcPlan *plan;                // [B]
    int *sort_slot;              // [NS] live slots ordered by (logL, list position), written by k_sort_live
    unsigned long long *sort_key; // [NS] sortable logL keys (pc_keys.h) in the same order
    int *slot_src;               // [Ncap] -1: live[] row is current; >=0: chain whose last baby now owns the slot
    int *slot_dead;              // [Ncap] pool mode: chain whose step killed the slot's occupant in the last launch (its row is still in live[]), -1: none
    int *slot_step;              // [Ncap] step of the last parallel-contraction launch at which the slot's occupant was accepted, -1: older
    int defer_update;            // the parallel contraction may run past update triggers (no host work is tied to an update)
    // pool mode (same conditions): the babies of a nursery are written by k_slice straight into the phantom array -- chain c
    // owns rows pool_base + c nr .. -- so becoming a phantom is a cluster id in a side array, not a row copy; updates
    // invalidate in place, and the array is compacted only when it runs full
    int pool, pool_base, pool_rows;
    // ---- fast/slow parameter grades (chordal_sampling.f90:94-145): grade g moves the parameters from g_off[g] to the
    //      last one with g_nr[g] directions taken from g_nb[g] orthonormal bases of that subspace; nr = sum g_nr.
    //      One grade: g_off = 0, g_nr = nr.  g_col0 = first direction of the grade in generation order, g_e0 = first
    //      deviate of the grade in the chain's PC_DOM_NHAT stream, n_dev = deviates per chain.
    int ngrade, nb_total; unsigned n_dev;
    int g_off[PC_MAX_GRADE], g_nr[PC_MAX_GRADE], g_nb[PC_MAX_GRADE], g_col0[PC_MAX_GRADE], g_e0[PC_MAX_GRADE];
    int *ch_nlike_g;             // [B][PC_MAX_GRADE] evaluations per grade of each chain (only when ngrade > 1)
    // ---- nearest-neighbour candidate lists for identify_cluster (run_time_info.f90:913-949), built once per nursery by
    //      the whole chip (k_nn_lists) at a moment T0: for every baby of every unconsumed chain the PC_NN_K nearest points
    //      among the live set at T0 and the last babies of the chains consumed before its own, ascending.  Entry >= 0:
    //      live slot as occupied at T0; entry < 0: -(1 + chain) whose last baby may have entered since; PC_NN_NONE: end.
    const double *logn;          // [Ncap + 4] log(k), k = 0 .. Ncap + 3 (log 0 = -huge): the evidence update needs log n, log(n+1), log(n+2)
    int *nn_list;                // [B][nr][PC_NN_K]
    int *nn_slot_owner;          // [Ncap] -1: occupant of T0 still there; -2: emptied since; w >= 0: last baby of chain w
    int *nn_chain_slot;          // [B] slot the chain's last baby went to since T0, or -1
    double *nn_pts;              // [Ncap + B][D] the points a baby can be nearest to, gathered once per nursery (k_nn_gather): live slots as occupied at T0,
                                 //   then the last baby of every chain still in the nursery
    int *nn_code;                // [Ncap + B] their codes (slot, -(1 + chain), PC_NN_NONE for an empty slot)
    int nn_valid;                // set by the host for the launches after T0 of the same nursery
    int ablate;                  // developer / bench switches (bit mask), 0 in production: bit 0 = the built-in quadratic-form
                                 // likelihoods are evaluated like any device functor (one reduction per trial) instead of in closed form
                                 // along the chord; bits 1 / 2 / 3 = no pool mode / no deferred update / no fused update (the same numbers
                                 // by the older kernels: tests/test_gpu_parity.py); bit 4 = the parallel contraction's evidence prefixes by pair
                                 // scans only (no linear-space path); bit 5 = several clusters: the general contraction kernel for every launch (not
                                 // the one-wave kernel of pc_clus.hip); bit 30 = trace of Cholesky fallbacks
    int seq_mode;                // tests: ONE running Philox stream consumed in the reference's program order
                                 // (B = 1 only; PcCtl::seq is the position), cf. oracle `sequential` mode
    int epoch_discard;           // 1: nested_sampling.F90:313 as written (a change of the cluster list loses every chain in flight); 0: only the ended cluster's
    int seed_override;           // test hook: chain c starts from slot c instead of a random seed
    int spec_guard;              // k_slice: enqueued ahead of the host's decision -- return unless ctl->spec_ok names this nursery
    PcCtl *ctl;
    // host notification: the contraction kernels copy the control block into a pinned, device-visible host mirror when
    // they are done and stamp it with notify_seq, so the host learns the outcome of a round by watching memory instead
    // of a copy + stream synchronisation (and goes on enqueueing while the row-copy kernels of the round still run)
    PcCtl *ctl_host; unsigned notify_seq;
};

// One run's share of a launch made for several runs at once (pchip_run_repeats: the runs of a device go round by round together,
// and each kernel of a round is launched ONCE for all of them, blockIdx.y = run): its state, and the buffers the one-run
// kernel takes as arguments (update: 0 keep, 1 block counts, 2 total, 3-6 the alternate phantom arrays, 7 partial sums, 8 shift).
struct PcManyRec { PcState S; void *p[10]; int ia[6]; int pad[2]; };      // ia: 0 the nursery's number, 1 phantom rows in use (update), 2 its blocks

#ifdef __HIPCC__
// ---- kernels that take their state from a record in memory (the runs of a device in step, blockIdx.y = run) --------------------------
// A pointer that a kernel READS FROM MEMORY is a generic pointer to the compiler (only kernel arguments are known to be global), and
// every access through it becomes a flat_load / flat_store.  FLAT instructions count on lgkmcnt as well as on vmcnt: every wait for an
// LDS read or a scalar load then also waits for the global loads and stores in flight -- the prefetches a serial loop lives on
// (k_consume_cl_many averaged 958 us per launch against 498 us for the one-run kernel with the SAME body: round 4's open question;
// its ISA: 210 flat_load + 170 flat_store where the one-run kernel has global_load / global_store).  Sent through an integer into the
// global address space a pointer is global again (free: the bits are the same; null stays null AND stays testable -- rebuilt as
// known_global_base + offset it was global too, but a GEP from a non-null base "cannot" be null, the compiler dropped the `p ? p[i] : ..`
// tests and the kernels read address 0).  Casts through address_space(1) and straight back, and __builtin_assume(!is_shared &
// !is_private), are folded away before the address-space inference sees them (ROCm 7.2).
template <class T, class B> __device__ __forceinline__ T *pc_as_global(T *p, const B * /* (the kernel's argument: not needed) */)
{
    typedef __attribute__((address_space(1))) char gchar;
    return (T *)(gchar *)(uintptr_t)p;
}
// the state of run blockIdx.y with every pointer member global (a member added to PcState must be added here: the size is checked)
static_assert(sizeof(PcState) == 944, "PcState changed: add the new pointer members to pc_many_state, then update this size");
__device__ __forceinline__ PcState pc_many_state(const PcManyRec *R, int run)
{
    PcState S = R[run].S;
#define G(f) S.f = pc_as_global(S.f, R)
    G(like.invcov); G(like.mean); G(prior.lo); G(prior.hi); G(dyn_loglikes); G(dyn_nlives); G(live); G(live_logL);
    G(live_cluster); G(live_pos); G(live_entry); G(cl_list); G(cl_n); G(logZp); G(logXp); G(logZXp);
    G(logZp2); G(logZpXp); G(logLp); G(XpXq); G(imin_slot); G(lse_ref); G(lse_sum); G(death_thr);
    G(cl_uid); G(chol); G(cov); G(logZp_dead); G(logZp2_dead); G(cl_uid_dead); G(phantom); G(ph_logL);
    G(ph_cuid); G(ph_uid); G(dead); G(dead_logw); G(dead_postX); G(dead_postZ); G(dead_cuid); G(dead_entry);
    G(babies); G(baby_logL); G(baby_logL_T); G(ch_cluster); G(ch_epoch); G(ch_nlike); G(ch_seed_slot); G(ch_contour);
    G(nhat); G(nhat_w); G(nhat_raw); G(nhat_Ms); G(ch_My); G(plan); G(sort_slot); G(sort_key);
    G(slot_src); G(slot_dead); G(slot_step); G(ch_nlike_g); G(logn); G(nn_list); G(nn_slot_owner); G(nn_chain_slot);
    G(nn_pts); G(nn_code); G(ctl); G(ctl_host);
#undef G
    return S;
}
// ... and the whole record: the state, the buffers (global too) and the integers; the members a kernel does not touch are never loaded
struct PcManyView { PcState S; void *p[10]; int ia[6]; };
__device__ __forceinline__ PcManyView pc_many_view(const PcManyRec *R, int run)
{
    PcManyView v;
    v.S = pc_many_state(R, run);
#pragma unroll
    for (int i = 0; i < 10; ++i) v.p[i] = pc_as_global(R[run].p[i], R);
#pragma unroll
    for (int i = 0; i < 6; ++i) v.ia[i] = R[run].ia[i];
    return v;
}
#endif

// Called by EVERY thread of the (single) workgroup at the end of a contraction kernel, after thread 0 has stored the new
// control block to *S.ctl.  The host mirror is fine-grained host memory over PCIe; a system-scope release (fence + L2
// write-back) in front of a stamp costs the kernel ~15 us, and a dozen dependent stores by one thread ~6 us, so: no
// ordering is asked for and the words leave in ONE store instruction, one lane each.  Each word carries the launch's
// sequence number in its upper half -- a word is valid when its stamp is the expected one, whatever order the words
// arrive in.  Only what the host's round loop needs travels; the rest of the control block is read with a copy at the
// moments that synchronise anyway (updates with host work, the end of the run).
#define PC_NOTE_WORDS 5
__device__ __forceinline__ void pc_publish_ctl(const PcState &S)
{
    __shared__ unsigned pc_note_sh[8];
    if (!S.ctl_host) return;                       // (uniform)
    if (threadIdx.x == 0) {
        const PcCtl *c = S.ctl;
        pc_note_sh[0] = (unsigned)c->status | ((unsigned)c->error << 8) | ((unsigned)(c->cluster_deleted != 0) << 16) |
                        ((unsigned)(c->upd_pending != 0) << 17) | ((unsigned)(c->upd_marks > 0x3FFF ? 0x3FFF : c->upd_marks) << 18);   // (a launch consumes <= 1024 chains: <= 1024 marks)
        pc_note_sh[1] = ((unsigned)c->i_nursery & 0xFFFFu) | ((unsigned)(c->upd_in < 0 ? 0 : (c->upd_in > 0xFFFF ? 0xFFFF : c->upd_in)) << 16);   // (a nursery holds <= 65535 chains: Engine::setup refuses a larger batch)
        pc_note_sh[2] = (unsigned)c->ndead; pc_note_sh[3] = (unsigned)c->nphantom;
        pc_note_sh[4] = (unsigned)c->ncluster | ((unsigned)(c->ncluster_dead & 0xFFFF) << 16);   // (ncluster <= 16384: Engine::grow_clusters; the full dead count comes with the block)
    }
    __syncthreads();
    if (threadIdx.x < PC_NOTE_WORDS)
        ((volatile unsigned long long *)S.ctl_host)[threadIdx.x] = ((unsigned long long)S.notify_seq << 32) | pc_note_sh[threadIdx.x];
}

// element g of a small settings array without dynamic indexing of the kernel argument block
__device__ __forceinline__ int pc_sel(const int (&a)[PC_MAX_GRADE], int g)
{
    int r = a[0];
#pragma unroll
    for (int k = 1; k < PC_MAX_GRADE; ++k) r = (g == k) ? a[k] : r;
    return r;
}
// grade of direction `col` (generation order): speeds(), chordal_sampling.f90:128
__device__ __forceinline__ int pc_grade_of(const PcState &S, int col)
{
    int g = 0;
#pragma unroll
    for (int k = 1; k < PC_MAX_GRADE; ++k) if (k < S.ngrade && col >= S.g_col0[k]) g = k;
    return g;
}
// block -> (grade, basis within the grade)
__device__ __forceinline__ void pc_grade_of_basis(const PcState &S, int block, int &g, int &b)
{
    g = 0; b = block;
#pragma unroll
    for (int k = 0; k + 1 < PC_MAX_GRADE; ++k) if (g == k && k + 1 < S.ngrade && b >= S.g_nb[k]) { b -= S.g_nb[k]; g = k + 1; }
}

// uniform number n of the single sequential stream (tests: the order the reference program consumes its generator)
__device__ __forceinline__ double pc_seq_uniform(const PcState &S, unsigned long long n)
{
    return pc_uniform(S.k0, S.k1, PC_DOM_SEQ, (uint32_t)(n >> 32), 0u, (uint32_t)n);
}
// ---- host side: dynamic LDS beyond the default limit.  hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE) and
// only ever has to grow.  Launch helpers are called from several host threads (the scheduler groups of pchip_run_repeats, runs on several
// devices): a guard that is a plain function-local static races (one thread lowers the limit between another's check and its launch, and a
// mark set for one device hides the need on the next).  One mutex, one high-water mark per (kernel, device); never lowered.
#include <mutex>
#include <map>
#include <utility>
inline void pc_need_dyn_lds(const void *kernel, size_t bytes)
{
    static std::mutex mtx;
    static std::map<std::pair<const void *, int>, size_t> mark;
    thread_local std::map<std::pair<const void *, int>, size_t> seen;      // what this thread knows to be set already (marks only grow)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(kernel, dev);
    size_t &mine = seen[key];
    if (bytes <= mine) return;
    std::lock_guard<std::mutex> g(mtx);
    size_t &m = mark[key];
    if (bytes > m) { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); m = bytes; }
    mine = m;
}


