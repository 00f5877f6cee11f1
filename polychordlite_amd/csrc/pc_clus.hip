// pc_clus.hip -- the contraction with SEVERAL clusters and a static number of live points: k_consume_cl.
//
// Same decisions as k_consume (pc_contract.hip) -- replace_point / delete_outermost_point / update_evidence /
// find_min_loglikelihoods / more_samples_needed / the update trigger: src/polychord/run_time_info.f90:716-817, 211-296,
// 883-909, nested_sampling.F90:239-341, 514-543 -- for the case the clustered BASELINE configurations spend their time in.
// k_consume walks the nursery with a whole workgroup and pays, per consumed chain, two to seven workgroup barriers, a scan
// of every slot for the dying cluster's next minimum, six log-add-exps (an exp AND a log each) and a log-sum-exp over the
// clusters: 8.5 k cycles, 4.2 us, on one CU, 35 k times per run.  What the reference asks for per chain is much less:
//
//   * the point that dies is the GLOBAL minimum of the live set (minpos(logLp), run_time_info.f90:800): with the live set
//     sorted once per launch (k_sort_live) the deaths of the snapshot are a pointer walk, and the newcomers of the launch
//     -- each above the contour it was accepted at -- wait in a bitmap over their presorted ranks: no scan, no per-cluster
//     minimum (the per-cluster contours are rebuilt once, when the launch is over);
//   * update_evidence is a handful of multiply-adds per accumulator once everything is held in LINEAR space about
//     references that are fixed for the launch: the k-th death of a launch lies below the k-th smallest point of its
//     snapshot, so with Lhi = that bound (at most 300 nats above the first death: a launch ends where the next death would
//     leave this window, and the host starts the next one with new references) every exp(L - Lhi) is in (e^-300, 1], each
//     accumulator's reference is the larger of its value at launch and the largest term the launch can add, and the one
//     exponential of a death -- exp(L - Lhi) -- is known a death ahead.  The O(ncluster) cross terms are one cluster per
//     lane: lane q carries X_q, <Z X_q> and the factors that X_p X_q picked up since the launch began; the cross-volume
//     matrix itself is only READ (its row for the cluster expected to lose the next point is requested one death ahead)
//     and rewritten once at the end;
//   * live_logZ and the update trigger are sums over clusters of quantities that change in the dying and the receiving
//     cluster only: one DPP wave sum per accepted chain, compared in linear space;
//   * identify_cluster comes from the candidate lists of k_nn_lists, one baby per lane, the eight candidates' liveness
//     fetched side by side (two LDS round trips, one 16-byte record per slot);
//   * nothing is stored to HBM inside the loop (on gfx9 a wait for a prefetched load also waits for every store issued
//     before it): plan records, phantom masks and slot sources collect in LDS and leave together.
//
// ONE wavefront runs the loop -- no barrier inside it, and a single wave issues an instruction every 5-6 cycles (tools/dev/ubench_fp64.hip; "32 cycles a dependent fp64
// operation" until round 6 was a loop's own overhead), an exp or a division costs a hundred: the loop is written so that what is left of them sits off the chain of decisions.
// The other three waves of the workgroup stage the state before and write it back after.  Anything outside this kernel's
// envelope (dynamic nlive, the reference's list rule of the sequential-stream test mode, kill-off, live sets or cluster
// counts beyond the LDS) stays with k_consume, which is also the arbiter: settings.ablate bit 5 sends every launch there,
// and the two must produce the same run (tests/test_gpu_parity.py).  A cluster that dies ends the loop: the deletion
// (delete_cluster, run_time_info.f90:507-598) is done by the workgroup on the way out.
#include "pc_state.h"
#include "pc_keys.h"
#include <cstdlib>

#define CL_NT 256
#define CL_MAXC 128               /* two clusters per lane at most */
#define CL_WINDOW 300.0           /* nats between the first death of a launch and the last one it may make */

struct ClSlot { int c, p, o, src; };                       // cluster, list position, nn owner code, chain whose last baby owns the slot in this launch (-1)
struct ClChain { double last; int nlike, epoch, ca, rank; };
struct ClSorted { double L, e; int slot, pad; };             // logL, exp(logL - Lhi), slot (sorted snapshot) / chain (candidates)
struct ClCand { double L, e; int w, pad; };
struct ClHead { double logw, postXs, zl, contour; int dead_idx, dead_src; unsigned dead_cuid, ph_cuid; int ph_base, pad; };
struct ClOwn { double zp, zp2, zpx, kzp, kp2a, kp2b, kzpx, rzp, rzp2, rzpx; int touched, pad; };   // a cluster's own accumulators (linear) and their scales
// what wave 0 (the decisions) leaves for the waves behind it about a consumed chain, 16 bytes (the LDS block is full: 150 KB at nlive
// 1000 / 500 chains): kind 0 = nothing died; 1 = the point of cluster cd (nd live points before) in `slot` died with exp(L - Lhi) = eL
// and the chain's last baby joined ITS cluster (ClChain::ca; na points there after the death); predC = the cluster expected to lose the
// next point (-1: a newcomer).  a = kind | cd << 1 | (predC + 1) << 8 | slot << 16, b = nd | na << 16 (pc_consume_cl_fits: nlive < 65536)
struct ClEvt { double eL; unsigned a, b; };
__device__ __forceinline__ ClEvt cl_evt(int cd, int predC, int slot, int nd, int na, double eL)
{
    return ClEvt{eL, 1u | ((unsigned)cd << 1) | ((unsigned)(predC + 1) << 8) | ((unsigned)slot << 16), (unsigned)nd | ((unsigned)na << 16)};
}
#define CL_NO_LIMIT (-0x7fffffff)

struct ClLayout {                 // byte offsets into the dynamic LDS block; the same function sizes it on the host
    size_t slot, sL, sorted, cand, chain, head, own, logn, rcp, fg, masks, kmin, sCS, lst, lstOff, tag, evt, total;
};
__host__ __device__ inline ClLayout cl_layout(int Ncap, int B, int nr)
{
    ClLayout o{};
    size_t p = 0;
    const size_t NS = ((size_t)Ncap + 63) & ~(size_t)63, nw = ((size_t)nr + 63) / 64;
    auto take = [&](size_t &field, size_t bytes) { field = p; p += (bytes + 15) & ~(size_t)15; };
    take(o.slot, sizeof(ClSlot) * (size_t)Ncap); take(o.sL, 8 * (size_t)Ncap); take(o.sorted, sizeof(ClSorted) * (NS + 1));
    take(o.cand, sizeof(ClCand) * ((size_t)B + 1)); take(o.chain, sizeof(ClChain) * (size_t)B); take(o.head, sizeof(ClHead) * (size_t)B);
    take(o.own, sizeof(ClOwn) * CL_MAXC); take(o.logn, 8 * ((size_t)Ncap + 4)); take(o.rcp, 8 * ((size_t)Ncap + 4)); take(o.fg, 8 * 2 * CL_MAXC);
    take(o.masks, 8 * (size_t)B * nw); take(o.kmin, 8 * CL_MAXC);
    take(o.sCS, 4 * (size_t)B); take(o.lst, 4 * ((size_t)Ncap + B)); take(o.lstOff, 4 * (CL_MAXC + 1)); take(o.tag, 4 * ((size_t)Ncap + B + 1));
    take(o.evt, sizeof(ClEvt) * (size_t)B);
    o.total = p;
    return o;
}

// LDS words that another wave of the workgroup reads or writes while the loop runs (progress words, a slot's cluster / owner / position): relaxed
// atomics of workgroup scope = plain ds_read / ds_write that the compiler neither caches nor reorders against other memory operations.  As
// `*(volatile int *)&x` they were FLAT instructions (the address-space inference does not rewrite volatile accesses): slower than a ds
// operation, and counted on vmcnt -- every wait for one of them also waited for the next chain's records, requested a moment before from L2.
__device__ __forceinline__ int cl_lld(const int *p) { return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void cl_lst(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int cl_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double cl_unid(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int J> __device__ __forceinline__ double cl_get(const double (&v)[J], int c)
{
    if (J == 1) return readlane_f64(v[0], c);
    return c < 64 ? readlane_f64(v[0], c) : readlane_f64(v[J - 1], c - 64);
}
template <int J> __device__ __forceinline__ int cl_geti(const int (&v)[J], int c)
{
    if (J == 1) return __builtin_amdgcn_readlane(v[0], c);
    return c < 64 ? __builtin_amdgcn_readlane(v[0], c) : __builtin_amdgcn_readlane(v[J - 1], c - 64);
}
// the logarithm a linear accumulator stands for: ref + log(v); one that holds nothing keeps the reference's logzero
__device__ __forceinline__ double cl_log(double ref, double v, double logzero)
{
    const double x = v > 0.0 ? ref + log(v) : NEGBIG;
    return x > logzero ? x : logzero;
}

// the live slots by (logL, list position), free slots last -- k_sort_live's order (pc_fast.hip sort_live_body: the same comparator, a
// bitonic network over the next power of two), by this workgroup, in LDS that holds nothing any more; sort_slot / sort_key as that kernel
// leaves them (the ranks it also writes serve the candidate lists of the NEXT nursery, which the host sorts for)
template <int NT> __device__ __forceinline__ void cl_sort_live_nt(const PcState &S, char *smem);
__device__ __forceinline__ void cl_sort_live(const PcState &S, char *smem);
template <int NT> __device__ __forceinline__ void cl_sort_live_nt(const PcState &S, char *smem)
{
    int npow2 = 2;
    while (npow2 < S.Ncap) npow2 <<= 1;
    double *kv = (double *)smem;            // [npow2]
    int *kp = (int *)(kv + npow2);          // [npow2] list position
    int *ks = kp + npow2;                   // [npow2] slot
    const int tid = threadIdx.x;
    for (int i = tid; i < npow2; i += NT) {
        const bool used = i < S.Ncap && S.live_cluster[i] >= 0;
        kv[i] = used ? S.live_logL[i] : PC_HUGE; kp[i] = used ? S.live_pos[i] : 0x7fffffff; ks[i] = i < S.Ncap ? i : -1;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const double a = kv[i], b = kv[l]; const int pa = kp[i], pb = kp[l];
                    const bool gt = (a > b) || (a == b && pa > pb);
                    if (gt == up) { kv[i] = b; kv[l] = a; kp[i] = pb; kp[l] = pa; const int t = ks[i]; ks[i] = ks[l]; ks[l] = t; }
                }
            }
            __syncthreads();
        }
    const int NS = (S.Ncap + 63) & ~63;
    for (int i = tid; i < NS; i += NT) { S.sort_slot[i] = (i < npow2) ? ks[i] : -1; S.sort_key[i] = (i < npow2) ? d2key(kv[i]) : KEY_HUGE; }
}
__device__ __forceinline__ void cl_sort_live(const PcState &S, char *smem) { cl_sort_live_nt<CL_NT>(S, smem); }

template <int J>
__global__ __launch_bounds__(CL_NT) void k_consume_cl(PcState S)
{
#pragma clang fp contract(on)
#include "pc_consume_cl_body.inc"
}
// several runs in step: R one-wave contractions on R compute units (blockIdx.y = run).  (The body is included, not called: see k_slice_many)
template <int J>
__global__ __launch_bounds__(CL_NT) void k_consume_cl_many(const PcManyRec *__restrict__ R)
{
    const PcState S = pc_many_state(R, blockIdx.y);
    {
#pragma clang fp contract(on)
#include "pc_consume_cl_body.inc"
    }
}


// ------------------------------------------------------------------------------------------
// k_consume_clp -- the same contraction with its decisions made in parallel (pc_consume_clp_body.inc, round 6): eight waves; wave 0 resolves the
// acceptance vector 64 steps at a time, all waves derive what the consumed chains leave behind, waves 1 - 3 follow with the evidence work of
// k_consume_cl.  k_consume_cl stays: the arbiter next to the general kernel (settings.ablate bit 10) and the fallback for shapes whose LDS
// block this kernel's extra tables push over the limit.
#define CLP_NT 512
#define CLP_NEVER 0x7FFu
#define CLP_ALIVE 0x7fffffff
struct ClSL { double L, e; };                                  // sorted snapshot: logL, exp(logL - Lhi)
// a candidate's entry: cluster + 1 (0: not alive at any step of the launch) | step of birth + 1 (0: alive when the launch begins) << 8 | step of death << 19
__device__ __forceinline__ unsigned clp_tag(int cluster, int birth1, unsigned death) { return (unsigned)(cluster + 1) | ((unsigned)birth1 << 8) | (death << 19); }
struct ClpLayout {
    size_t slot, sL, sorted, sortSlot, cand, chain, head, own, logn, rcp, fg, masks, kmin, sCS, lst, lstOff, tag, evt,
           idxOf, rxS, below, kacc, U, stepOf, accw, clN, clX, ctot, total;
};
__host__ __device__ inline ClpLayout clp_layout(int Ncap, int B, int nr)
{
    ClpLayout o{};
    size_t p = 0;
    const size_t NS = ((size_t)Ncap + 63) & ~(size_t)63, nw = ((size_t)nr + 63) / 64;
    auto take = [&](size_t &field, size_t bytes) { field = p; p += (bytes + 15) & ~(size_t)15; };
    take(o.slot, sizeof(ClSlot) * (size_t)Ncap); take(o.sL, 8 * (size_t)Ncap); take(o.sorted, sizeof(ClSL) * (NS + 1)); take(o.sortSlot, 2 * (NS + 1));
    take(o.cand, sizeof(ClSL) * ((size_t)B + 1)); take(o.chain, sizeof(ClChain) * (size_t)B); take(o.head, sizeof(ClHead) * (size_t)B);
    take(o.own, sizeof(ClOwn) * CL_MAXC); take(o.logn, 8 * ((size_t)Ncap + 4)); take(o.rcp, 8 * ((size_t)Ncap + 4)); take(o.fg, 8 * 2 * CL_MAXC);
    take(o.masks, 8 * (size_t)B * nw); take(o.kmin, 8 * CL_MAXC);
    take(o.sCS, 4 * (size_t)B); take(o.lst, 4 * ((size_t)Ncap + B)); take(o.lstOff, 4 * (CL_MAXC + 1)); take(o.tag, 4 * ((size_t)Ncap + B + 1));
    take(o.evt, sizeof(ClEvt) * (size_t)B);
    // (rxS / below: where snapshot and candidates interleave; behind the order of deaths the same bytes hold the deaths' link, slot and cluster pair)
    take(o.idxOf, 2 * (size_t)Ncap); take(o.rxS, 2 * ((size_t)B + 2)); take(o.below, 2 * 2 * ((size_t)B + 2)); take(o.kacc, 2 * ((size_t)B + 2)); take(o.U, 2 * ((size_t)B + 2));
    take(o.stepOf, 2 * ((size_t)B + 2));
    take(o.accw, 8 * 16 + 4 * 20); take(o.clN, 4 * 5 * CL_MAXC); take(o.clX, 8 * 2 * CL_MAXC); take(o.ctot, 2 * CL_MAXC * (((size_t)B + 63) / 64));
    o.total = p;
    return o;
}
__device__ __forceinline__ void clp_sort_live(const PcState &S, char *smem) { cl_sort_live_nt<CLP_NT>(S, smem); }

template <int J>
__global__ __launch_bounds__(CLP_NT) void k_consume_clp(PcState S)
{
#pragma clang fp contract(on)
#include "pc_consume_clp_body.inc"
}
template <int J>
__global__ __launch_bounds__(CLP_NT) void k_consume_clp_many(const PcManyRec *__restrict__ R)
{
    const PcState S = pc_many_state(R, blockIdx.y);
    {
#pragma clang fp contract(on)
#include "pc_consume_clp_body.inc"
    }
}

// ------------------------------------------------------------------------------------------
// k_killoff_cl -- nested_sampling.F90:381-384 for a run that ends with several clusters: every remaining live point dies,
// lowest first (delete_outermost_point, run_time_info.f90:789-817; update_evidence :211-296; delete_cluster :507-598 when a
// cluster has lost its last point).  The general kernel (k_consume, final_mode 1) makes a death with a workgroup: two barriers,
// a scan of every slot for the dying cluster's next minimum, a dozen lanes of log-add-exps -- 4.5 us a death, a thousand deaths
// at BASELINE configs[2], and with runs in step one such tail per run.  Nothing is born here, so the ORDER of the deaths is the
// sorted order of the live set and everything a death needs but the evidence state is known beforehand:
//   * the workgroup sorts the live set by (logL, cluster, list position) in LDS (an exact tie in logL between clusters goes to
//     the lower cluster as in minpos; inside a cluster the general kernel's positions move with every death -- ties there are
//     broken by the position at the start);
//   * ONE wavefront makes the deaths, cluster q in lane q (as many lanes as clusters alive, in the list's order: a cluster's
//     end moves the lanes behind it up, so every wave-wide sum adds the same terms in the same places as the general kernel),
//     the six accumulations of the dying cluster and of the run in lanes 56-61: every lane one three-term log-sum-exp
//     m + log((e^(a-m) + e^(b-m)) + e^(c-m)) -- for the two-term ones c = -huge, and the sum is then exactly pc_logaddexp's
//     (one of its terms is e^0 = 1) -- so the wave does not diverge;
//   * the cross-volume matrix stays in LDS under the numbers the clusters had at launch (each lane knows its own): a death
//     rewrites the dying cluster's row and column; the row the NEXT death needs, that cluster's log n and the sorted record are
//     requested at the end of a death;
//   * no global LOAD inside the loop, so the per-death stores (log weight, logZ, cluster id) cost their issue slots only; the
//     volume column's logarithm and the rows leave behind the loop, all threads.
// The same statements in the same order as k_consume's kill_lowest: the two give the same numbers (tests/test_gpu_parity.py;
// settings.ablate bit 9 = the general kernel).  Not here: more than 56 clusters alive, a live set beyond the LDS.
#define KO_NT 512
#define KO_MAXC 56
struct KoLayout { size_t kv, kc, kp, ks, xq, lgn, os, total; int ld; };
static __host__ __device__ inline KoLayout ko_layout(int Ncap, int npow2, int nc)
{
    KoLayout l; size_t o = 0;
    l.ld = nc | 1;
    l.kv = o; o += sizeof(double) * (size_t)npow2;
    l.xq = o; o += sizeof(double) * (size_t)nc * l.ld;
    l.lgn = o; o += sizeof(double) * (size_t)(Ncap + 4);
    l.os = o; o += sizeof(double) * (size_t)npow2;
    l.kc = o; o += sizeof(int) * (size_t)npow2;
    l.kp = o; o += sizeof(int) * (size_t)npow2;
    l.ks = o; o += sizeof(int) * (size_t)npow2;
    l.total = o;
    return l;
}

__global__ __launch_bounds__(KO_NT) void k_killoff_cl(PcState S, int npow2, int nc0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KoLayout lay = ko_layout(S.Ncap, npow2, nc0);
    double *kv = (double *)(smem + lay.kv), *xq = (double *)(smem + lay.xq), *lgn = (double *)(smem + lay.lgn);
    double *os = (double *)(smem + lay.os), *om = kv;      // (the volume column's maximum takes the place of the dead point's logL: entry k is read before death k)
    int *kc = (int *)(smem + lay.kc), *kp = (int *)(smem + lay.kp), *ks = (int *)(smem + lay.ks);
    __shared__ int sh_nk, sh_ncd;
    const int tid = threadIdx.x, lane = tid & 63, Ncap = S.Ncap, nT = S.nT, maxc = S.maxc, ld = lay.ld;
    PcCtl *ctl = S.ctl;
    // ---- stage: the live set, the matrix, the table of logarithms
    for (int i = tid; i < npow2; i += KO_NT) {
        const int c = i < Ncap ? S.live_cluster[i] : -1;
        const bool used = c >= 0;
        kv[i] = used ? S.live_logL[i] : PC_HUGE; kc[i] = used ? c : 0x7fffffff; kp[i] = used ? S.live_pos[i] : 0x7fffffff; ks[i] = i < Ncap ? i : -1;
    }
    for (int e = tid; e < nc0 * nc0; e += KO_NT) xq[(size_t)(e / nc0) * ld + e % nc0] = S.XpXq[(size_t)(e / nc0) * maxc + e % nc0];
    for (int i = tid; i < Ncap + 4; i += KO_NT) lgn[i] = S.logn[i];
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += KO_NT) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const double a = kv[i], b = kv[l]; const int ca = kc[i], cb = kc[l], pa = kp[i], pb = kp[l];
                    const bool gt = (a > b) || (a == b && (ca > cb || (ca == cb && pa > pb)));
                    if (gt == up) { kv[i] = b; kv[l] = a; kc[i] = cb; kc[l] = ca; kp[i] = pb; kp[l] = pa; const int t = ks[i]; ks[i] = ks[l]; ks[l] = t; }
                }
            }
            __syncthreads();
        }
    const int ndead0 = ctl->ndead;
    if (tid < 64) {
        // ---- the deaths: one wavefront, no barrier
        int nc = nc0, nc_dead = ctl->ncluster_dead, ntot = 0;
        const bool cl = lane < nc0;
        double Xp = cl ? S.logXp[lane] : -PC_HUGE, ZXp = cl ? S.logZXp[lane] : 0.0, Zp = cl ? S.logZp[lane] : 0.0;
        double Zp2 = cl ? S.logZp2[lane] : 0.0, ZpXp = cl ? S.logZpXp[lane] : 0.0;
        int nq = cl ? S.cl_n[lane] : 0, orig = cl ? lane : -1;
        unsigned uid = cl ? S.cl_uid[lane] : 0u;
        for (int c = 0; c < nc0; ++c) ntot += __builtin_amdgcn_readlane(nq, c);
        double logZ = ctl->logZ, logZ2 = ctl->logZ2;
        const double log2v = log(2.0);
        int room = S.Dcap - ndead0;
        if (room < 0) room = 0;
        const int nk = ntot < room ? ntot : room;
        // what the first death needs
        int cdo = __builtin_amdgcn_readfirstlane(kc[0]);
        int cd = nk > 0 ? __ffsll((long long)__ballot(orig == cdo)) - 1 : 0;
        int n = __builtin_amdgcn_readlane(nq, cd);
        double l0 = lgn[n], l1 = lgn[n + 1], l2 = lgn[n + 2];
        double row = (lane < nc) ? xq[(size_t)cdo * ld + orig] : 0.0;
        double L = kv[0];
        for (int k = 0; k < nk; ++k) {
            const double Xpd = readlane_f64(Xp, cd), XX = readlane_f64(row, cd);
            const double ZXpd = readlane_f64(ZXp, cd), Zpd = readlane_f64(Zp, cd), Zp2d = readlane_f64(Zp2, cd), ZpXpd = readlane_f64(ZpXp, cd);
            const unsigned uidd = (unsigned)__builtin_amdgcn_readlane((int)uid, cd);
            const double logweight = Xpd - l1;
            // update_evidence (run_time_info.f90:211-296): every accumulation reads the state before the death
            double a = 0.0, b = -INFINITY, c3 = -INFINITY;       // (e^-inf = 0 exactly)
            if (lane < nc) { a = ZXp; b = row + L - l1; }
            if (lane == 56) { a = logZ; b = Xpd + L - l1; }
            if (lane == 57) { a = Zpd; b = Xpd + L - l1; }
            if (lane == 58) { a = logZ2; b = log2v + ZXpd + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; }
            if (lane == 59) { a = ZXpd + l0 - l1; b = XX + L + l0 - l1 - l2; }
            if (lane == 60) { a = Zp2d; b = log2v + ZpXpd + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; }
            if (lane == 61) { a = ZpXpd + l0 - l1; b = XX + L + l0 - l1 - l2; }
            const double m3 = fmax(a, fmax(b, c3));
            const double r = m3 + log(exp(a - m3) + exp(b - m3) + exp(c3 - m3));
            logZ = readlane_f64(r, 56); logZ2 = readlane_f64(r, 58);
            const double nZp = readlane_f64(r, 57), nZXp = readlane_f64(r, 59), nZp2 = readlane_f64(r, 60), nZpXp = readlane_f64(r, 61);
            if (lane < nc) {
                if (lane == cd) {
                    Zp = nZp; ZXp = nZXp; Zp2 = nZp2; ZpXp = nZpXp; Xp = Xpd + l0 - l1; nq = n - 1;
                    xq[(size_t)cdo * ld + cdo] = XX + l0 - l2;
                } else {
                    ZXp = r;
                    const double v = row + l0 - l1;
                    xq[(size_t)cdo * ld + orig] = v; xq[(size_t)orig * ld + cdo] = v;
                }
            }
            // the posterior stack's volume column: log sum_p X_p after the death, as a (maximum, sum) pair
            double lxm, lxs;
            if (nc == 1) { lxm = Xpd + l0 - l1; lxs = 1.0; }
            else {
                lxm = wave_max(lane < nc ? Xp : -PC_HUGE);
                lxs = wave_sum<4>(lane < nc ? exp(Xp - lxm) : 0.0);
            }
            if (lane == 0) {
                const int di = ndead0 + k;
                S.dead_logw[di] = logweight; S.dead_postZ[di] = logZ; S.dead_cuid[di] = uidd;
                om[k] = lxm; os[k] = lxs;
            }
            // delete_cluster (run_time_info.f90:507-598): the cluster's evidences go to the record of the dead, the lanes behind it move up
            if (n - 1 == 0) {
                if (lane == 0 && nc_dead < S.maxc_dead) { S.logZp_dead[nc_dead] = nZp; S.logZp2_dead[nc_dead] = nZp2; S.cl_uid_dead[nc_dead] = uidd; }
                nc_dead++;
                const double tX = __shfl_down(Xp, 1), tZX = __shfl_down(ZXp, 1), tZ = __shfl_down(Zp, 1), tZ2 = __shfl_down(Zp2, 1), tZpX = __shfl_down(ZpXp, 1);
                const int tn = __shfl_down(nq, 1), to = __shfl_down(orig, 1); const unsigned tu = (unsigned)__shfl_down((int)uid, 1);
                if (lane >= cd && lane < nc - 1) { Xp = tX; ZXp = tZX; Zp = tZ; Zp2 = tZ2; ZpXp = tZpX; nq = tn; orig = to; uid = tu; }
                if (lane == nc - 1) { Xp = -PC_HUGE; nq = 0; orig = -1; }
                nc--;
            }
            // what the next death needs (behind this death's writes to the matrix)
            if (k + 1 < nk) {
                cdo = __builtin_amdgcn_readfirstlane(kc[k + 1]);
                cd = __ffsll((long long)__ballot(orig == cdo)) - 1;
                n = __builtin_amdgcn_readlane(nq, cd);
                l0 = lgn[n]; l1 = lgn[n + 1]; l2 = lgn[n + 2];
                row = (lane < nc) ? xq[(size_t)cdo * ld + orig] : 0.0;
                L = kv[k + 1];
            }
        }
        if (lane == 0) {
            sh_nk = nk; sh_ncd = nc_dead;
            ctl->status = PC_ST_DONE; ctl->error = (nk < ntot) ? PC_ERR_DEAD_CAP : PC_ERR_NONE;
            ctl->ncluster = nc; ctl->ncluster_dead = nc_dead; ctl->ndead = ndead0 + nk;
            ctl->seg_hi = ctl->i_nursery - 1; ctl->seg_lo = ctl->i_nursery; ctl->cluster_deleted = 1;
            ctl->logZ = logZ; ctl->logZ2 = logZ2; ctl->live_logZ = S.logzero;
        }
    }
    __syncthreads();
    // ---- behind the loop, all threads: the volume column, the rows of the points that died (eight loads in flight), the slots' labels
    const int nk = sh_nk;
    for (int k = tid; k < nk; k += KO_NT) {
        S.dead_postX[ndead0 + k] = om[k] + log(os[k]);
        S.dead_entry[ndead0 + k] = S.live_entry[ks[k]];
        S.live_cluster[ks[k]] = -1;
    }
    const long long ne = (long long)nk * nT;
    for (long long e0 = tid; e0 < ne; e0 += (long long)KO_NT * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long e = e0 + (long long)u * KO_NT;
            if (e < ne) { const int k = (int)(e / nT), d = (int)(e - (long long)k * nT); v[u] = S.live[(size_t)ks[k] * nT + d]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long long e = e0 + (long long)u * KO_NT; if (e < ne) S.dead[(size_t)ndead0 * nT + e] = v[u]; }
    }
    for (int c = tid; c < nc0; c += KO_NT) S.cl_n[c] = 0;
    __syncthreads();
    pc_publish_ctl(S);
}

// 0: launched; 1: not this way (the caller takes the general kernel)
extern "C" int pc_launch_killoff_cl(const PcState *S, int nc, hipStream_t st)
{
    static const bool off = std::getenv("PC_KILLOFF_GENERAL") != nullptr;
    if (off || (S->ablate & 512) || nc < 2 || nc > KO_MAXC || S->seq_mode) return 1;
    int npow2 = 64;
    while (npow2 < S->Ncap) npow2 <<= 1;
    const size_t sh = ko_layout(S->Ncap, npow2, nc).total;
    if (sh + 1024 > (size_t)158 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_killoff_cl, sh);
    hipLaunchKernelGGL(k_killoff_cl, dim3(1), dim3(KO_NT), sh, st, *S, npow2, nc);
    return 0;
}

extern "C" int pc_consume_cl_fits(const PcState *S, int nc)
{
    if (nc < 2 || nc > CL_MAXC || S->B > 1024 || S->nr > 64 * PC_MASK_WORDS || S->Ncap >= 65536) return 0;
    return cl_layout(S->Ncap, S->B, S->nr).total + 1024 <= (size_t)160 * 1024;
}

// the kernel with parallel decisions: the same envelope, its own (larger) LDS block, at most 127 clusters in a death's packed record
extern "C" int pc_consume_clp_fits(const PcState *S, int nc)
{
    static const bool off = std::getenv("PC_CONSUME_CLP_OFF") != nullptr;
    if (off || (S->ablate & 1024) || !pc_consume_cl_fits(S, nc)) return 0;
    int npow2 = 2;
    while (npow2 < S->Ncap) npow2 <<= 1;
    if ((size_t)npow2 * 16 + 1024 > (size_t)160 * 1024) return 0;      // (the in-kernel sort of the live set between passes)
    return clp_layout(S->Ncap, S->B, S->nr).total + 1024 <= (size_t)160 * 1024;
}

// (the runs of a launch: one shape -- live points, chains, repeats -- and one width J of the per-cluster registers)
extern "C" int pc_launch_consume_cl_many(const PcState *S, const PcManyRec *dR, int R, int wide, hipStream_t st)
{
    if (pc_consume_clp_fits(S, 2)) {
        const size_t shp = clp_layout(S->Ncap, S->B, S->nr).total;
        if (!wide) { pc_need_dyn_lds((const void *)k_consume_clp_many<1>, shp); hipLaunchKernelGGL(k_consume_clp_many<1>, dim3(1, R), dim3(CLP_NT), shp, st, dR); }
        else { pc_need_dyn_lds((const void *)k_consume_clp_many<2>, shp); hipLaunchKernelGGL(k_consume_clp_many<2>, dim3(1, R), dim3(CLP_NT), shp, st, dR); }
        return 0;
    }
    const size_t sh = cl_layout(S->Ncap, S->B, S->nr).total;
    if (!wide) {
        pc_need_dyn_lds((const void *)k_consume_cl_many<1>, sh);
        hipLaunchKernelGGL(k_consume_cl_many<1>, dim3(1, R), dim3(CL_NT), sh, st, dR);
    } else {
        pc_need_dyn_lds((const void *)k_consume_cl_many<2>, sh);
        hipLaunchKernelGGL(k_consume_cl_many<2>, dim3(1, R), dim3(CL_NT), sh, st, dR);
    }
    return 0;
}

extern "C" int pc_launch_consume_cl(const PcState *S, int nc, hipStream_t st)
{
    if (pc_consume_clp_fits(S, nc)) {
        const size_t shp = clp_layout(S->Ncap, S->B, S->nr).total;
        if (nc <= 64) { pc_need_dyn_lds((const void *)k_consume_clp<1>, shp); hipLaunchKernelGGL(k_consume_clp<1>, dim3(1), dim3(CLP_NT), shp, st, *S); }
        else { pc_need_dyn_lds((const void *)k_consume_clp<2>, shp); hipLaunchKernelGGL(k_consume_clp<2>, dim3(1), dim3(CLP_NT), shp, st, *S); }
        return 0;
    }
    const size_t sh = cl_layout(S->Ncap, S->B, S->nr).total;
    if (nc <= 64) {
        pc_need_dyn_lds((const void *)k_consume_cl<1>, sh);
        hipLaunchKernelGGL(k_consume_cl<1>, dim3(1), dim3(CL_NT), sh, st, *S);
    } else {
        pc_need_dyn_lds((const void *)k_consume_cl<2>, sh);
        hipLaunchKernelGGL(k_consume_cl<2>, dim3(1), dim3(CL_NT), sh, st, *S);
    }
    return 0;
}
