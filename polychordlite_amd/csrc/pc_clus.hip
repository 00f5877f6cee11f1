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
// ONE wavefront runs the loop -- no barrier inside it, and on a single wave a dependent fp64 operation costs ~32 cycles, an
// exp or a division a dozen of those: the loop is written so that what is left of them sits off the chain of decisions.
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
__device__ __forceinline__ void cl_sort_live(const PcState &S, char *smem)
{
    int npow2 = 2;
    while (npow2 < S.Ncap) npow2 <<= 1;
    double *kv = (double *)smem;            // [npow2]
    int *kp = (int *)(kv + npow2);          // [npow2] list position
    int *ks = kp + npow2;                   // [npow2] slot
    const int tid = threadIdx.x;
    for (int i = tid; i < npow2; i += CL_NT) {
        const bool used = i < S.Ncap && S.live_cluster[i] >= 0;
        kv[i] = used ? S.live_logL[i] : PC_HUGE; kp[i] = used ? S.live_pos[i] : 0x7fffffff; ks[i] = i < S.Ncap ? i : -1;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += CL_NT) {
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
    for (int i = tid; i < NS; i += CL_NT) { S.sort_slot[i] = (i < npow2) ? ks[i] : -1; S.sort_key[i] = (i < npow2) ? d2key(kv[i]) : KEY_HUGE; }
}

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

extern "C" int pc_consume_cl_fits(const PcState *S, int nc)
{
    if (nc < 2 || nc > CL_MAXC || S->B > 1024 || S->nr > 64 * PC_MASK_WORDS || S->Ncap >= 65536) return 0;
    return cl_layout(S->Ncap, S->B, S->nr).total + 1024 <= (size_t)160 * 1024;
}

// (the runs of a launch: one shape -- live points, chains, repeats -- and one width J of the per-cluster registers)
extern "C" int pc_launch_consume_cl_many(const PcState *S, const PcManyRec *dR, int R, int wide, hipStream_t st)
{
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
