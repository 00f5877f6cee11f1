// pc_contract.hip -- nested-sampling contraction on the device.
//
//   k_consume   ONE workgroup walks the nursery in the reference's order and takes every decision
//               of replace_point / delete_outermost_point / update_evidence / find_min_loglikelihoods /
//               more_samples_needed / delete_cluster (run_time_info.f90:716-817, 211-296, 883-909,
//               507-598; nested_sampling.F90:239-341, 514-543).  The live logL column and the
//               (cluster, list position) labels of every slot sit in LDS; the independent log-space
//               updates of one death are spread over lanes; min-logL scans are DPP argmin butterflies.
//               No point row is moved here: the kernel emits a PLAN.
//   k_apply_*   execute the plan in parallel over the whole chip (dead rows, phantoms, new live rows).
//   k_clean_* / k_cov_* / k_chol   the "update" step: clean_phantoms (run_time_info.f90:820-877) as a
//               threshold test + deterministic stream compaction, calculate_covmats (:601-641) as a
//               two-pass mean / centred SYRK with fixed-order reduction, calc_cholesky (utils.F90:621-649).
#include "pc_state.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

// ------------------------------------------------------------------------------------------
// block-level helpers (NT threads; NT == 64 needs no barrier traffic)
// ------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ vk_t block_argmin(vk_t v, vk_t *scratch)
{
    v = wave_argmin(v);
    if (NT == 64) return v;
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wid] = v;
    __syncthreads();
    vk_t r = scratch[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = vk_min(r, scratch[i]);
    __syncthreads();
    return r;
}

struct ConsumeShared {
    double *sL; int *sC; int *sP;                      // per slot
    double *cLogLp, *cLogXp, *cLogZp, *cLogZXp, *cLogZp2, *cLogZpXp, *cLseRef, *cLseSum, *cThr;
    int *cN, *cMinSlot; unsigned *cUid;
    double *jobres;                                     // [NT] results of the lane-parallel jobs
    vk_t *red;                                          // [16]
    double *xbuf;                                       // [IDG][D] cube coordinates of the babies being identified
    double *sX;                                         // [xrows][D] cube coordinates: all live points (xrows == Ncap) or one tile
    int xrows;
    double *gd2; int *gkey; int *ids;                   // [IDG][IDP] partial minima; [nr] cluster of each baby
    int *misc;                                          // small scratch
    int *sO, *sCS;                                      // nn_slot_owner [Ncap], nn_chain_slot [B] (only when S.nn_valid)
    double *xq; int xq_ld;                              // log <X_p X_q>: LDS copy [xq_ld][xq_ld] when the clusters fit, else S.XpXq [maxc][maxc]
};

// identify_cluster (run_time_info.f90:913-949) for ALL babies of a chain at once: the cluster of the
// nearest live point over all clusters; ties keep the first point in (cluster, list position) order
// like the reference's strict '<' scan.  Babies are taken in groups of IDG; inside a group IDP = NT/IDG
// threads share one baby, each scanning a stride of the live slots; the live coordinates are served
// from the LDS cache H.sX when it fits (updated on every insertion), else from HBM.
#define PC_IDG 32
template <int NT>
__device__ void block_identify_all(const PcState &S, const ConsumeShared &H, int w, const double *blog, double Lg, int nc,
                                   bool only_unresolved = false)
{
    const int tid = threadIdx.x, D = S.D, nr = S.nr, nT = S.nT;
    constexpr int IDG = (NT >= 1024) ? PC_IDG : 4, IDP = NT / IDG;
    if (!only_unresolved) for (int i = tid; i < nr; i += NT) H.ids[i] = -1;
    if (nc == 1) { __syncthreads(); for (int i = tid; i < nr; i += NT) H.ids[i] = 0; __syncthreads(); return; }
    for (int g0 = 0; g0 < nr; g0 += IDG) {
        __syncthreads();
        if (only_unresolved) {                          // skip groups that the candidate lists settled completely
            bool need = false;
            for (int g = 0; g < IDG && g0 + g < nr; ++g) need = need || (H.ids[g0 + g] == -2);
            if (!need) continue;
        }
        for (int e = tid; e < IDG * D; e += NT) {
            const int g = e / D, d = e % D, i = g0 + g;
            H.xbuf[e] = (i < nr) ? S.babies[((size_t)w * nr + i) * nT + d] : 0.0;
        }
        __syncthreads();
        const int g = tid / IDP, p = tid % IDP, i = g0 + g;
        double bd = PC_HUGE; int bk = 0x7fffffff;
        const bool mine = i < nr && blog[i] > Lg && (!only_unresolved || H.ids[i] == -2);
        const double *x = H.xbuf + (size_t)g * D;
        if (H.xrows >= S.Ncap) {                       // every live coordinate is resident in LDS
            if (mine)
                for (int s = p; s < S.Ncap; s += IDP) {
                    const int c = H.sC[s];
                    if (c < 0) continue;
                    const double *q = H.sX + (size_t)s * D;
                    double d2 = 0.0;
                    for (int d = 0; d < D; ++d) { const double t = x[d] - q[d]; d2 += t * t; }
                    const int key = c * S.Ncap + H.sP[s];
                    if (d2 < bd || (d2 == bd && key < bk)) { bd = d2; bk = key; }
                }
        } else {                                       // stream the live coordinates through an LDS tile
            for (int t0 = 0; t0 < S.Ncap; t0 += H.xrows) {
                const int tn = min(H.xrows, S.Ncap - t0);
                __syncthreads();
                for (int e = tid; e < tn * D; e += NT) {
                    const int s = t0 + e / D, d = e % D, src = S.slot_src[s];
                    H.sX[e] = (src >= 0) ? S.babies[((size_t)src * nr + (nr - 1)) * nT + d] : S.live[(size_t)s * nT + d];
                }
                __syncthreads();
                if (mine)
                    for (int sl = p; sl < tn; sl += IDP) {
                        const int s = t0 + sl, c = H.sC[s];
                        if (c < 0) continue;
                        const double *q = H.sX + (size_t)sl * D;
                        double d2 = 0.0;
                        for (int d = 0; d < D; ++d) { const double t = x[d] - q[d]; d2 += t * t; }
                        const int key = c * S.Ncap + H.sP[s];
                        if (d2 < bd || (d2 == bd && key < bk)) { bd = d2; bk = key; }
                    }
            }
        }
        H.gd2[tid] = bd; H.gkey[tid] = bk;
        __syncthreads();
        if (tid < IDG && g0 + tid < nr && (!only_unresolved || H.ids[g0 + tid] == -2)) {
            double md = PC_HUGE; int mk = 0x7fffffff;
            for (int q = 0; q < IDP; ++q) {
                const double v = H.gd2[tid * IDP + q]; const int k = H.gkey[tid * IDP + q];
                if (v < md || (v == md && k < mk)) { md = v; mk = k; }
            }
            H.ids[g0 + tid] = (mk == 0x7fffffff) ? -1 : mk / S.Ncap;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// k_nn_lists: the nearest-cluster search of a whole nursery, done once by the whole chip.
//
// identify_cluster (run_time_info.f90:913-949) asks, for every baby, for the nearest live point AT THE MOMENT ITS
// CHAIN IS CONSUMED; between the consumption of two chains the live set changes by one point, so the reference's
// answer is a property of an evolving set and the serial contraction used to recompute N x D distances per baby on one
// CU.  But every point that can be alive then is known now (T0): the live points of T0 and the last babies of the
// chains consumed before this one (reverse nursery order: chains w' > w).  One workgroup per unconsumed chain ranks
// that superset for each of its babies and keeps the PC_NN_K nearest, ascending; the contraction then only walks the
// list for the first entry that is still / already alive (exact: everything nearer is dead), and falls back to the
// full search when a list is exhausted.
//   256 threads = 32 babies x 8 partial scanners; points are staged through an LDS tile.
// ------------------------------------------------------------------------------------------
#define NNL_G 32
#define NNL_P 8
__device__ __forceinline__ void nn_lists_body(const PcState &S, int nleft, int tile_pts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, D = S.D, nr = S.nr, nT = S.nT, Ncap = S.Ncap;
    const int w = blockIdx.x;                         // chain, w < nleft = entries still in the nursery
    double *xb = (double *)smem;                      // [NNL_G][D] babies of the group
    double *pts = xb + (size_t)NNL_G * D;             // [tile_pts][D]
    double *md = pts + (size_t)tile_pts * D;          // [256][PC_NN_K] merge buffer
    int *mc = (int *)(md + 256 * PC_NN_K);            // [256][PC_NN_K]
    int *pcode = mc + 256 * PC_NN_K;                  // [tile_pts] code of each staged point, PC_NN_NONE = skip
    if (w == 0) {                                     // liveness bookkeeping starts now
        for (int s = tid; s < Ncap; s += 256) S.nn_slot_owner[s] = -1;
        for (int c = tid; c < S.B; c += 256) S.nn_chain_slot[c] = -1;
    }
    const int nc = S.ctl->ncluster;
    double Lg0 = S.logLp[0];
    for (int c = 1; c < nc; ++c) Lg0 = fmin(Lg0, S.logLp[c]);
    const double *blog = S.baby_logL + (size_t)w * nr;
    const int ncand = nleft - 1 - w;                  // chains w+1 .. nleft-1 are consumed before w
    const int npts = Ncap + ncand;
    const int g = tid / NNL_P, p = tid % NNL_P;
    for (int g0 = 0; g0 < nr; g0 += NNL_G) {
        __syncthreads();
        for (int e = tid; e < NNL_G * D; e += 256) {
            const int gg = e / D, d = e % D, i = g0 + gg;
            xb[e] = (i < nr) ? S.babies[((size_t)w * nr + i) * nT + d] : 0.0;
        }
        const int i = g0 + g;
        const bool mine = i < nr && blog[i] > Lg0;    // the contour only rises: others never need a cluster
        double bd[PC_NN_K]; int bc[PC_NN_K];
#pragma unroll
        for (int k = 0; k < PC_NN_K; ++k) { bd[k] = PC_HUGE; bc[k] = PC_NN_NONE; }
        for (int t0 = 0; t0 < npts; t0 += tile_pts) {
            const int tn = min(tile_pts, npts - t0);
            __syncthreads();
            for (int q = tid; q < tn; q += 256) {
                const int gi = t0 + q;
                int code = PC_NN_NONE;
                if (gi < Ncap) { if (S.live_cluster[gi] >= 0) code = gi; }
                else code = -(1 + (w + 1 + (gi - Ncap)));
                pcode[q] = code;
            }
            __syncthreads();
            for (int e = tid; e < tn * D; e += 256) {
                const int q = e / D, d = e % D, code = pcode[q];
                double v = 0.0;
                if (code >= 0) { const int src = S.slot_src[code]; v = (src >= 0) ? S.babies[((size_t)src * nr + (nr - 1)) * nT + d] : S.live[(size_t)code * nT + d]; }
                else if (code != PC_NN_NONE) v = S.babies[((size_t)(-(1 + code)) * nr + (nr - 1)) * nT + d];
                pts[e] = v;
            }
            __syncthreads();
            if (mine) {
                const double *x = xb + (size_t)g * D;
                for (int q = p; q < tn; q += NNL_P) {
                    const int code = pcode[q];
                    if (code == PC_NN_NONE) continue;
                    const double *y = pts + (size_t)q * D;
                    double d2 = 0.0;
                    for (int d = 0; d < D; ++d) { const double t = x[d] - y[d]; d2 += t * t; }
                    if (d2 < bd[PC_NN_K - 1]) {       // sorted insertion, registers only
                        double cd = d2; int cc = code;
#pragma unroll
                        for (int k = 0; k < PC_NN_K; ++k) {
                            const bool sw = cd < bd[k];
                            const double td = sw ? bd[k] : cd; const int tc = sw ? bc[k] : cc;
                            bd[k] = sw ? cd : bd[k]; bc[k] = sw ? cc : bc[k];
                            cd = td; cc = tc;
                        }
                    }
                }
            }
        }
        // merge the NNL_P partial lists of a baby: each is sorted, NNL_P-way pick by the first scanner
#pragma unroll
        for (int k = 0; k < PC_NN_K; ++k) { md[tid * PC_NN_K + k] = bd[k]; mc[tid * PC_NN_K + k] = bc[k]; }
        __syncthreads();
        if (p == 0 && i < nr) {
            int head[NNL_P];
#pragma unroll
            for (int q = 0; q < NNL_P; ++q) head[q] = 0;
            int out[PC_NN_K];
#pragma unroll
            for (int k = 0; k < PC_NN_K; ++k) {
                double best = PC_HUGE; int bq = -1;
#pragma unroll
                for (int q = 0; q < NNL_P; ++q) {
                    const double v = (head[q] < PC_NN_K) ? md[(g * NNL_P + q) * PC_NN_K + head[q]] : PC_HUGE;
                    if (v < best) { best = v; bq = q; }
                }
                int code = PC_NN_NONE;
#pragma unroll
                for (int q = 0; q < NNL_P; ++q) if (q == bq) { code = mc[(g * NNL_P + q) * PC_NN_K + head[q]]; head[q]++; }
                out[k] = mine ? code : PC_NN_NONE;
            }
            int4 *dst = (int4 *)(S.nn_list + ((size_t)w * nr + i) * PC_NN_K);
            dst[0] = make_int4(out[0], out[1], out[2], out[3]);
            dst[1] = make_int4(out[4], out[5], out[6], out[7]);
        }
    }
}
__global__ __launch_bounds__(256) void k_nn_lists(PcState S, int nleft, int tile_pts) { nn_lists_body(S, nleft, tile_pts); }
// several runs in step (blockIdx.y = run): each run its own number of chains left in the nursery (PcManyRec::ia[1])
__global__ __launch_bounds__(256) void k_nn_lists_many(const PcManyRec *__restrict__ R, int tile_pts)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    if ((int)blockIdx.x >= r.ia[1]) return;
    nn_lists_body(r.S, r.ia[1], tile_pts);
}

// ------------------------------------------------------------------------------------------
// k_nn_lists_d<D> (nDims <= 32): the same lists from registers.  The kernel above reads both operands of every (x - y)^2 from LDS
// and fills 32 x 8 threads with the babies of a chain in groups of 32 (num_repeats = 40: 62 % of the slots); with several runs of a
// device in step it is the heaviest kernel of a round (1.6 ms for sixteen runs at BASELINE configs[3]).  Here a thread holds TWO
// babies' coordinates in registers (the loop over d is unrolled: one instantiation per nDims), the pairs of a chain share the
// workgroup with as many scanners each as fit (num_repeats 40: 20 pairs x 12 scanners = 240 threads), a staged point is read
// from LDS once for two babies, and a thread walks two points at a time (four sums side by side).  The same sums in the same order (d ascending, fused multiply-add of the
// difference), the same lists: ties between different points -- exact duplicates only -- go to the lower scanner, then the
// lower point, as above.
// ------------------------------------------------------------------------------------------
// the candidates of a nursery, once: every workgroup of k_nn_lists_d stages its tiles from this array with plain consecutive loads
// (resolving slot -> occupant -> row per workgroup was a chain of three dependent global loads per staged point: at nDims = 10 a
// workgroup spent four fifths of its time there)
// (use_rank: k_sort_live has just ranked the live set -- the snapshot's points go to the place of their rank, death order, and the empty
//  slots nowhere; k_nn_lists_d then finds the points that cannot die before a chain is looked at as ONE range.  Else: slot order.)
__device__ __forceinline__ void nn_gather_body(const PcState &S, int nleft, int use_rank)
{
    const int D = S.D, nr = S.nr, nT = S.nT, Ncap = S.Ncap;
    const int e = blockIdx.x * 256 + threadIdx.x, n = (Ncap + nleft) * D;
    if (e >= n) return;
    const int gi = e / D, d = e - gi * D;
    int code = PC_NN_NONE, pos = gi; double v = 0.0;
    if (gi < Ncap) {
        if (S.live_cluster[gi] >= 0) {
            code = gi; const int src = S.slot_src[gi]; v = (src >= 0) ? S.babies[((size_t)src * nr + (nr - 1)) * nT + d] : S.live[(size_t)gi * nT + d];
            if (use_rank) pos = S.nn_code[(size_t)Ncap + S.B + gi];
        } else if (use_rank) return;
        if (pos < 0 || pos >= Ncap) return;           // (a slot the sort did not see: not a candidate of this nursery)
    } else { const int c = gi - Ncap; code = -(1 + c); v = S.babies[((size_t)c * nr + (nr - 1)) * nT + d]; }
    S.nn_pts[(size_t)pos * D + d] = v;
    if (d == 0) S.nn_code[pos] = code;
}
__global__ __launch_bounds__(256) void k_nn_gather(PcState S, int nleft, int use_rank) { nn_gather_body(S, nleft, use_rank); }
__global__ __launch_bounds__(256) void k_nn_gather_many(const PcManyRec *__restrict__ R, int use_rank) { const PcManyView r = pc_many_view(R, blockIdx.y); nn_gather_body(r.S, r.ia[1], use_rank); }

#define NND_SC 16                                   /* at most this many scanners per pair of babies */
// Round 4: most of the candidates cannot die before the chain is looked at.  The deaths of a nursery take the snapshot's points in
// ascending logL (k_sort_live's order); before chain w is consumed at most ncand = nleft - 1 - w chains were, so a snapshot point of
// rank >= ncand is CERTAINLY alive then ("safe").  The nearest safe point ends every walk of a list: what the contraction needs is the
// candidates that may be dead by then ("uncertain": the ncand lowest snapshot points, the last babies of the chains before) that are
// NEARER than the nearest safe point, ascending, and then that point -- the old list of the eight nearest, cut off behind its first safe
// entry.  So a safe point costs a comparison (running minimum), and an uncertain one is inserted only if it beats that minimum: the
// sorted insertion, three quarters of the kernel's instructions while every candidate went through it, is now rare.  use_rank = 0 (no
// sorted order at hand): every candidate is uncertain, the old lists.
template <int D>
__device__ __forceinline__ void nn_lists_d_body(const PcState &S, int nleft, int tile_pts, int use_rank)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DP = D | 1;                         // odd row stride: the scanners of a wave read different banks
    const int tid = threadIdx.x, nr = S.nr, nT = S.nT, Ncap = S.Ncap;
    const int w = blockIdx.x;                         // chain, w < nleft = entries still in the nursery
    double *pts = (double *)smem;                     // [tile_pts][DP]
    int *pcode = (int *)(pts + (size_t)tile_pts * DP);   // [tile_pts] code of each staged point, PC_NN_NONE = skip
    double *md = (double *)smem;                      // [256][2][PC_NN_K] merge buffer: over the tile, once the last one has been scanned
    int *mc = (int *)(md + 256 * 2 * PC_NN_K);        // [256][2][PC_NN_K]
    double *msd = (double *)smem;                     // [256][2] the scanners' nearest safe points (between the two scans)
    int *msc = (int *)(msd + 256 * 2);                // [256][2]
    if (w == 0) {                                     // liveness bookkeeping starts now
        for (int s = tid; s < Ncap; s += 256) S.nn_slot_owner[s] = -1;
        for (int c = tid; c < S.B; c += 256) S.nn_chain_slot[c] = -1;
    }
#ifdef NND_DBG
    long long nd_t[5] = {0, 0, 0, 0, 0}; long long nd_c = clock64(); const long long nd_start = nd_c;
#define NND_MARK(i) { const long long t_ = clock64(); nd_t[i] += t_ - nd_c; nd_c = t_; }
#else
#define NND_MARK(i)
#endif
    const int nc = S.ctl->ncluster;
    double Lg0 = S.logLp[0];
    for (int c = 1; c < nc; ++c) Lg0 = fmin(Lg0, S.logLp[c]);
    const double *blog = S.baby_logL + (size_t)w * nr;
    const int ncand = nleft - 1 - w;                  // chains w+1 .. nleft-1 are consumed before w
    // The gathered array (k_nn_gather).  use_rank: the snapshot's points stand in their death order (k_sort_live), n of them: the first
    // min(ncand, n) may be dead by the time this chain is looked at, the others cannot.  Without a sorted order at hand: slot order, all
    // of them uncertain (empty slots carry PC_NN_NONE).  Behind them, at Ncap, the last babies of the nursery's chains.
    const int *nn_rank = S.nn_code + (size_t)Ncap + S.B;
    const int nsnap = use_rank ? nn_rank[Ncap] : Ncap;
    const int nunc = use_rank ? (ncand < nsnap ? ncand : nsnap) : Ncap;      // uncertain snapshot entries: [0, nunc)
    const int nsafe = nsnap - nunc;                                          // safe ones: [nunc, nsnap)
    const int npair = (nr + 1) / 2;
    const int PG = npair < 256 ? npair : 256;         // pairs per pass
    const int nscan = (256 / PG) < NND_SC ? (256 / PG) : NND_SC;
    const int pi = tid / nscan, p = tid - pi * nscan;
    for (int pg0 = 0; pg0 < npair; pg0 += PG) {
        const bool act = pi < PG && pg0 + pi < npair;
        const int ia = 2 * (pg0 + pi), ib = ia + 1;
        double xa[D], xb[D];
        {
            const double *ra = S.babies + ((size_t)w * nr + (act ? ia : 0)) * nT, *rb = S.babies + ((size_t)w * nr + ((act && ib < nr) ? ib : 0)) * nT;
#pragma unroll
            for (int d = 0; d < D; ++d) { xa[d] = ra[d]; xb[d] = rb[d]; }
        }
        const bool minea = act && blog[ia] > Lg0;                 // the contour only rises: others never need a cluster
        const bool mineb = act && ib < nr && blog[ib] > Lg0;
        double bda[PC_NN_K], bdb[PC_NN_K]; int bca[PC_NN_K], bcb[PC_NN_K];
#pragma unroll
        for (int k = 0; k < PC_NN_K; ++k) { bda[k] = PC_HUGE; bdb[k] = PC_HUGE; bca[k] = PC_NN_NONE; bcb[k] = PC_NN_NONE; }
        double sa = PC_HUGE, sb = PC_HUGE; int sca = PC_NN_NONE, scb = PC_NN_NONE;      // nearest safe point: this scanner's, then the pair's
        auto insert = [&](double (&bd)[PC_NN_K], int (&bc)[PC_NN_K], double d2, int code) __attribute__((always_inline)) {
            double cd = d2; int cc = code;                        // sorted insertion, registers only (the caller has compared with the last entry)
#pragma unroll
            for (int k = 0; k < PC_NN_K; ++k) {
                const bool sw = cd < bd[k];
                const double td = sw ? bd[k] : cd; const int tc = sw ? bc[k] : cc;
                bd[k] = sw ? cd : bd[k]; bc[k] = sw ? cc : bc[k];
                cd = td; cc = tc;
            }
        };
        // one scan over `count` candidates, entry v of it = entry first + v of the gathered array (v < split) or the baby of chain
        // w + 1 + (v - split).  SAFE: a running minimum a point; else a sorted insertion for the few that beat the pair's nearest safe point.
        auto scan = [&](auto safe_tag, int first, int split, int count) __attribute__((always_inline)) {
            constexpr bool SAFE = decltype(safe_tag)::value;
            // (no early way out for an empty slot: with the sums used only inside a branch the compiler moved each of them INTO it, one
            //  dependent chain behind the other -- three times the latency of the four side by side)
            auto take = [&](double a0, double b0, int c0) __attribute__((always_inline)) {
                const bool is = c0 != PC_NN_NONE;
                if constexpr (SAFE) {
                    const bool ua = is & (a0 < sa), ub = is & (b0 < sb);
                    sa = ua ? a0 : sa; sca = ua ? c0 : sca;
                    sb = ub ? b0 : sb; scb = ub ? c0 : scb;
                } else {
                    if (is & minea & (a0 < bda[PC_NN_K - 1]) & (a0 < sa)) insert(bda, bca, a0, c0);
                    if (is & mineb & (b0 < bdb[PC_NN_K - 1]) & (b0 < sb)) insert(bdb, bcb, b0, c0);
                }
            };
            for (int t0 = 0; t0 < count; t0 += tile_pts) {
                const int tn = min(tile_pts, count - t0);
                __syncthreads();
                NND_MARK(2)
                for (int q = tid; q < tn; q += 256) { const int v = t0 + q; pcode[q] = S.nn_code[v < split ? first + v : Ncap + w + 1 + (v - split)]; }
                for (int e = tid; e < tn * D; e += 256) {
                    const int q = e / D, d = e - q * D, v = t0 + q;
                    pts[(size_t)q * DP + d] = S.nn_pts[(size_t)(v < split ? first + v : Ncap + w + 1 + (v - split)) * D + d];
                }
                __syncthreads();
                NND_MARK(1)
                if (minea || mineb) {
                    int q = p;
                    for (; q + nscan < tn; q += 2 * nscan) {          // two points at a time: four sums in flight
                        const int c0 = pcode[q], c1 = pcode[q + nscan];
                        const double *y0 = pts + (size_t)q * DP, *y1 = pts + (size_t)(q + nscan) * DP;
                        double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const double u0 = y0[d], u1 = y1[d];
                            const double ta0 = xa[d] - u0, tb0 = xb[d] - u0, ta1 = xa[d] - u1, tb1 = xb[d] - u1;
                            a0 = fma(ta0, ta0, a0); b0 = fma(tb0, tb0, b0); a1 = fma(ta1, ta1, a1); b1 = fma(tb1, tb1, b1);
                        }
                        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));      // (the four sums are made HERE, side by side)
                        take(a0, b0, c0); take(a1, b1, c1);
                    }
                    if (q < tn) {
                        const int c0 = pcode[q];
                        const double *y0 = pts + (size_t)q * DP;
                        double a0 = 0.0, b0 = 0.0;
#pragma unroll
                        for (int d = 0; d < D; ++d) { const double u0 = y0[d]; const double ta0 = xa[d] - u0, tb0 = xb[d] - u0; a0 = fma(ta0, ta0, a0); b0 = fma(tb0, tb0, b0); }
                        take(a0, b0, c0);
                    }
                }
            }
        };
        // Round 6: the safe points FIRST.  The list a walk needs is the uncertain candidates nearer than the pair's nearest safe point, and
        // then that point; while safe and uncertain candidates came mixed, a scanner measured an uncertain one against the nearest safe
        // point IT had seen so far, and nearly every iteration some lane of the wave made a sorted insertion -- eight compare-and-swap
        // steps on two register arrays, four fifths of the scan's cycles at nDims = 10.  Now the snapshot stands in death order: the
        // safe ones are a range, scanned first with a running minimum; the scanners' minima meet; and of the uncertain candidates
        // only those nearer than the PAIR's nearest safe point are inserted -- a handful per baby.
        scan(std::true_type{}, nunc, nsafe, nsafe);
        // the pair's nearest safe point: the scanners' minima meet (equal distances -- exact duplicates only -- go to the lower scanner)
        __syncthreads();                              // (the buffers lie over the tile)
        NND_MARK(2)
        msd[tid * 2 + 0] = sa; msc[tid * 2 + 0] = sca; msd[tid * 2 + 1] = sb; msc[tid * 2 + 1] = scb;
        __syncthreads();
        double fs[2] = {PC_HUGE, PC_HUGE}; int fc[2] = {PC_NN_NONE, PC_NN_NONE};
        if (act) {
            const int t0p = tid - p;                  // the pair's first scanner
#pragma unroll
            for (int h = 0; h < 2; ++h)
                for (int q = 0; q < nscan; ++q) { const double v = msd[(t0p + q) * 2 + h]; if (v < fs[h]) { fs[h] = v; fc[h] = msc[(t0p + q) * 2 + h]; } }
        }
        sa = fs[0]; sb = fs[1];
        scan(std::false_type{}, 0, nunc, nunc + ncand);
        // merge the partial lists of a baby: each is sorted, nscan-way pick by the pair's first scanner, up to the nearest safe point
        __syncthreads();
        NND_MARK(2)
#pragma unroll
        for (int k = 0; k < PC_NN_K; ++k) {
            md[(tid * 2 + 0) * PC_NN_K + k] = bda[k]; mc[(tid * 2 + 0) * PC_NN_K + k] = bca[k];
            md[(tid * 2 + 1) * PC_NN_K + k] = bdb[k]; mc[(tid * 2 + 1) * PC_NN_K + k] = bcb[k];
        }
        __syncthreads();
        if (act && p == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = ia + h;
                if (i >= nr) continue;
                const bool mine = h ? mineb : minea;
                int head[NND_SC];
#pragma unroll
                for (int q = 0; q < NND_SC; ++q) head[q] = 0;
                int out[PC_NN_K];
                bool closed = false;                              // the safe point has been written: nothing behind it matters
#pragma unroll
                for (int k = 0; k < PC_NN_K; ++k) {
                    double best = PC_HUGE; int bq = -1;
#pragma unroll
                    for (int q = 0; q < NND_SC; ++q) {
                        const double v = (q < nscan && head[q] < PC_NN_K) ? md[((tid + q) * 2 + h) * PC_NN_K + head[q]] : PC_HUGE;
                        if (v < best) { best = v; bq = q; }
                    }
                    int code = PC_NN_NONE;
                    if (!closed && best < fs[h]) {
#pragma unroll
                        for (int q = 0; q < NND_SC; ++q) if (q == bq) { code = mc[((tid + q) * 2 + h) * PC_NN_K + head[q]]; head[q]++; }
                    } else if (!closed) { code = fc[h]; closed = true; }
                    out[k] = mine ? code : PC_NN_NONE;
                }
                int4 *dst = (int4 *)(S.nn_list + ((size_t)w * nr + i) * PC_NN_K);
                dst[0] = make_int4(out[0], out[1], out[2], out[3]);
                dst[1] = make_int4(out[4], out[5], out[6], out[7]);
            }
        }
        __syncthreads();
        NND_MARK(3)
    }
#ifdef NND_DBG
    if (w == 0 && tid == 0) {      // (chain 0: the most candidates) set-up / staging / scanning / merges, and the whole kernel in nn_fallbacks
        for (int x = 0; x < 4; ++x) atomicAdd((unsigned long long *)&S.ctl->gen_cyc[x], (unsigned long long)nd_t[x]);
        atomicAdd((unsigned long long *)&S.ctl->nn_fallbacks, (unsigned long long)(clock64() - nd_start));
    }
#endif
}
template <int D> __global__ __launch_bounds__(256) void k_nn_lists_d(PcState S, int nleft, int tile_pts, int use_rank) { nn_lists_d_body<D>(S, nleft, tile_pts, use_rank); }
template <int D> __global__ __launch_bounds__(256) void k_nn_lists_d_many(const PcManyRec *__restrict__ R, int tile_pts, int use_rank)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    if ((int)blockIdx.x >= r.ia[1]) return;
    nn_lists_d_body<D>(r.S, r.ia[1], tile_pts, use_rank);
}
static size_t nn_lists_d_lds(const PcState *S, int &tile)
{
    const int DP = S->D | 1;
    const size_t merge = (sizeof(double) + sizeof(int)) * 256 * 2 * PC_NN_K;      // 48 KB, over the tile: three workgroups to a compute unit
    tile = (int)((merge - 64) / (sizeof(double) * DP + sizeof(int)));
    tile = tile < 32 ? 32 : (tile > 1024 ? 1024 : tile);
    tile &= ~1;                                       // (the codes lie behind tile x DP doubles: eight-byte aligned either way)
    const size_t t = (sizeof(double) * DP + sizeof(int)) * (size_t)tile;
    return (t > merge ? t : merge) + 64;
}
// nDims <= 32: the register kernel; dR == null: one run
static int nn_lists_d_dispatch(const PcState *S, const PcManyRec *dR, int R, int nleft, int use_rank, hipStream_t st)
{
    static const bool off = std::getenv("PC_NN_LISTS_OLD") != nullptr;
    if (off || S->D > 32 || S->D < 1) return 1;
    int tile;
    const size_t sh = nn_lists_d_lds(S, tile);
    if (!S->nn_pts) return 1;
    if (dR) hipLaunchKernelGGL(k_nn_gather_many, dim3(((S->Ncap + nleft) * S->D + 255) / 256, R), dim3(256), 0, st, dR, use_rank);
    else hipLaunchKernelGGL(k_nn_gather, dim3(((S->Ncap + nleft) * S->D + 255) / 256), dim3(256), 0, st, *S, nleft, use_rank);
    switch (S->D) {
#define PC_NND(n) case n: \
        if (dR) { pc_need_dyn_lds((const void *)k_nn_lists_d_many<n>, sh); \
                  hipLaunchKernelGGL(k_nn_lists_d_many<n>, dim3(nleft, R), dim3(256), sh, st, dR, tile, use_rank); } \
        else { pc_need_dyn_lds((const void *)k_nn_lists_d<n>, sh); \
               hipLaunchKernelGGL(k_nn_lists_d<n>, dim3(nleft), dim3(256), sh, st, *S, nleft, tile, use_rank); } \
        return 0;
        PC_NND(1) PC_NND(2) PC_NND(3) PC_NND(4) PC_NND(5) PC_NND(6) PC_NND(7) PC_NND(8) PC_NND(9) PC_NND(10) PC_NND(11) PC_NND(12) PC_NND(13) PC_NND(14) PC_NND(15) PC_NND(16)
        PC_NND(17) PC_NND(18) PC_NND(19) PC_NND(20) PC_NND(21) PC_NND(22) PC_NND(23) PC_NND(24) PC_NND(25) PC_NND(26) PC_NND(27) PC_NND(28) PC_NND(29) PC_NND(30) PC_NND(31) PC_NND(32)
#undef PC_NND
    }
    return 1;
}

static size_t nn_lists_lds(const PcState *S, int &tile)
{
    tile = (int)(24576 / (sizeof(double) * S->D));            // ~24 KB of coordinates per tile
    tile = tile < 16 ? 16 : (tile > 512 ? 512 : tile);
    return sizeof(double) * ((size_t)NNL_G * S->D + (size_t)tile * S->D + 256 * PC_NN_K) + sizeof(int) * (256 * PC_NN_K + tile) + 64;
}
// use_rank: k_sort_live has run on this state in front of this launch (its order tells which candidates cannot die before a chain is looked at)
extern "C" void pc_launch_nn_lists(const PcState *S, int nleft, int use_rank, hipStream_t st)
{
    if (nleft <= 0) return;
    if (nn_lists_d_dispatch(S, nullptr, 0, nleft, use_rank, st) == 0) return;
    int tile;
    const size_t sh = nn_lists_lds(S, tile);
    pc_need_dyn_lds((const void *)k_nn_lists, sh);
    hipLaunchKernelGGL(k_nn_lists, dim3(nleft), dim3(256), sh, st, *S, nleft, tile);
}
extern "C" int pc_launch_nn_lists_many(const PcState *S, const PcManyRec *dR, int R, int nleft_max, int use_rank, hipStream_t st)
{
    if (nleft_max <= 0) return 0;
    if (nn_lists_d_dispatch(S, dR, R, nleft_max, use_rank, st) == 0) return 0;
    int tile;
    const size_t sh = nn_lists_lds(S, tile);
    pc_need_dyn_lds((const void *)k_nn_lists_many, sh);
    hipLaunchKernelGGL(k_nn_lists_many, dim3(nleft_max, R), dim3(256), sh, st, dR, tile);
    return 0;
}

// dynamic nlive target (run_time_info.f90:766-771)
__device__ __forceinline__ int nlive_target(const PcState &S, double logL)
{
    int best = -1;
    for (int i = 0; i < S.n_nlives; ++i)
        if (logL > S.dyn_loglikes[i] && (best < 0 || S.dyn_loglikes[i] > S.dyn_loglikes[best])) best = i;
    return best < 0 ? S.N : S.dyn_nlives[best];
}

template <int NT>
// slots_global: the per-slot arrays (logL, cluster, list position, list owner) stay in HBM instead of LDS -- the way out
// for live sets beyond ~8000 points (20 B of LDS per slot); slower, same code
__global__ __launch_bounds__(NT) void k_consume(PcState S, int final_mode, int cache_x, int xq_n, int slots_global)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Ncap = S.Ncap, maxc = S.maxc, nr = S.nr, nT = S.nT;
    ConsumeShared H;
    {
        char *p = smem;
        const bool sg0 = slots_global != 0;
        H.sL = sg0 ? S.live_logL : (double *)p; if (!sg0) p += sizeof(double) * Ncap;
        double **cd[] = { &H.cLogLp, &H.cLogXp, &H.cLogZp, &H.cLogZXp, &H.cLogZp2, &H.cLogZpXp, &H.cLseRef, &H.cLseSum, &H.cThr };
        for (int i = 0; i < 9; ++i) { *cd[i] = (double *)p; p += sizeof(double) * maxc; }
        H.jobres = (double *)p; p += sizeof(double) * NT;
        H.xbuf = (double *)p; p += sizeof(double) * S.D * PC_IDG;
        H.gd2 = (double *)p; p += sizeof(double) * NT;
        H.sX = (double *)p; H.xrows = cache_x; p += sizeof(double) * (size_t)cache_x * S.D;
        H.red = (vk_t *)p; p += sizeof(vk_t) * 16;
        H.sC = sg0 ? S.live_cluster : (int *)p; if (!sg0) p += sizeof(int) * Ncap;
        H.sP = sg0 ? S.live_pos : (int *)p; if (!sg0) p += sizeof(int) * Ncap;
        H.cN = (int *)p; p += sizeof(int) * maxc;
        H.cMinSlot = (int *)p; p += sizeof(int) * maxc;
        H.cUid = (unsigned *)p; p += sizeof(unsigned) * maxc;
        H.misc = (int *)p; p += sizeof(int) * 8;
        H.gkey = (int *)p; p += sizeof(int) * NT;
        H.ids = (int *)p; p += sizeof(int) * nr;
        H.sO = sg0 ? S.nn_slot_owner : (int *)p; if (!sg0) p += sizeof(int) * Ncap;
        H.sCS = (int *)p; p += sizeof(int) * S.B;
        p = (char *)(((size_t)p + 15) & ~(size_t)15);
        H.xq = (double *)p; H.xq_ld = xq_n;
    }
    PcCtl *ctl = S.ctl;
    // ---- stage the state in LDS
    const bool sg = slots_global != 0;
    if (!sg) for (int s = tid; s < Ncap; s += NT) { H.sL[s] = S.live_logL[s]; H.sC[s] = S.live_cluster[s]; H.sP[s] = S.live_pos[s]; }
    if (H.xrows >= Ncap) for (int e = tid; e < Ncap * S.D; e += NT) H.sX[e] = S.live[(size_t)(e / S.D) * nT + e % S.D];
    const bool nn = S.nn_valid != 0 && !final_mode;
    if (nn) {
        if (!sg) for (int s = tid; s < Ncap; s += NT) H.sO[s] = S.nn_slot_owner[s];
        for (int c = tid; c < S.B; c += NT) H.sCS[c] = S.nn_chain_slot[c];
    }
    int nc = ctl->ncluster;
    // the cross-volume matrix: every death reads and rewrites a row and a column of it; in global memory that is a
    // store -> load round trip per death (and 2 nc^2 of them, serially, when a cluster is deleted)
    const bool xq_lds = nc <= xq_n;                    // clusters only appear between launches
    if (xq_lds) { for (int e = tid; e < nc * nc; e += NT) H.xq[(size_t)(e / nc) * xq_n + e % nc] = S.XpXq[(size_t)(e / nc) * maxc + e % nc]; }
    else { H.xq = S.XpXq; H.xq_ld = maxc; }
    const int nc_at_launch = nc;
    for (int c = tid; c < maxc; c += NT) {
        H.cLogLp[c] = S.logLp[c]; H.cLogXp[c] = S.logXp[c]; H.cLogZp[c] = S.logZp[c]; H.cLogZXp[c] = S.logZXp[c];
        H.cLogZp2[c] = S.logZp2[c]; H.cLogZpXp[c] = S.logZpXp[c]; H.cLseRef[c] = S.lse_ref[c]; H.cLseSum[c] = S.lse_sum[c];
        H.cThr[c] = S.death_thr[c]; H.cN[c] = S.cl_n[c]; H.cMinSlot[c] = S.imin_slot[c]; H.cUid[c] = S.cl_uid[c];
    }
    // replicated scalars (every thread runs the same scalar code on the same inputs)
    int i_nursery = ctl->i_nursery, epoch = ctl->admin_epoch, failures = ctl->failures, ndead = ctl->ndead,
        nph = ctl->nphantom, nc_dead = ctl->ncluster_dead;
    long long nlike = ctl->nlike, niter = ctl->niter, nlike_failed = ctl->nlike_failed;
    double logZ = ctl->logZ, logZ2 = ctl->logZ2, lx_last = ctl->logX_last_update;
    unsigned next_uid = ctl->next_cluster_uid;
    int status = PC_ST_RUNNING, error = PC_ERR_NONE, cluster_deleted = 0;
    const int seg_hi = i_nursery - 1;
    double live_logZ_val = S.logzero, ll_m = -PC_HUGE, ll_s = 0.0;   // termination estimate, as value and as (max, sum) pair
    const double log2v = log(2.0);
    __syncthreads();
    // Reductions over the clusters, one cluster per lane.  Every wave computes them for itself from LDS: no barrier,
    // no broadcast (a serial loop over ~40 clusters with an exp each cost more than the rest of an iteration).
    // log sum_p X_p as a pair (m, s): the value is m + log s.  A fp64 log costs ~2000 cycles of latency on this serial
    // path; the update trigger compares in linear space and the posterior-stack column gets its log on the apply side.
    auto lse_logXp = [&](double &m_out, double &s_out) {
        if (nc == 1) { m_out = H.cLogXp[0]; s_out = 1.0; return; }
        double m = -PC_HUGE;
        for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; m = fmax(m, c < nc ? H.cLogXp[c] : -PC_HUGE); }
        m = wave_max(m);
        double sum = 0.0;
        for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; sum += (c < nc) ? exp(H.cLogXp[c] - m) : 0.0; }
        m_out = m; s_out = wave_sum<4>(sum);
    };
    auto lowest_contour = [&]() -> vk_t {              // (min_p logL_p, first cluster that has it): minpos
        if (nc == 1) return vk_t{H.cLogLp[0], 0};
        vk_t b{PC_HUGE, 0x7fffffff};
        for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; if (c < nc) b = vk_min(b, vk_t{H.cLogLp[c], c}); }
        return wave_argmin(b);
    };

    // ================================================================================
    // one death: delete_outermost_point (run_time_info.f90:789-817) without the row copy
    // ================================================================================
    int last_cd = -1, last_pos_del = -1;
    double lx_m = 0.0, lx_s = 1.0; bool lx_known = false;   // log sum_p X_p = lx_m + log lx_s: only a death changes it
    int n_total = 0;                                   // live points over all clusters
    for (int c = 0; c < nc; ++c) n_total += H.cN[c];
    // Several waves, live set of a few thousand points: the three expensive parts of a death -- the evidence jobs (an exp and a
    // log each), the scan of the slots for the cluster's next minimum, the log-sum-exp of the volumes -- read only the state
    // before the death and do not need each other: the last wave takes the scan and the volumes while the others do the jobs,
    // and the state is rewritten once, behind one barrier (two barriers per death instead of seven; measured before: jobs
    // 2.5 k cycles, scan 2.3 k, volumes 0.8 k, one after the other).
    const bool fastkill = NT >= 256 && Ncap <= 4096 && !slots_global;
    const int ndead_kill0 = ndead;                            // (the kill-off's first death: its rows are copied behind the loop)
    auto kill_lowest = [&](int plan_w) {
        // cluster with the lowest contour (minpos: first minimum)
        const int cd = lowest_contour().k;
        const int n = H.cN[cd];
        n_total--;
        const double L = H.cLogLp[cd];
        const int slot_del = H.cMinSlot[cd];
        const int pos_del = H.sP[slot_del];
        last_cd = cd; last_pos_del = pos_del;
        // ---- update_evidence (run_time_info.f90:211-296): every log-space accumulation reads only
        //      pre-update values, so they are independent jobs: one lane each.
        const double l0 = S.logn[n], l1 = S.logn[n + 1], l2 = S.logn[n + 2];
        const double Xp = H.cLogXp[cd], XX = H.xq[(size_t)cd * H.xq_ld + cd];
        const double logweight = Xp - l1;
        if (fastkill && nc <= NT - 64 - 8) {
            constexpr int NJ = NT - 64;                       // job lanes; the last wave scans
            int new_min_slot = -1; double new_min_L = PC_HUGE;
            if (tid >= NJ) {
                // ---- delete_point (array_utils.f90:433-458) + find_min_loglikelihoods (run_time_info.f90:883-909): the dying slot
                //      is skipped (its label is cleared below, with the rest of the state), the last list element takes its place
                vk_t best{PC_HUGE, 0x7fffffff};
                int myslot = -1;
                for (int s0 = lane; s0 < Ncap; s0 += 256) {             // four labels in flight: members are rare (n_p of Ncap slots)
                    int lab[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) lab[u] = (s0 + 64 * u < Ncap) ? H.sC[s0 + 64 * u] : -1;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = s0 + 64 * u;
                        if (lab[u] != cd || s == slot_del) continue;
                        int p = H.sP[s];
                        if (p == n - 1) { p = pos_del; H.sP[s] = p; }
                        const vk_t cand{H.sL[s], p};
                        const vk_t nb = vk_min(best, cand);
                        if (nb.k != best.k || nb.v != best.v) myslot = s;
                        best = nb;
                    }
                }
                const vk_t mine = best;
                best = wave_argmin(best);
                if (n - 1 > 0) { if (myslot >= 0 && mine.k == best.k && mine.v == best.v) { H.misc[0] = myslot; H.jobres[NT - 3] = best.v; } }
                else if (lane == 0) { H.misc[0] = -1; H.jobres[NT - 3] = PC_HUGE; }
            }
            // log sum_p X_p after the death (the dying cluster's new volume substituted): the scanning wave's job, or that of the
            // last job wave when its lanes have nothing else to do (100 clusters: the scan alone is the long pole)
            const bool lse_sep = 8 + nc <= NJ - 64;
            if (lse_sep ? (tid >= NJ - 64 && tid < NJ) : (tid >= NJ)) {
                double m = -PC_HUGE;
                for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; m = fmax(m, c < nc ? (c == cd ? Xp + l0 - l1 : H.cLogXp[c]) : -PC_HUGE); }
                m = wave_max(m);
                double sum = 0.0;
                for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; sum += (c < nc) ? exp((c == cd ? Xp + l0 - l1 : H.cLogXp[c]) - m) : 0.0; }
                sum = wave_sum<4>(sum);
                if (nc == 1) { m = Xp + l0 - l1; sum = 1.0; }
                if (lane == 0) { H.jobres[NT - 1] = m; H.jobres[NT - 2] = sum; }
            }
            if (tid < (lse_sep ? NJ - 64 : NJ)) {
                double a = 0.0, b = 0.0, c3 = 0.0; bool has = false, has3 = false;
                if (tid == 0) { a = logZ; b = Xp + L - l1; has = true; }
                else if (tid == 1) { a = H.cLogZp[cd]; b = Xp + L - l1; has = true; }
                else if (tid == 2) { a = logZ2; b = log2v + H.cLogZXp[cd] + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; has = has3 = true; }
                else if (tid == 3) { a = H.cLogZXp[cd] + l0 - l1; b = XX + L + l0 - l1 - l2; has = true; }
                else if (tid == 4) { a = H.cLogZp2[cd]; b = log2v + H.cLogZpXp[cd] + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; has = has3 = true; }
                else if (tid == 5) { a = H.cLogZpXp[cd] + l0 - l1; b = XX + L + l0 - l1 - l2; has = true; }
                else if (tid == 6) { a = exp(L - H.cLseRef[cd]); }                       // live logsumexp bookkeeping
                else if (tid >= 8 && tid - 8 < nc && tid - 8 != cd && tid < NJ) {
                    const int q = tid - 8;
                    a = H.cLogZXp[q]; b = H.xq[(size_t)cd * H.xq_ld + q] + L - l1; has = true;
                }
                double r = a;
                if (has3) { const double m3 = fmax(a, fmax(b, c3)); r = m3 + log(exp(a - m3) + exp(b - m3) + exp(c3 - m3)); }
                else if (has) r = pc_logaddexp(a, b);
                H.jobres[tid] = r;
            }
            __syncthreads();
            logZ = H.jobres[0]; logZ2 = H.jobres[2];
            const double lxm = H.jobres[NT - 1], lxs = H.jobres[NT - 2];
            new_min_slot = H.misc[0]; new_min_L = H.jobres[NT - 3];
            if (tid == 0) {
                H.cLogZp[cd] = H.jobres[1]; H.cLogZXp[cd] = H.jobres[3]; H.cLogZp2[cd] = H.jobres[4]; H.cLogZpXp[cd] = H.jobres[5];
                H.cLogXp[cd] = Xp + l0 - l1;
                H.cLseSum[cd] -= H.jobres[6];
                H.cThr[cd] = L;
                H.cN[cd] = n - 1;
                H.sC[slot_del] = -1;
                if (nn) H.sO[slot_del] = -2;
                H.cMinSlot[cd] = new_min_slot; H.cLogLp[cd] = new_min_L;
            }
            if (tid < nc && tid != cd) H.cLogZXp[tid] = H.jobres[8 + tid];
            for (int q = tid; q < nc; q += NT) {
                if (q == cd) H.xq[(size_t)cd * H.xq_ld + cd] = XX + l0 - l2;
                else {
                    const double v = H.xq[(size_t)cd * H.xq_ld + q] + l0 - l1;
                    H.xq[(size_t)cd * H.xq_ld + q] = v; H.xq[(size_t)q * H.xq_ld + cd] = v;
                }
            }
            __syncthreads();                                  // (jobres is rewritten by the next death only after this)
            lx_m = lxm; lx_s = lxs; lx_known = true;
            if (ndead >= S.Dcap) { error = PC_ERR_DEAD_CAP; status = PC_ST_ERROR; }
            if (status != PC_ST_ERROR) {
                if (plan_w >= 0) {
                    if (tid == 0) {
                        const int src = S.slot_src[slot_del];
                        S.plan[plan_w].dead_idx = ndead;
                        S.plan[plan_w].dead_src = (src >= 0) ? -(1 + src) : slot_del;
                        S.plan[plan_w].logw = logweight; S.plan[plan_w].postX = lxm; S.plan[plan_w].postXs = lxs; S.plan[plan_w].postZ = logZ;
                        S.plan[plan_w].dead_cuid = H.cUid[cd];
                    }
                } else if (final_mode == 1) {
                    // the kill-off: nothing is born any more, the rows stay what they are -- they leave together behind the loop (a copy
                    // here is a global load and a store per death on the loop's critical path: a third of its 4.5 us)
                    if (tid == 0) {
                        S.sort_slot[ndead - ndead_kill0] = slot_del;
                        S.dead_logw[ndead] = logweight; S.dead_postX[ndead] = lxm + log(lxs); S.dead_postZ[ndead] = logZ;
                        S.dead_cuid[ndead] = H.cUid[cd];
                    }
                } else {   // trimming: rows are current in live[], copy immediately
                    const double *row = S.live + (size_t)slot_del * nT;
                    double *dst = S.dead + (size_t)ndead * nT;
                    for (int e = tid; e < nT; e += NT) dst[e] = row[e];
                    if (tid == 0) {
                        S.dead_logw[ndead] = logweight; S.dead_postX[ndead] = lxm + log(lxs); S.dead_postZ[ndead] = logZ;
                        S.dead_cuid[ndead] = H.cUid[cd]; S.dead_entry[ndead] = S.live_entry[slot_del];
                    }
                }
            }
            ndead++;
            return slot_del;
        }
        {
            double a = 0.0, b = 0.0, c3 = 0.0; bool has = false, has3 = false;
            if (tid == 0) { a = logZ; b = Xp + L - l1; has = true; }
            else if (tid == 1) { a = H.cLogZp[cd]; b = Xp + L - l1; has = true; }
            else if (tid == 2) { a = logZ2; b = log2v + H.cLogZXp[cd] + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; has = has3 = true; }
            else if (tid == 3) { a = H.cLogZXp[cd] + l0 - l1; b = XX + L + l0 - l1 - l2; has = true; }
            else if (tid == 4) { a = H.cLogZp2[cd]; b = log2v + H.cLogZpXp[cd] + L - l1; c3 = log2v + XX + 2 * L - l1 - l2; has = has3 = true; }
            else if (tid == 5) { a = H.cLogZpXp[cd] + l0 - l1; b = XX + L + l0 - l1 - l2; has = true; }
            else if (tid == 6) { a = exp(L - H.cLseRef[cd]); }                       // live logsumexp bookkeeping
            else if (tid >= 8 && tid - 8 < nc && tid - 8 != cd && tid - 8 < NT - 8) {
                const int q = tid - 8;
                a = H.cLogZXp[q]; b = H.xq[(size_t)cd * H.xq_ld + q] + L - l1; has = true;
            }
            double r = a;
            if (has3) {                                 // one log instead of two nested logaddexp's
                const double m3 = fmax(a, fmax(b, c3));
                r = m3 + log(exp(a - m3) + exp(b - m3) + exp(c3 - m3));
            } else if (has) r = pc_logaddexp(a, b);
            H.jobres[tid] = r;
        }
        __syncthreads();
        // clusters beyond the lane budget (nc > NT-8): serial tail, rare
        for (int q = NT - 8 + tid; q < nc; q += NT)
            if (q != cd) H.cLogZXp[q] = pc_logaddexp(H.cLogZXp[q], H.xq[(size_t)cd * H.xq_ld + q] + L - l1);
        logZ = H.jobres[0]; logZ2 = H.jobres[2];
        const double nZp = H.jobres[1], nZXp = H.jobres[3], nZp2 = H.jobres[4], nZpXp = H.jobres[5], edel = H.jobres[6];
        __syncthreads();
        if (tid == 0) {
            H.cLogZp[cd] = nZp; H.cLogZXp[cd] = nZXp; H.cLogZp2[cd] = nZp2; H.cLogZpXp[cd] = nZpXp;
            H.cLogXp[cd] = Xp + l0 - l1;
            H.cLseSum[cd] -= edel;
            H.cThr[cd] = L;
            H.cN[cd] = n - 1;
            H.sC[slot_del] = -1;
            if (nn) H.sO[slot_del] = -2;
        }
        for (int q = tid; q < nc; q += NT) {
            if (q == cd) { if (true) H.xq[(size_t)cd * H.xq_ld + cd] = XX + l0 - l2; }
            else {
                if (q < NT - 8) H.cLogZXp[q] = H.jobres[8 + q];
                const double v = H.xq[(size_t)cd * H.xq_ld + q] + l0 - l1;
                H.xq[(size_t)cd * H.xq_ld + q] = v; H.xq[(size_t)q * H.xq_ld + cd] = v;
            }
        }
        __syncthreads();
        // ---- delete_point (array_utils.f90:433-458): the last list element moves into the hole;
        //      find_min_loglikelihoods (run_time_info.f90:883-909) for the shrunk cluster, one pass.
        vk_t best{PC_HUGE, 0x7fffffff};
        int myslot = -1;                               // slot of this thread's own candidate
        for (int s = tid; s < Ncap; s += NT) {
            if (H.sC[s] != cd) continue;
            int p = H.sP[s];
            if (p == n - 1) { p = pos_del; H.sP[s] = p; }
            const vk_t cand{H.sL[s], p};
            const vk_t nb = vk_min(best, cand);
            if (nb.k != best.k || nb.v != best.v) myslot = s;
            best = nb;
        }
        const vk_t mine = best;
        best = block_argmin<NT>(best, H.red);
        // slot of the new minimum: the thread whose own candidate (logL, pos) is the winner knows it (positions are unique)
        if (n - 1 > 0) {
            if (myslot >= 0 && mine.k == best.k && mine.v == best.v) { H.cMinSlot[cd] = myslot; H.cLogLp[cd] = best.v; }
        } else if (tid == 0) { H.cMinSlot[cd] = -1; H.cLogLp[cd] = PC_HUGE; }
        __syncthreads();
        // posterior-stack columns (calculate.f90:53-79): volume after the update, logZ after the update
        double lxm, lxs;
        lse_logXp(lxm, lxs);
        lx_m = lxm; lx_s = lxs; lx_known = true;
        if (ndead >= S.Dcap) { error = PC_ERR_DEAD_CAP; status = PC_ST_ERROR; }
        if (status != PC_ST_ERROR) {
            if (plan_w >= 0) {
                if (tid == 0) {
                    const int src = S.slot_src[slot_del];
                    S.plan[plan_w].dead_idx = ndead;
                    S.plan[plan_w].dead_src = (src >= 0) ? -(1 + src) : slot_del;
                    S.plan[plan_w].logw = logweight; S.plan[plan_w].postX = lxm; S.plan[plan_w].postXs = lxs; S.plan[plan_w].postZ = logZ;
                    S.plan[plan_w].dead_cuid = H.cUid[cd];
                }
            } else if (final_mode == 1) {      // (the kill-off: the rows leave together behind the loop, see above)
                if (tid == 0) {
                    S.sort_slot[ndead - ndead_kill0] = slot_del;
                    S.dead_logw[ndead] = logweight; S.dead_postX[ndead] = lxm + log(lxs); S.dead_postZ[ndead] = logZ;
                    S.dead_cuid[ndead] = H.cUid[cd];
                }
            } else {   // trimming: rows are current in live[], copy immediately
                const double *row = S.live + (size_t)slot_del * nT;
                double *dst = S.dead + (size_t)ndead * nT;
                for (int e = tid; e < nT; e += NT) dst[e] = row[e];
                if (tid == 0) {
                    S.dead_logw[ndead] = logweight; S.dead_postX[ndead] = lxm + log(lxs); S.dead_postZ[ndead] = logZ;
                    S.dead_cuid[ndead] = H.cUid[cd]; S.dead_entry[ndead] = S.live_entry[slot_del];
                }
            }
        }
        ndead++;
        return slot_del;
    };

    // delete_cluster (run_time_info.f90:507-598): drop the first empty cluster, keep the others' order
    int dropped_p = -1;
    auto drop_empty_cluster = [&]() -> bool {
        int p = -1;                                    // first empty cluster: one cluster per lane, every wave for itself
        for (int c0 = 0; c0 < nc && p < 0; c0 += 64) {
            const unsigned long long m = __ballot(c0 + lane < nc && H.cN[c0 + lane] == 0);
            if (m) p = c0 + __ffsll((long long)m) - 1;
        }
        if (p < 0) return false;
        __syncthreads();
        if (tid == 0) {
            if (nc_dead < S.maxc_dead) {
                S.logZp_dead[nc_dead] = H.cLogZp[p]; S.logZp2_dead[nc_dead] = H.cLogZp2[p]; S.cl_uid_dead[nc_dead] = H.cUid[p];
            }
        }
        nc_dead++;
        // XpXq compaction (row-major in place is safe: targets are lexicographically <= sources)
        if (tid == 0) {
            for (int a = 0, na = 0; a < nc; ++a) {
                if (a == p) continue;
                for (int b = 0, nb = 0; b < nc; ++b) { if (b == p) continue; H.xq[(size_t)na * H.xq_ld + nb] = H.xq[(size_t)a * H.xq_ld + b]; nb++; }
                na++;
            }
            for (int c = p; c < nc - 1; ++c) {
                H.cLogLp[c] = H.cLogLp[c + 1]; H.cLogXp[c] = H.cLogXp[c + 1]; H.cLogZp[c] = H.cLogZp[c + 1];
                H.cLogZXp[c] = H.cLogZXp[c + 1]; H.cLogZp2[c] = H.cLogZp2[c + 1]; H.cLogZpXp[c] = H.cLogZpXp[c + 1];
                H.cLseRef[c] = H.cLseRef[c + 1]; H.cLseSum[c] = H.cLseSum[c + 1]; H.cThr[c] = H.cThr[c + 1];
                H.cN[c] = H.cN[c + 1]; H.cMinSlot[c] = H.cMinSlot[c + 1]; H.cUid[c] = H.cUid[c + 1];
            }
        }
        const int DD = S.D * S.D;
        for (int c = p; c < nc - 1; ++c) {
            for (int e = tid; e < DD; e += NT) {
                S.chol[(size_t)c * DD + e] = S.chol[(size_t)(c + 1) * DD + e];
                S.cov[(size_t)c * DD + e] = S.cov[(size_t)(c + 1) * DD + e];
            }
            __syncthreads();
        }
        for (int s = tid; s < Ncap; s += NT) if (H.sC[s] > p) H.sC[s] -= 1;
        nc--;
        cluster_deleted = 1;
        dropped_p = p;
        __syncthreads();
        return true;
    };

    if (final_mode == 1) {
        // nested_sampling.F90:381-384: kill every remaining live point, lowest first
        while (nc > 0 && status == PC_ST_RUNNING) {
            kill_lowest(-1);
            drop_empty_cluster();
        }
        {   // the rows of the points that died here (the k-th death of the loop took slot sort_slot[k]): all threads, eight loads in flight
            __threadfence_block();
            __syncthreads();
            const int nk = ndead - ndead_kill0;
            const long long ne = (long long)nk * nT;
            for (long long e0 = tid; e0 < ne; e0 += (long long)NT * 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long long e = e0 + (long long)u * NT;
                    if (e < ne) { const int k = (int)(e / nT), d = (int)(e - (long long)k * nT); v[u] = S.live[(size_t)((volatile int *)S.sort_slot)[k] * nT + d]; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const long long e = e0 + (long long)u * NT; if (e < ne) S.dead[(size_t)ndead_kill0 * nT + e] = v[u]; }
            }
            for (int k = tid; k < nk; k += NT) S.dead_entry[ndead_kill0 + k] = S.live_entry[((volatile int *)S.sort_slot)[k]];
        }
        status = PC_ST_DONE;
    } else if (final_mode == 2) {
        // nested_sampling.F90:201-205: nprior > nlive, trim the initial set down to nlive
        while (H.cN[0] > S.N && status == PC_ST_RUNNING) kill_lowest(-1);
    }

    // ================================================================================
    // main loop: nested_sampling.F90:239-374 for the entries left in the nursery
    // ================================================================================
    long long cyT = 0, cyI = 0, cyK = 0, cyE = 0, cyF = 0, cyW = 0;
    bool live_changed = true;
    if (!final_mode && nc > 1) {
        // The nursery's records were written by other XCDs: a first touch costs 1-2 us, and the loop below would pay
        // that once per chain and array, serially.  Touch everything it will read now, in bulk, so that the loop's
        // loads hit this XCD's L2 (a few hundred KB at most).
        auto touch = [&](const void *base, size_t bytes) {
            const char *b = (const char *)base;
            for (size_t o = (size_t)tid * 64; o < bytes; o += (size_t)NT * 64) { const int v = *(const volatile int *)(b + o); asm volatile("" :: "v"(v)); }
        };
        touch(S.baby_logL, sizeof(double) * (size_t)i_nursery * nr);
        touch(S.ch_nlike, sizeof(int) * (size_t)i_nursery); touch(S.ch_epoch, sizeof(int) * (size_t)i_nursery); touch(S.ch_cluster, sizeof(int) * (size_t)i_nursery);
        if (nn) touch(S.nn_list, sizeof(int) * (size_t)i_nursery * nr * PC_NN_K);
        touch(S.slot_src, sizeof(int) * (size_t)Ncap);
        if (!xq_lds) for (int c = 0; c < nc; ++c) touch(S.XpXq + (size_t)c * maxc, sizeof(double) * (size_t)nc);
        // cube coordinates of every chain's last baby (they enter the LDS copy of the live set when the chain is accepted)
        for (int w2 = tid; w2 < i_nursery; w2 += NT) {
            const char *b = (const char *)(S.babies + ((size_t)w2 * nr + (nr - 1)) * nT);
            for (size_t o = 0; o < sizeof(double) * (size_t)S.D; o += 64) { const int v = *(const volatile int *)(b + o); asm volatile("" :: "v"(v)); }
        }
    }
    // The records of a chain -- its counters, the logL of its babies, their candidate lists -- are in L2 at best: three
    // dependent round trips per chain on this serial path.  They do not change during the launch: the next chain's are
    // requested while this one is processed.
    int pf_nlike = 0, pf_epoch = 0, pf_ca = 0;
    double pf_blog = 0.0, pf_last = 0.0;
    int4 pf_a = make_int4(PC_NN_NONE, PC_NN_NONE, PC_NN_NONE, PC_NN_NONE), pf_b = pf_a;
    auto prefetch_chain = [&](int wn) {
        if (wn < 0) return;
        pf_nlike = S.ch_nlike[wn]; pf_epoch = S.ch_epoch[wn]; pf_ca = S.ch_cluster[wn];
        const double *bl = S.baby_logL + (size_t)wn * nr;
        pf_blog = tid < nr ? bl[tid] : 0.0; pf_last = bl[nr - 1];
        if (nn && tid < nr) { const int4 *L4 = (const int4 *)(S.nn_list + ((size_t)wn * nr + tid) * PC_NN_K); pf_a = L4[0]; pf_b = L4[1]; }
    };
    if (!final_mode) prefetch_chain(i_nursery - 1);
    while (!final_mode && status == PC_ST_RUNNING) {
        const long long q0 = clock64();
        // ---- more_samples_needed (nested_sampling.F90:514-543) + failures guard (:239)
        bool more = true;
        if (S.max_ndead == 0) more = false;
        else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
        else if (S.use_prec) {
            // live_logZ (run_time_info.f90:683-709); per-cluster logsumexp kept incrementally.
            // One cluster per lane, log-sum-exp over the wave; the other waves pick the result up from LDS.
            // (a chain that replaced nothing left live set and evidence as they were: the estimate stands)
            if (live_changed)
            {   // every wave for itself (the state it reads was published before the last barrier of the iteration).
                // live_logZ = log sum_p exp(lse_p - log n_p + logX_p) with lse_p = ref_p + log(sum_p): kept as a pair
                // (mxv, acc), acc = sum_p (sum_p / n_p) exp(ref_p + logX_p - mxv) -- no log on the critical path
                double mxv = -PC_HUGE;
                for (int c0 = 0; c0 < nc; c0 += 64) {
                    const int c = c0 + lane;
                    mxv = fmax(mxv, (c < nc && H.cN[c] > 0) ? H.cLseRef[c] + H.cLogXp[c] : -PC_HUGE);
                }
                mxv = wave_max(mxv);
                double acc = 0.0;
                for (int c0 = 0; c0 < nc; c0 += 64) {
                    const int c = c0 + lane;
                    if (c < nc && H.cN[c] > 0) acc += (H.cLseSum[c] / ((double)H.cN[c] + 0.0)) * exp(H.cLseRef[c] + H.cLogXp[c] - mxv);
                }
                acc = wave_sum<4>(acc);
                ll_m = mxv; ll_s = acc;
            }
            // more_samples_needed: live_logZ < log(precision) + logZ   <=>   acc < exp(log(precision) + logZ - mxv)
            if (!(ll_s > 0.0) || ll_s < exp(S.log_prec + logZ - ll_m)) more = false;
        }
        if (!more || failures > S.nfail) { status = PC_ST_DONE; break; }
        if (i_nursery == 0) break;                      // batch exhausted: host launches the next one

        const long long q1 = clock64(); cyT += q1 - q0;
        const int w = i_nursery - 1;
        i_nursery--;
        const int n_total_before = n_total;
        live_changed = false;
        const int w_nlike = pf_nlike, w_epoch = pf_epoch, ca = pf_ca;
        const double my_blog = pf_blog, Llast_pf = pf_last;
        const int4 my_a = pf_a, my_b = pf_b;
        prefetch_chain(i_nursery - 1);
        nlike += w_nlike;
        niter++;
        if (tid == 0) { S.plan[w].dead_idx = -1; S.plan[w].ph_base = nph; S.plan[w].ph_count = 0; S.plan[w].contour = S.logzero; for (int m = 0; m < PC_MASK_WORDS; ++m) S.plan[w].ph_mask[m] = 0ull; }
        __syncthreads();
        if (w_epoch != epoch) { nlike_failed += w_nlike; continue; }   // nested_sampling.F90:313 epoch guard

        // ---- replace_point (run_time_info.f90:716-787)
        const double Lg = lowest_contour().v;
        const double *blog = S.baby_logL + (size_t)w * nr;
        if (tid == 0) S.plan[w].contour = Lg;
        // phantoms: babies 1..nr-1 that beat the global contour and fall in the seed cluster's cell
        int nph_add = 0;
        if (nc == 1) {
            for (int base = 0; base < nr - 1; base += NT) {
                const int i = base + tid;
                const bool f = (i < nr - 1) && ((base == 0 ? my_blog : blog[i]) > Lg);
                const unsigned long long m = __ballot(f);
                if (lane == 0 && m) S.plan[w].ph_mask[(base >> 6) + (tid >> 6)] = m;
                if (NT == 64) nph_add += __popcll(m);
            }
            if (NT > 64) {   // count after the masks are visible
                __syncthreads();
                for (int m = 0; m < (nr + 62) / 64; ++m) nph_add += __popcll(S.plan[w].ph_mask[m]);
            }
        } else {
            if (nn) {
                // identify_cluster from the candidate lists: the first entry that is alive NOW is the nearest live
                // point (everything nearer in the superset is dead or not yet born); one baby per thread
                int unresolved = 0;
                for (int i = tid; i < nr; i += NT) {
                    int res = -1;
                    if ((i == tid ? my_blog : blog[i]) > Lg) {
                        res = -2;
                        const int4 *L4 = (const int4 *)(S.nn_list + ((size_t)w * nr + i) * PC_NN_K);
                        const int4 a = (i == tid) ? my_a : L4[0], b = (i == tid) ? my_b : L4[1];
                        const int codes[PC_NN_K] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
                        for (int k = 0; k < PC_NN_K; ++k) {
                            const int code = codes[k];
                            if (res != -2 || code == PC_NN_NONE) continue;
                            const int s = code >= 0 ? code : H.sCS[-(1 + code)];
                            const bool alive = code >= 0 ? (H.sO[s] == -1) : (s >= 0 && H.sO[s] == -(1 + code));
                            if (alive) res = H.sC[s];
                        }
                        if (res == -2) unresolved = 1;
                    }
                    H.ids[i] = res;
                }
                cyW++;
                if (__syncthreads_or(unresolved)) { cyF++; block_identify_all<NT>(S, H, w, blog, Lg, nc, true); }
            } else block_identify_all<NT>(S, H, w, blog, Lg, nc);
            if (NT > 64 && nr <= NT) {
                // the masks of the 16 waves meet in LDS (scratch of the search); a store to the plan followed by a
                // load of the same words would cost a global round trip per chain
                const bool f = (tid < nr - 1) && (my_blog > Lg) && (H.ids[tid] == ca);
                const unsigned long long m = __ballot(f);
                if (lane == 0) { H.gkey[2 * (tid >> 6)] = (int)(unsigned)m; H.gkey[2 * (tid >> 6) + 1] = (int)(unsigned)(m >> 32); }
                __syncthreads();
                for (int q = 0; q < (nr + 62) / 64; ++q) nph_add += __popc((unsigned)H.gkey[2 * q]) + __popc((unsigned)H.gkey[2 * q + 1]);
                if (tid < PC_MASK_WORDS && tid < (nr + 62) / 64) {
                    const unsigned long long mm = ((unsigned long long)(unsigned)H.gkey[2 * tid + 1] << 32) | (unsigned)H.gkey[2 * tid];
                    if (mm) S.plan[w].ph_mask[tid] = mm;
                }
                __syncthreads();                       // gkey is reused by the next search
            } else {
                for (int base = 0; base < nr - 1; base += NT) {
                    const int i = base + tid;
                    const bool f = (i < nr - 1) && (blog[i] > Lg) && (H.ids[i] == ca);
                    const unsigned long long m = __ballot(f);
                    if (lane == 0 && m) S.plan[w].ph_mask[(base >> 6) + (tid >> 6)] = m;
                    if (NT == 64) nph_add += __popcll(m);
                }
                if (NT > 64) {   // count after the masks are visible
                    __syncthreads();
                    for (int m = 0; m < (nr + 62) / 64; ++m) nph_add += __popcll(S.plan[w].ph_mask[m]);
                }
            }
        }
        const long long q2 = clock64(); cyI += q2 - q1;
        if (nph + nph_add > S.Pcap) { status = PC_ST_ERROR; error = PC_ERR_PHANTOM_CAP; break; }
        if (tid == 0) S.plan[w].ph_cuid = H.cUid[ca];
        nph += nph_add;

        const double Llast = Llast_pf;
        bool replaced = false;
        if (Llast > Lg) {
            const int id = (nc == 1) ? 0 : H.ids[nr - 1];
            if (id == ca) {
                const int nl = nlive_target(S, Lg);
                int tot = n_total;
                int free_slot = -1;
                if (tot >= (nl > 1 ? nl : 1)) { free_slot = kill_lowest(w); replaced = true; tot--; }
                if (status == PC_ST_ERROR) break;
                if (tot < nl) {
                    if (free_slot < 0) {               // growing live set: take any free slot
                        if (tid == 0) H.misc[1] = -1;
                        __syncthreads();
                        for (int s = tid; s < Ncap; s += NT) if (H.sC[s] < 0) atomicMax(&H.misc[1], s);
                        __syncthreads();
                        free_slot = H.misc[1];
                        if (free_slot < 0) { status = PC_ST_ERROR; error = PC_ERR_NOSLOT; break; }
                    }
                    // add_point + find_min_loglikelihoods for the receiving cluster
                    if (tid == 0) {
                        const int pos = H.cN[ca];
                        H.sL[free_slot] = Llast; H.sC[free_slot] = ca; H.sP[free_slot] = pos;
                        H.cN[ca] = pos + 1;
                        if (pos == 0 || Llast < H.cLogLp[ca]) { H.cLogLp[ca] = Llast; H.cMinSlot[ca] = free_slot; }
                        // live logsumexp of the cluster
                        if (pos == 0) { H.cLseRef[ca] = Llast; H.cLseSum[ca] = 1.0; }
                        else if (Llast > H.cLseRef[ca]) { H.cLseSum[ca] = H.cLseSum[ca] * exp(H.cLseRef[ca] - Llast) + 1.0; H.cLseRef[ca] = Llast; }
                        else H.cLseSum[ca] += exp(Llast - H.cLseRef[ca]);
                        S.slot_src[free_slot] = w;
                        if (nn) { H.sO[free_slot] = w; H.sCS[w] = free_slot; }
                    }
                    n_total++;
                    if (H.xrows >= Ncap) for (int d = tid; d < S.D; d += NT) H.sX[(size_t)free_slot * S.D + d] = S.babies[((size_t)w * nr + nr - 1) * nT + d];
                    __syncthreads();
                    // engine rule (oracle keyed mode): a point that replaces a death of its own cluster
                    // takes the dead point's list position instead of being appended
                    if (!S.seq_mode && replaced && last_cd == ca && last_pos_del < H.cN[ca] - 1) {
                        const int pos_new = H.cN[ca] - 1;
                        for (int s = tid; s < Ncap; s += NT)
                            if (s != free_slot && H.sC[s] == ca && H.sP[s] == last_pos_del) H.sP[s] = pos_new;
                        __syncthreads();
                        if (tid == 0) H.sP[free_slot] = last_pos_del;
                        __syncthreads();
                    }
                }
            }
        } else {
            // failed spawn: the last baby is recorded as dead with zero weight (run_time_info.f90:781-785)
            if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
            if (tid == 0) {
                S.plan[w].dead_idx = ndead; S.plan[w].dead_src = -(1 + w);
                S.plan[w].logw = S.logzero; S.plan[w].postX = 0.0; S.plan[w].postXs = 1.0; S.plan[w].postZ = 0.0; S.plan[w].dead_cuid = 0xFFFFFFFFu;
            }
            ndead++;
        }
        failures = replaced ? 0 : failures + 1;
        if (!replaced) nlike_failed += w_nlike;
        live_changed = replaced || n_total != n_total_before;
        const long long q3 = clock64(); cyK += q3 - q2;

        // ---- update trigger (nested_sampling.F90:321) and delete_cluster (:339)
        if (!lx_known) { lse_logXp(lx_m, lx_s); lx_known = true; }
        // nested_sampling.F90:321  logsumexp(logXp) <= logX_last_update + log(compression_factor), in linear space
        const bool update = lx_s <= exp(lx_last + S.log_cf - lx_m);
        if (update) lx_last = lx_m + log(lx_s);
        if (drop_empty_cluster()) {
            if (S.epoch_discard) epoch++;          // nested_sampling.F90:339-341 as written: every chain in flight fails the guard of :313
            else {
                // the engine's rule (oracle: remap_chains): only the chains seeded in the cluster that ended are lost; the clusters behind
                // it moved up one place and the chains seeded in them follow
                const int p = dropped_p;
                for (int w2 = tid; w2 < i_nursery; w2 += NT) {
                    const int c = S.ch_cluster[w2];
                    if (c == p) { S.ch_cluster[w2] = -1; S.ch_epoch[w2] = -1; }
                    else if (c > p) S.ch_cluster[w2] = c - 1;
                }
                if (pf_ca == p) { pf_ca = -1; pf_epoch = -1; } else if (pf_ca > p) pf_ca -= 1;      // (the next chain's record is in registers already)
                __syncthreads();
            }
        }
        cyE += clock64() - q3;
        if (nc == 0) { status = PC_ST_DONE; break; }
        if (update) { status = PC_ST_UPDATE; break; }
    }

    // ---- write the state back
    __syncthreads();
    if (!sg) for (int s = tid; s < Ncap; s += NT) { S.live_logL[s] = H.sL[s]; S.live_cluster[s] = H.sC[s]; S.live_pos[s] = H.sP[s]; }
    for (int s = tid; s < Ncap; s += NT) if (H.sC[s] >= 0) S.cl_list[(size_t)H.sC[s] * Ncap + H.sP[s]] = s;
    if (xq_lds) {                                      // (a deleted cluster leaves stale entries beyond nc: never read)
        for (int e = tid; e < nc_at_launch * nc_at_launch; e += NT) S.XpXq[(size_t)(e / nc_at_launch) * maxc + e % nc_at_launch] = H.xq[(size_t)(e / nc_at_launch) * xq_n + e % nc_at_launch];
    }
    if (nn) {
        if (!sg) for (int s = tid; s < Ncap; s += NT) S.nn_slot_owner[s] = H.sO[s];
        for (int c = tid; c < S.B; c += NT) S.nn_chain_slot[c] = H.sCS[c];
    }
    for (int c = tid; c < maxc; c += NT) {
        S.logLp[c] = H.cLogLp[c]; S.logXp[c] = H.cLogXp[c]; S.logZp[c] = H.cLogZp[c]; S.logZXp[c] = H.cLogZXp[c];
        S.logZp2[c] = H.cLogZp2[c]; S.logZpXp[c] = H.cLogZpXp[c]; S.lse_ref[c] = H.cLseRef[c]; S.lse_sum[c] = H.cLseSum[c];
        S.death_thr[c] = H.cThr[c]; S.cl_n[c] = H.cN[c]; S.imin_slot[c] = H.cMinSlot[c]; S.cl_uid[c] = H.cUid[c];
    }
    if (tid == 0) {
        ctl->status = status; ctl->error = error; ctl->i_nursery = i_nursery; ctl->admin_epoch = epoch;
        ctl->failures = failures; ctl->ncluster = nc; ctl->ncluster_dead = nc_dead; ctl->ndead = ndead;
        ctl->nphantom = nph; ctl->seg_hi = seg_hi; ctl->seg_lo = i_nursery; ctl->cluster_deleted = cluster_deleted;
        ctl->next_cluster_uid = next_uid; ctl->nlike = nlike; ctl->niter = niter; ctl->nlike_failed = nlike_failed;
        if (ll_s > 0.0) { const double v = ll_m + log(ll_s); live_logZ_val = (v > S.logzero + 800.0) ? v : pc_logaddexp(S.logzero, v); }
        ctl->logZ = logZ; ctl->logZ2 = logZ2; ctl->logX_last_update = lx_last; ctl->live_logZ = live_logZ_val;
        ctl->gen_cyc[0] += cyT; ctl->gen_cyc[1] += cyI; ctl->gen_cyc[2] += cyK; ctl->gen_cyc[3] += cyE; ctl->nn_walks += cyW; ctl->nn_fallbacks += cyF;
    }
    pc_publish_ctl(S);
}

// ------------------------------------------------------------------------------------------
// apply the plan: dead rows + phantoms (reads live[] before k_apply_live overwrites slots)
// one wave per consumed chain
// ------------------------------------------------------------------------------------------
// Copy the rows selected by a wave-uniform bit mask (row of bit b = src0 + b*nT) to consecutive rows at dst.
// Eight rows travel together: their loads are independent, so a batch costs one memory round trip
// instead of eight (rows written by other XCDs come from HBM / Infinity Cache, ~1 us each).
__device__ __forceinline__ int wave_copy_masked(const double *src0, unsigned long long mask, double *dst, int nT, int lane)
{
    int n = 0;
    while (mask) {
        int idx[8], cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            idx[u] = 0;
            if (mask) { idx[u] = __ffsll((long long)mask) - 1; mask &= mask - 1; cnt = u + 1; }
        }
        for (int e = lane; e < nT; e += 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < cnt) v[u] = src0[(size_t)idx[u] * nT + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < cnt) dst[(size_t)(n + u) * nT + e] = v[u];
        }
        n += cnt;
    }
    return n;
}

// four waves per consumed chain: wave 0 also moves the dead row; the phantoms of a mask word are shared out among the
// waves by runs of bits (a wave's rows go to consecutive rows behind those of the waves before it)
#define PC_APPLY_WAVES 4
__device__ __forceinline__ void apply_dead_ph_body(const PcState &S, unsigned batch)
{
    const PcCtl *ctl = S.ctl;
    const int w = ctl->seg_lo + blockIdx.x;
    if (w > ctl->seg_hi) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nT = S.nT, nr = S.nr;
    const int di = S.plan[w].dead_idx;
    if (di >= 0 && wv == 0) {
        const int src = S.plan[w].dead_src;
        const double *row = (src >= 0) ? S.live + (size_t)src * nT
                                       : S.babies + ((size_t)(-src - 1) * nr + (nr - 1)) * nT;
        double *dst = S.dead + (size_t)di * nT;
        for (int e = lane; e < nT; e += 64) dst[e] = row[e];
        if (lane == 0) {
            S.dead_logw[di] = S.plan[w].logw; S.dead_postX[di] = S.plan[w].postX + log(S.plan[w].postXs); S.dead_postZ[di] = S.plan[w].postZ;
            S.dead_cuid[di] = S.plan[w].dead_cuid;
            S.dead_entry[di] = (src >= 0) ? S.live_entry[src] : S.plan[-src - 1].contour;
        }
    }
    int base = S.plan[w].ph_base;
    const unsigned cuid = S.plan[w].ph_cuid;
    if (S.plan[w].ph_count == -2) {
        // a region of nr rows for this chain (parallel contraction): baby i is a phantom if it lies above the contour
        // the chain was consumed at and then occupies row base + i; the rest of the region is marked with PC_CUID_NONE
        const double Lg = S.plan[w].contour;
        for (int m = 0; m < (nr + 63) / 64; ++m) {
            const int i = m * 64 + lane;
            const double bl = (i < nr) ? S.baby_logL[(size_t)w * nr + i] : -PC_HUGE;
            const bool sel = (i < nr - 1) && bl > Lg;
            const unsigned long long mask = __ballot(sel);
            if (wv == 0 && i < nr) {
                S.ph_logL[base + i] = bl; S.ph_cuid[base + i] = sel ? cuid : PC_CUID_NONE;
                S.ph_uid[base + i] = ((unsigned long long)batch << 32) | (unsigned)(w * nr + i);
            }
            // wave k copies the rows of bits [16k, 16k + 16), eight in flight
            unsigned long long mine = mask & (0xFFFFull << (16 * wv));
            const double *src0 = S.babies + ((size_t)w * nr + m * 64) * nT;
            double *dst0 = S.phantom + (size_t)(base + m * 64) * nT;
            while (mine) {
                int idx[8], cnt = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) { idx[u] = 0; if (mine) { idx[u] = __ffsll((long long)mine) - 1; mine &= mine - 1; cnt = u + 1; } }
                for (int e = lane; e < nT; e += 64) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (u < cnt) v[u] = src0[(size_t)idx[u] * nT + e];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (u < cnt) dst0[(size_t)idx[u] * nT + e] = v[u];
                }
            }
        }
        return;
    }
    for (int m = 0; m < (nr + 62) / 64; ++m) {
        const unsigned long long mask = S.plan[w].ph_mask[m];
        // side arrays: lane b of wave 0 owns baby m*64+b, its row offset is the number of selected babies before it
        if (wv == 0 && ((mask >> lane) & 1ull)) {
            const int i = m * 64 + lane, dsti = base + __popcll(mask & ((1ull << lane) - 1ull));
            S.ph_logL[dsti] = S.baby_logL[(size_t)w * nr + i]; S.ph_cuid[dsti] = cuid;
            S.ph_uid[dsti] = ((unsigned long long)batch << 32) | (unsigned)(w * nr + i);
        }
        // wave k takes bits [16k, 16k + 16)
        const unsigned long long mine = mask & (0xFFFFull << (16 * wv));
        const int before = __popcll(mask & ((1ull << (16 * wv)) - 1ull));
        wave_copy_masked(S.babies + ((size_t)w * nr + m * 64) * nT, mine, S.phantom + (size_t)(base + before) * nT, nT, lane);
        base += __popcll(mask);
    }
}
__global__ __launch_bounds__(64 * PC_APPLY_WAVES) void k_apply_dead_ph(PcState S, unsigned batch) { apply_dead_ph_body(S, batch); }
__global__ __launch_bounds__(64 * PC_APPLY_WAVES) void k_apply_dead_ph_many(const PcManyRec *__restrict__ R) { apply_dead_ph_body(pc_many_state(R, blockIdx.y), (unsigned)R[blockIdx.y].ia[0]); }

// pool mode: both of the above in ONE launch (a kernel boundary on the main stream costs 6 us, 79 times per run at the metric
// configuration).  No row is copied to become a phantom, so a chain's workgroup only writes the side arrays of its region and,
// if the point that died at its step was a newcomer of this launch, that baby's row; the rows of snapshot points that died
// are moved out by the workgroup of their SLOT (k_consume_par left the killer's chain in slot_dead) just before the slot's
// new occupant moves in -- the one place where the order of the two copies matters.
// (four wavefronts a workgroup, a chain or a slot each: with one wavefront a workgroup the launch was as long as the dispatcher
//  took to hand 3000 workgroups out -- 192 k of them for sixty-four runs in step, 122 us)
__device__ __forceinline__ void apply_pool_body(const PcState &S, unsigned batch, int nchains)
{
    const PcCtl *ctl = S.ctl;
    const int lane = threadIdx.x & 63, nT = S.nT, nr = S.nr;
    const int item = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (item >= nchains + S.Ncap) return;
    if (item < nchains) {
        const int w = ctl->seg_lo + item;
        if (w > ctl->seg_hi) return;
        const int di = S.plan[w].dead_idx, src = S.plan[w].dead_src;
        if (di >= 0 && src < 0) {
            const double *row = S.babies + ((size_t)(-src - 1) * nr + (nr - 1)) * nT;
            double *dst = S.dead + (size_t)di * nT;
            for (int e = lane; e < nT; e += 64) dst[e] = row[e];
            if (lane == 0) {
                S.dead_logw[di] = S.plan[w].logw; S.dead_postX[di] = S.plan[w].postX + log(S.plan[w].postXs); S.dead_postZ[di] = S.plan[w].postZ;
                S.dead_cuid[di] = S.plan[w].dead_cuid; S.dead_entry[di] = S.plan[-src - 1].contour;
            }
        }
        const int base = S.plan[w].ph_base;
        const unsigned cuid = S.plan[w].ph_cuid;
        const double Lg = S.plan[w].contour;
        for (int i = lane; i < nr; i += 64) {
            const double bl = S.baby_logL[(size_t)w * nr + i];
            S.ph_logL[base + i] = bl; S.ph_cuid[base + i] = ((i < nr - 1) && bl > Lg) ? cuid : PC_CUID_NONE;
            S.ph_uid[base + i] = ((unsigned long long)batch << 32) | (unsigned)(w * nr + i);
        }
        return;
    }
    const int slot = item - nchains;
    const int src = S.slot_src[slot], killer = S.slot_dead[slot];
    if (src < 0 && killer < 0) return;
    double *lrow = S.live + (size_t)slot * nT;
    if (killer >= 0) {
        const int di = S.plan[killer].dead_idx;
        double *dst = S.dead + (size_t)di * nT;
        for (int e = lane; e < nT; e += 64) dst[e] = lrow[e];
        if (lane == 0) {
            S.dead_logw[di] = S.plan[killer].logw; S.dead_postX[di] = S.plan[killer].postX + log(S.plan[killer].postXs); S.dead_postZ[di] = S.plan[killer].postZ;
            S.dead_cuid[di] = S.plan[killer].dead_cuid; S.dead_entry[di] = S.live_entry[slot];
        }
    }
    if (src >= 0) {
        const double *row = S.babies + ((size_t)src * nr + (nr - 1)) * nT;
        double v[4];                                    // (nTotal <= 256 elements a lane: the loads before the stores)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (lane + 64 * q < nT) ? row[lane + 64 * q] : 0.0;
        for (int e = lane + 256; e < nT; e += 64) lrow[e] = row[e];
#pragma unroll
        for (int q = 0; q < 4; ++q) if (lane + 64 * q < nT) lrow[lane + 64 * q] = v[q];
    }
    // (a lane's copies are in program order, and the slot's records below are its wavefront's alone: no barrier)
    if (lane == 0) { S.slot_src[slot] = -1; S.slot_dead[slot] = -1; if (src >= 0) S.live_entry[slot] = S.plan[src].contour; }
}
__global__ __launch_bounds__(256) void k_apply_pool(PcState S, unsigned batch, int nchains) { apply_pool_body(S, batch, nchains); }
__global__ __launch_bounds__(256) void k_apply_pool_many(const PcManyRec *R, int nchains) { apply_pool_body(pc_many_state(R, blockIdx.y), (unsigned)R[blockIdx.y].ia[0], nchains); }


// new live rows: every slot now owned by a chain's last baby
__device__ __forceinline__ void apply_live_body(const PcState &S)
{
    const int slot = blockIdx.x, lane = threadIdx.x, nT = S.nT, nr = S.nr;
    const int src = S.slot_src[slot];
    if (src < 0) return;
    const double *row = S.babies + ((size_t)src * nr + (nr - 1)) * nT;
    double *dst = S.live + (size_t)slot * nT;
    for (int e = lane; e < nT; e += 64) dst[e] = row[e];
    __syncthreads();
    if (lane == 0) { S.slot_src[slot] = -1; S.live_entry[slot] = S.plan[src].contour; }
}
__global__ __launch_bounds__(64) void k_apply_live(PcState S) { apply_live_body(S); }
__global__ __launch_bounds__(64) void k_apply_live_many(const PcManyRec *__restrict__ R) { apply_live_body(pc_many_state(R, blockIdx.y)); }

// ------------------------------------------------------------------------------------------
// install the initial live set: rows -> slots, labels, contour (generate.F90:291-320)
// single workgroup; also trims nprior > nlive through k_consume(final)-like deaths on the host side
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_install_live(PcState S, const double *rows, int n)
{
    const int tid = threadIdx.x;
    __shared__ vk_t red[4];
    for (int s = tid; s < S.Ncap; s += 256) {
        if (s < n) {                                       // (the rows themselves: one device-to-device copy in front of this launch)
            S.live_logL[s] = rows[(size_t)s * S.nT + S.l0]; S.live_cluster[s] = 0; S.live_pos[s] = s; S.live_entry[s] = S.logzero;
            S.cl_list[s] = s;
        } else { S.live_logL[s] = PC_HUGE; S.live_cluster[s] = -1; S.live_pos[s] = 0; S.live_entry[s] = S.logzero; }
        S.slot_src[s] = -1;
    }
    __syncthreads();
    vk_t best{PC_HUGE, 0x7fffffff};
    double mx = -PC_HUGE;
    for (int s = tid; s < n; s += 256) { best = vk_min(best, vk_t{S.live_logL[s], s}); mx = fmax(mx, S.live_logL[s]); }
    best = block_argmin<256>(best, red);
    __shared__ double smx[4];
    mx = wave_max(mx);
    if ((tid & 63) == 0) smx[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    double sum = 0.0;
    for (int s = tid; s < n; s += 256) sum += exp(S.live_logL[s] - mx);
    sum = wave_sum<4>(sum);
    __syncthreads();
    if ((tid & 63) == 0) smx[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        S.cl_n[0] = n; S.logLp[0] = best.v; S.imin_slot[0] = best.k;
        S.lse_ref[0] = mx; S.lse_sum[0] = (smx[0] + smx[1]) + (smx[2] + smx[3]);
    }
}

// ------------------------------------------------------------------------------------------
// update step 1: clean_phantoms (run_time_info.f90:820-877)
// A phantom of cluster c is dropped iff some death of c since the last clean has a larger logL,
// i.e. iff its logL < the logL of c's latest death; phantoms of dead clusters are dropped too.
// Deterministic 3-kernel stream compaction into the alternate phantom buffer.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cluster_of_uid(const PcState &S, unsigned uid, int nc)
{
    for (int c = 0; c < nc; ++c) if (S.cl_uid[c] == uid) return c;
    return -1;
}

__device__ __forceinline__ void clean_flag_body(const PcState &S, int nph, unsigned char *keep, int *blk_count)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int nc = S.ctl->ncluster;
    bool k = false;
    if (j < nph) {
        const int c = cluster_of_uid(S, S.ph_cuid[j], nc);
        k = (c >= 0) && !(S.ph_logL[j] < S.death_thr[c]);
        keep[j] = k ? 1 : 0;
    }
    const unsigned long long m = __ballot(k);
    __shared__ int cnt[4];
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}
__global__ __launch_bounds__(256) void k_clean_flag(PcState S, int nph, unsigned char *keep, int *blk_count) { clean_flag_body(S, nph, keep, blk_count); }
__global__ __launch_bounds__(256) void k_clean_flag_many(const PcManyRec *R) { const PcManyView r = pc_many_view(R, blockIdx.y); if ((int)blockIdx.x >= r.ia[2]) return; clean_flag_body(r.S, r.ia[1], (unsigned char *)r.p[0], (int *)r.p[1]); }


__device__ __forceinline__ void scan_blocks_body(int *blk_count, int nblk, int *total, int *total2)
{   // exclusive scan, single workgroup, fixed order
    __shared__ int carry;
    __shared__ int tmp[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blk_count[i] : 0;
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int t = threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
            __syncthreads();
            tmp[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) blk_count[i] = carry + tmp[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += tmp[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total = carry; if (total2) *total2 = carry; }   // total2: the control block's phantom count
}
__global__ __launch_bounds__(256) void k_scan_blocks(int *blk_count, int nblk, int *total, int *total2) { scan_blocks_body(blk_count, nblk, total, total2); }
__global__ __launch_bounds__(256) void k_scan_blocks_many(const PcManyRec *R) { const PcManyView r = pc_many_view(R, blockIdx.y); scan_blocks_body((int *)r.p[1], r.ia[2], (int *)r.p[2], &r.S.ctl->nphantom); }


__device__ __forceinline__ void clean_scatter_body(const PcState &S, int nph, const unsigned char *keep, const int *blk_off,
                                                   double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2,
                                                   int *dst_index /* [nph] or nullptr */)
{
    const int j = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool k = (j < nph) && keep[j];
    const unsigned long long m = __ballot(k);
    __shared__ int wcnt[4];
    if (lane == 0) wcnt[wid] = __popcll(m);
    __syncthreads();
    int woff = blk_off[blockIdx.x];
    for (int i = 0; i < wid; ++i) woff += wcnt[i];
    const int off = woff + __popcll(m & ((1ull << lane) - 1ull));
    if (dst_index && j < nph) dst_index[j] = k ? off : -1;
    if (k) { phL2[off] = S.ph_logL[j]; phC2[off] = S.ph_cuid[j]; phU2[off] = S.ph_uid[j]; }
    // rows: the wave copies its surviving rows cooperatively (coalesced, eight rows in flight)
    const int j0 = blockIdx.x * 256 + wid * 64;
    wave_copy_masked(S.phantom + (size_t)j0 * S.nT, m, ph2 + (size_t)woff * S.nT, S.nT, lane);
}
__global__ __launch_bounds__(256) void k_clean_scatter(PcState S, int nph, const unsigned char *keep, const int *blk_off,
                                                      double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2, int *dst_index)
{ clean_scatter_body(S, nph, keep, blk_off, ph2, phL2, phC2, phU2, dst_index); }
__global__ __launch_bounds__(256) void k_clean_scatter_many(const PcManyRec *R)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    if ((int)blockIdx.x >= r.ia[2]) return;
    clean_scatter_body(r.S, r.ia[1], (const unsigned char *)r.p[0], (const int *)r.p[1], (double *)r.p[3], (double *)r.p[4], (unsigned *)r.p[5], (unsigned long long *)r.p[6], nullptr);
}


__global__ void k_reset_thresholds(PcState S) { for (int c = threadIdx.x; c < S.maxc; c += blockDim.x) S.death_thr[c] = -PC_HUGE; }
__global__ void k_reset_thresholds_many(const PcManyRec *__restrict__ R) { const PcState S = pc_many_state(R, blockIdx.y); for (int c = threadIdx.x; c < S.maxc; c += blockDim.x) S.death_thr[c] = -PC_HUGE; }

// ------------------------------------------------------------------------------------------
// update step 2: calculate_covmats (run_time_info.f90:601-641), two passes, fixed-order sums
// pass A: per (chunk, cluster) partial sums of cube coordinates; pass B: centred outer products
// rows = live slots followed by phantoms.  partial buffers: [nchunk][nc][D] and [nchunk][nc][D*D]
// ------------------------------------------------------------------------------------------
#define PC_COV_ROWS 256          /* rows per chunk */

__device__ __forceinline__ const double *cov_row(const PcState &S, int r, int nlive_slots, int nc, int &c)
{
    if (r < nlive_slots) { c = S.live_cluster[r]; return S.live + (size_t)r * S.nT; }
    const int j = r - nlive_slots;
    c = cluster_of_uid(S, S.ph_cuid[j], nc);
    return S.phantom + (size_t)j * S.nT;
}

__device__ __forceinline__ const double *cov_ptr(const PcState &S, int r)
{
    return (r < S.Ncap) ? S.live + (size_t)r * S.nT : S.phantom + (size_t)(r - S.Ncap) * S.nT;
}

// thread layout of the reductions below: DPc = coordinates handled side by side, G = 256 / DPc row groups
__device__ __forceinline__ int cov_dpc(int D) { int p = 32; while (p < D && p < 256) p <<= 1; return p; }

__global__ __launch_bounds__(256) void k_cov_mean_partial(PcState S, int nrows, int nph, double *psum, int *pcnt, int CR)
{
    // grid (nchunk, nc); thread = (row group g, coordinate d): group g sums rows r0+g, r0+g+G, ... of
    // cluster c in that order, the groups are added in order: a fixed summation tree
    const int chunk = blockIdx.x, c = blockIdx.y, nc = gridDim.y, D = S.D, tid = threadIdx.x;
    nrows = min(nrows, S.Ncap + S.ctl->nphantom);   // the grid may have been sized before the phantoms were cleaned
    const int r0 = chunk * CR, r1 = min(nrows, r0 + CR);
    const int DPc = cov_dpc(D), G = 256 / DPc, g = tid / DPc;
    __shared__ int rc[PC_COV_ROWS];
    __shared__ double red[256];
    int mine = 0;
    for (int r = r0 + tid; r < r1; r += 256) { int cc; cov_row(S, r, S.Ncap, nc, cc); rc[r - r0] = cc; mine += (cc == c); }
    const int cnt = __syncthreads_count(mine);
    for (int d = tid % DPc; d < D; d += DPc) {
        double s = 0.0;
        int r = r0 + g;
        for (; r + 7 * G < r1; r += 8 * G) {              // eight independent loads in flight, added in row order
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = (rc[r + u * G - r0] == c) ? cov_ptr(S, r + u * G)[d] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t8[u];
        }
        for (; r < r1; r += G)
            if (rc[r - r0] == c) s += cov_ptr(S, r)[d];
        if (G > 1) red[tid] = s; else psum[((size_t)chunk * nc + c) * D + d] = s;
    }
    if (G > 1) {
        __syncthreads();
        if (tid < D) {
            double s = 0.0;
            for (int gg = 0; gg < G; ++gg) s += red[gg * DPc + tid];
            psum[((size_t)chunk * nc + c) * D + tid] = s;
        }
    }
    if (tid == 0) pcnt[(size_t)chunk * nc + c] = cnt;
}

__global__ __launch_bounds__(256) void k_cov_partial(PcState S, int nrows, int nchunk, const double *psum, const int *pcnt,
                                                    double *mean /* [nc][D] */, int *count /* [nc] */, double *pcov, int CR,
                                                    int TS /* tile row stride */, int use_mfma, int nchunk_rows)
{
    // grid (nchunk, nc).  Every workgroup first reduces the chunk sums to the cluster mean (fixed order,
    // identical in every workgroup), then accumulates the centred outer products of its own rows.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int chunk = blockIdx.x, c = blockIdx.y, nc = gridDim.y, D = S.D, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    nrows = min(nrows, S.Ncap + S.ctl->nphantom);
    double *tile = (double *)smem;               // [rows][TS], TS = D+1 or (D rounded up to 16)+1, zero padded
    double *mu = tile + (size_t)CR * TS;         // [D]
    double *red = mu + D;                        // [256]
    int *rc = (int *)(red + 256);                // [CR] member rows, in row order
    __shared__ int wcnt[4];
    __shared__ int nred[256];
    const int DPc = cov_dpc(D), G = 256 / DPc, g = tid / DPc;
    // ---- mean
    {
        int nn = 0;
        for (int k = tid; k < nchunk; k += 256) nn += pcnt[(size_t)k * nc + c];
        nred[tid] = nn;
    }
    for (int d = tid % DPc; d < D; d += DPc) {
        double s = 0.0;
        int k = g;
        for (; k + 7 * G < nchunk; k += 8 * G) {          // eight independent loads in flight, added in order
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = psum[((size_t)(k + u * G) * nc + c) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t8[u];
        }
        for (; k < nchunk; k += G) s += psum[((size_t)k * nc + c) * D + d];
        if (G > 1) red[tid] = s; else mu[d] = s;
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) nred[tid] += nred[tid + off]; __syncthreads(); }
    const int ntot = nred[0];
    if (G > 1 && tid < D) {
        double s = 0.0;
        for (int gg = 0; gg < G; ++gg) s += red[gg * DPc + tid];
        mu[tid] = s;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        const double m = mu[d] / (double)ntot;
        mu[d] = m;
        if (chunk == 0) mean[(size_t)c * D + d] = m;
    }
    if (chunk == 0 && tid == 0) count[c] = ntot;
    // ---- wide nDims: X^T X on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), upper triangle only, one 16x16 tile of
    // the result per wave at a time.  Operand maps (cdna_hip_programming.md): A[i=lane&15][k=lane>>4],
    // B[k=lane>>4][j=lane&15], D col = lane&15, row = (lane>>4) + 4*reg.  The workgroup is persistent: it takes the chunks
    // chunk, chunk + gridDim.x, ... and keeps its accumulators in registers across them, so that ONE partial matrix
    // per workgroup goes to memory (a 100-D update used to write and re-read one 80 KB partial per 128 rows: 1.3 GB).
    // Rows are consumed four at a time in row order, chunks in order: a fixed summation order.
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int MAXT = 9;                                       // tiles per wave: nDims <= 128 -> 36 tiles / 4 waves
    v4d acc[MAXT];
    const int W = TS - 1, nt = W / 16, ntile = nt * (nt + 1) / 2;
    // beyond 128 dimensions there are more result tiles than a wave's registers hold (136 at nDims = 256): the
    // workgroup walks its chunks once per group of 4 x MAXT tiles
    for (int tile0 = 0; tile0 == 0 || (use_mfma && tile0 < ntile); tile0 += 4 * MAXT) {
#pragma unroll
    for (int k = 0; k < MAXT; ++k) acc[k] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int ch = chunk; ch == chunk || (use_mfma && ch < nchunk_rows); ch += gridDim.x) {
    const int r0 = ch * CR, r1 = max(r0, min(nrows, r0 + CR));
    // ---- member rows of this chunk, in row order (CR <= 256: one row per thread)
    bool member = false;
    { const int r = r0 + tid; int cc = -1; if (tid < CR && r < r1) cov_row(S, r, S.Ncap, nc, cc); member = (tid < CR && r < r1 && cc == c); }
    const unsigned long long bm = __ballot(member);
    __syncthreads();                                              // the previous chunk's tile and rc are no longer read
    if (lane == 0) wcnt[wv] = __popcll(bm);
    __syncthreads();
    int base = 0;
    for (int x = 0; x < wv; ++x) base += wcnt[x];
    const int n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (member) rc[base + __popcll(bm & ((1ull << lane) - 1ull))] = r0 + tid;
    __syncthreads();
    if (!use_mfma) {
        {
            int e = tid;
            for (; e + 7 * 256 < n * D; e += 8 * 256) {   // eight independent row reads in flight
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int ee = e + u * 256; t8[u] = cov_ptr(S, rc[ee / D])[ee % D]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int ee = e + u * 256, i = ee / D, d = ee % D; tile[(size_t)i * TS + d] = t8[u] - mu[d]; }
            }
            for (; e < n * D; e += 256) { const int i = e / D, d = e % D; tile[(size_t)i * TS + d] = cov_ptr(S, rc[i])[d] - mu[d]; }
        }
        __syncthreads();
        for (int p = tid; p < D * D; p += 256) {
            const int a = p / D, b = p % D;
            // rows i, i+1, i+2, i+3 on four accumulators (four independent rows to a pass)
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int i = 0;
            for (; i + 3 < n; i += 4) {
                s0 += tile[(size_t)i * TS + a] * tile[(size_t)i * TS + b];
                s1 += tile[(size_t)(i + 1) * TS + a] * tile[(size_t)(i + 1) * TS + b];
                s2 += tile[(size_t)(i + 2) * TS + a] * tile[(size_t)(i + 2) * TS + b];
                s3 += tile[(size_t)(i + 3) * TS + a] * tile[(size_t)(i + 3) * TS + b];
            }
            for (; i < n; ++i) s0 += tile[(size_t)i * TS + a] * tile[(size_t)i * TS + b];
            pcov[((size_t)chunk * nc + c) * D * D + p] = (s0 + s1) + (s2 + s3);
        }
        return;
    }
    const int nP = (n + 3) & ~3;                                  // rows of the zero padded tile
    for (int e = tid; e < nP * W; e += 256) {
        const int i = e / W, d = e % W;
        tile[(size_t)i * TS + d] = (i < n && d < D) ? cov_ptr(S, rc[i])[d] - mu[d] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        const int t = tile0 + wv + 4 * k;
        if (t < ntile) {
            int ti = 0, rem = t;
            while (rem >= nt - ti) { rem -= nt - ti; ti++; }
            const int tj = ti + rem;
            const double *pa = tile + (size_t)(lane >> 4) * TS + ti * 16 + (lane & 15);
            const double *pb = tile + (size_t)(lane >> 4) * TS + tj * 16 + (lane & 15);
            v4d a4 = acc[k];
            for (int q0 = 0; q0 < nP; q0 += 4)
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)q0 * TS], pb[(size_t)q0 * TS], a4, 0, 0, 0);
            acc[k] = a4;
        }
    }
    }   // chunks of this workgroup
    double *out = pcov + ((size_t)chunk * nc + c) * D * D;
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        const int t = tile0 + wv + 4 * k;
        if (t < ntile) {
            int ti = 0, rem = t;
            while (rem >= nt - ti) { rem -= nt - ti; ti++; }
            const int tj = ti + rem;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
                if (row < D && col < D) { out[(size_t)row * D + col] = acc[k][r]; if (ti != tj) out[(size_t)col * D + row] = acc[k][r]; }
            }
        }
    }
    }   // groups of tiles
}

// Fold the per-chunk partial sums of a cluster into the first R chunk slots (slot r = chunks r, r+R, ... added in
// that order; slot r is only ever read by workgroup r, so the fold is in place).  A 100-D run has ~800 chunks
// of 80 KB each: one workgroup reading all of them for the final sum took 3 ms per update.
__global__ __launch_bounds__(256) void k_fold_partials(double *buf, int *cnt, int nchunk, int nc, int per)
{
    const int r = blockIdx.x, R = gridDim.x, c = blockIdx.y, tid = threadIdx.x;
    for (int p = tid; p < per; p += 256) {
        double s = 0.0;
        int k = r;
        for (; k + 7 * R < nchunk; k += 8 * R) {
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = buf[((size_t)(k + u * R) * nc + c) * per + p];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t8[u];
        }
        for (; k < nchunk; k += R) s += buf[((size_t)k * nc + c) * per + p];
        buf[((size_t)r * nc + c) * per + p] = s;
    }
    if (cnt && tid == 0) {
        int n = 0;
        for (int k = r; k < nchunk; k += R) n += cnt[(size_t)k * nc + c];
        cnt[(size_t)r * nc + c] = n;
    }
}

#define PC_CHOL_NT 1024
__global__ __launch_bounds__(PC_CHOL_NT) void k_cov_final_chol(PcState S, int nchunk, const double *pcov, const int *count, int a_global, int only_if_suspect)
{
    if (only_if_suspect && !S.ctl->chol_suspect) return;          // (uniform) the blocked factorisation in front of this launch stands
    // one workgroup per cluster: fixed-order sum of the partials, then calc_cholesky (utils.F90:621-649)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int c = blockIdx.x, nc = gridDim.x, D = S.D, DD = D * D, tid = threadIdx.x;
    // red [PC_CHOL_NT]: scratch of the reduction; it borrows L (zeroed afterwards) when the two matrices
    // alone fill the LDS (nDims = 100: 2 x 80 KB)
    // a_global (nDims > 101: two matrices exceed the LDS): the covariance is read back from S.cov, only L lives in LDS
    // a_global == 2 (nDims > 143: not even L fits): the factor is built in place in S.chol, the LDS holds the scratch only
    double *A = a_global ? S.cov + (size_t)c * DD : (double *)smem;
    double *L = a_global == 2 ? S.chol + (size_t)c * DD : (a_global ? (double *)smem : A + DD);
    double *red = a_global == 2 ? (double *)smem : ((DD >= PC_CHOL_NT) ? L : L + DD);
    __shared__ int bad;
    const double n = (double)count[c];
    // G groups of DDp threads; group g adds chunks g, g+G, ... in order, then the groups are added in order
    int DDp = (DD + 63) & ~63;
    if (DDp > PC_CHOL_NT) DDp = PC_CHOL_NT;
    const int G = PC_CHOL_NT / DDp, g = tid / DDp;
    for (int p0 = 0; p0 < DD; p0 += DDp) {
        const int p = p0 + tid % DDp;
        double s = 0.0;
        if (p < DD && g < G)
        {
            int k = g;
            for (; k + 7 * G < nchunk; k += 8 * G) {      // eight independent loads in flight, added in order
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t8[u] = pcov[((size_t)(k + u * G) * nc + c) * DD + p];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t8[u];
            }
            for (; k < nchunk; k += G) s += pcov[((size_t)k * nc + c) * DD + p];
        }
        red[tid] = s;
        __syncthreads();
        if (tid < DDp && p < DD) {
            double t = 0.0;
            for (int gg = 0; gg < G; ++gg) t += red[gg * DDp + tid];
            const double av = t / n;
            if (!a_global) A[p] = av;
            S.cov[(size_t)c * DD + p] = av;
        }
        __syncthreads();
    }
    for (int p = tid; p < DD; p += PC_CHOL_NT) L[p] = 0.0;
    if (tid == 0) bad = 0;
    __syncthreads();
    // calc_cholesky by ONE wavefront, lane = row: column i needs, for every row j >= i, the dot product
    // sum_{k<i} L(i,k) L(j,k) in ascending k (the reference's order); row i's own value gives the
    // diagonal.  No workgroup barrier inside the column loop, only wave-level LDS ordering.
    if (a_global == 2) {
        // The factor is built TRANSPOSED in HBM (L(j,k) at [k][j]): lane = row, so a wave reads one contiguous stretch per
        // k instead of 64 cache lines, the four rows a lane owns are four independent sums in the same ascending-k order,
        // and a column is stored as one contiguous row.  Transposed in place at the end.
        if (tid < 64) {
            for (int i = 0; i < D; ++i) {
                double tj[4] = {0.0, 0.0, 0.0, 0.0};
                int jc[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) jc[m] = min(m * 64 + tid, D - 1);     // rows past D: a harmless duplicate
                int k = 0;
                for (; k + 8 <= i; k += 8) {                 // forty loads in flight, then the sums in ascending k
                    double lik[8], l4[8][4];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double *Lk = L + (size_t)(k + u) * D;
                        lik[u] = Lk[i];
#pragma unroll
                        for (int m = 0; m < 4; ++m) l4[u][m] = Lk[jc[m]];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int m = 0; m < 4; ++m) tj[m] += lik[u] * l4[u][m];
                }
                for (; k < i; ++k) {
                    const double *Lk = L + (size_t)k * D;
                    const double lik = Lk[i];
#pragma unroll
                    for (int m = 0; m < 4; ++m) tj[m] += lik * Lk[jc[m]];
                }
                double ti = 0.0;
#pragma unroll
                for (int m = 0; m < 4; ++m) { const double v = __shfl(tj[m], i & 63); if ((i >> 6) == m) ti = v; }
                const double dii = A[i * D + i] - ti;
                if (dii <= 0.0) { if (tid == 0) bad = 1; break; }
                const double lii = sqrt(dii);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = m * 64 + tid;
                    if (j > i && j < D) L[(size_t)i * D + j] = (A[i * D + j] - tj[m]) / lii;
                }
                if (tid == 0) L[(size_t)i * D + i] = lii;
                __threadfence_block();
            }
        }
        __syncthreads();
        if (!bad)
            for (int p = tid; p < DD; p += PC_CHOL_NT) {
                const int a = p / D, b = p % D;
                if (a < b) { const double x = L[p], y = L[(size_t)b * D + a]; L[p] = y; L[(size_t)b * D + a] = x; }
            }
    }
    else if (tid < 64) {
        for (int i = 0; i < D; ++i) {
            double tj[4] = {0.0, 0.0, 0.0, 0.0};           // D <= 256: at most 4 rows per lane
            #pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = m * 64 + tid;
                if (j >= i && j < D) {
                    double t = 0.0;
                    for (int k = 0; k < i; ++k) t += L[i * D + k] * L[j * D + k];
                    tj[m] = t;
                }
            }
            double ti = 0.0;                                // row i's own dot product
            #pragma unroll
            for (int m = 0; m < 4; ++m) { const double v = __shfl(tj[m], i & 63); if ((i >> 6) == m) ti = v; }
            const double dii = A[i * D + i] - ti;
            if (dii <= 0.0) { if (tid == 0) bad = 1; break; }
            const double lii = sqrt(dii);
            #pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = m * 64 + tid;
                if (j > i && j < D) L[j * D + i] = (A[i * D + j] - tj[m]) / lii;
            }
            if (tid == 0) L[i * D + i] = lii;
            __threadfence_block();
        }
    }
    __syncthreads();
    if (bad && tid == 0 && (S.ablate & (1 << 30))) printf("engine chol fallback: cluster %d of %d, n = %d\n", c, nc, (int)n);   // developer trace
    if (bad) {   // no Cholesky factor: scaled identity (utils.F90:633-638)
        double tr = 0.0;
        for (int k = 0; k < D; ++k) tr += A[k * D + k];
        for (int p = tid; p < DD; p += PC_CHOL_NT) L[p] = (p / D == p % D) ? sqrt(tr) : 0.0;
    }
    __syncthreads();
    if (a_global != 2) for (int p = tid; p < DD; p += PC_CHOL_NT) S.chol[(size_t)c * DD + p] = L[p];
}

// ------------------------------------------------------------------------------------------
// initial values of the per-cluster state and of the control block (initialise_run_time_info,
// run_time_info.f90:164-206): one launch instead of a dozen small host copies
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_state(PcState S, double logzero)
{
    const int tid = threadIdx.x, maxc = S.maxc, D = S.D;
    for (int m = tid; m < maxc; m += 256) {
        S.logZp[m] = logzero; S.logZXp[m] = logzero; S.logZp2[m] = logzero; S.logZpXp[m] = logzero; S.logLp[m] = logzero;
        S.logXp[m] = 0.0; S.lse_ref[m] = 0.0; S.lse_sum[m] = 0.0; S.death_thr[m] = -PC_HUGE;
        S.cl_n[m] = 0; S.cl_uid[m] = 0u; S.imin_slot[m] = 0;
    }
    for (int e = tid; e < maxc * maxc; e += 256) S.XpXq[e] = 0.0;
    for (size_t e = tid; e < (size_t)maxc * D * D; e += 256) {
        const int r = (int)(e % ((size_t)D * D));
        const double v = (r / D == r % D) ? 1.0 : 0.0;
        S.chol[e] = v; S.cov[e] = v;
    }
    if (tid == 0) {
        PcCtl c0{};
        c0.status = PC_ST_RUNNING; c0.ncluster = 1; c0.logZ = logzero; c0.logZ2 = logzero;
        c0.logX_last_update = 0.0; c0.next_cluster_uid = 1; c0.live_logZ = logzero;
        *S.ctl = c0;
    }
}

// ------------------------------------------------------------------------------------------
// posterior moments of theta from the dead points: weights exp(logw + logL - max), fixed-order sums
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_max(PcState S, int nd, double *pmax)
{
    __shared__ double red[256];
    if (nd < 0) nd = S.ctl->ndead;                      // (enqueued behind the kill-off, before the host knows the count)
    double m = -PC_HUGE;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nd; i += gridDim.x * 256) {
        const double lw = S.dead_logw[i];
        if (lw > S.logzero) m = fmax(m, lw + S.dead[(size_t)i * S.nT + S.l0]);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]); __syncthreads(); }
    if (threadIdx.x == 0) pmax[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_post_moments(PcState S, int nd, const double *pmax, double *part /* [grid][2D+1] */)
{
    // thread = (row group g, coordinate d); group g takes rows blockIdx*G+g, +gridDim*G, ... in order
    __shared__ double red[256];
    __shared__ double red2[256];
    const int tid = threadIdx.x, D = S.D + S.nDer;          // theta then phi, contiguous from p0
    const int DPc = cov_dpc(D), G = 256 / DPc, g = tid / DPc;
    if (nd < 0) nd = S.ctl->ndead;
    double m = -PC_HUGE;
    for (int b = tid; b < (int)gridDim.x; b += 256) m = fmax(m, pmax[b]);
    red[tid] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] = fmax(red[tid], red[tid + off]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += DPc) {
        const int d = d0 + tid % DPc;
        double s1 = 0.0, s2 = 0.0, sw = 0.0;
        for (int i = blockIdx.x * G + g; i < nd; i += gridDim.x * G) {
            const double lw = S.dead_logw[i];
            if (!(lw > S.logzero)) continue;
            const double *row = S.dead + (size_t)i * S.nT;
            const double wgt = exp(lw + row[S.l0] - m);
            sw += wgt;
            if (d < D) { const double th = row[S.p0 + d]; s1 += wgt * th; s2 += wgt * th * th; }
        }
        red[tid] = s1; red2[tid] = s2;
        __syncthreads();
        if (tid < DPc && d < D) {
            double a = 0.0, b2 = 0.0;
            for (int gg = 0; gg < G; ++gg) { a += red[gg * DPc + tid]; b2 += red2[gg * DPc + tid]; }
            part[(size_t)blockIdx.x * (2 * D + 1) + d] = a; part[(size_t)blockIdx.x * (2 * D + 1) + D + d] = b2;
        }
        __syncthreads();
        if (d0 == 0) {
            red[tid] = (tid % DPc == 0) ? sw : 0.0;
            __syncthreads();
            if (tid == 0) { double a = 0.0; for (int gg = 0; gg < G; ++gg) a += red[gg * DPc]; part[(size_t)blockIdx.x * (2 * D + 1) + 2 * D] = a; }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static size_t consume_lds(const PcState *S, int NT, int xrows, int xq_n = 0, int slots_global = 0)
{
    if (xq_n > 0) return consume_lds(S, NT, xrows, 0, slots_global) + sizeof(double) * (size_t)xq_n * xq_n + 16;
    const size_t nslot = slots_global ? 0 : (size_t)S->Ncap;
    return sizeof(double) * (nslot + 9 * (size_t)S->maxc + 2 * NT + (size_t)S->D * PC_IDG + (size_t)xrows * S->D) +
           sizeof(vk_t) * 16 + sizeof(int) * (3 * nslot + 3 * (size_t)S->maxc + 8 + NT + S->nr + (size_t)S->B) + 64;
}
// the per-slot arrays do not fit next to the smallest coordinate tile: keep them in HBM
static int consume_slots_global(const PcState *S, int NT) { return consume_lds(S, NT, 32) > 150 * 1024; }

extern "C" int pc_launch_consume(const PcState *S, int final_mode, int wide, hipStream_t st)
{
    // With the candidate lists the heavy parallel part (the nearest-cluster search) is gone and what is left is barriers
    // and short scans: 256 threads (4 waves) run it 1.4x faster than 1024 (16 waves, 128-VGPR cap).  PC_CONSUME_NT overrides.
    static const int wide_nt = std::getenv("PC_CONSUME_NT") ? std::atoi(std::getenv("PC_CONSUME_NT")) : 256;
    if (wide && (S->nn_valid || final_mode == 1) && wide_nt != 1024) {
#define PC_CONSUME_LAUNCH(NTV) { \
            /* the lists make the coordinate cache a fallback: a tile is enough, the LDS goes to the cross-volume matrix */ \
            const int sgl = consume_slots_global(S, NTV); \
            int cache_x = S->Ncap < 64 ? S->Ncap : 64, xq_n = S->maxc; \
            while (xq_n > 8 && consume_lds(S, NTV, cache_x, xq_n, sgl) > 158 * 1024) xq_n -= 8; \
            const size_t sh = consume_lds(S, NTV, cache_x, xq_n, sgl); \
            if (sh > 160 * 1024) return 1; \
            pc_need_dyn_lds((const void *)k_consume<NTV>, sh); \
            hipLaunchKernelGGL((k_consume<NTV>), dim3(1), dim3(NTV), sh, st, *S, final_mode, cache_x, xq_n, sgl); \
            return 0; }
        if (wide_nt == 64) PC_CONSUME_LAUNCH(64)
        if (wide_nt == 128) PC_CONSUME_LAUNCH(128)
        if (wide_nt == 512) PC_CONSUME_LAUNCH(512)
        PC_CONSUME_LAUNCH(256)
#undef PC_CONSUME_LAUNCH
    }
    if (wide) {
        // all live coordinates in LDS when they fit, else the largest tile that does
        const int sgl = consume_slots_global(S, 1024);
        int cache_x = S->Ncap;
        while (cache_x > 32 && consume_lds(S, 1024, cache_x, 0, sgl) > 158 * 1024) cache_x = (cache_x + 1) / 2;
        const size_t sh = consume_lds(S, 1024, cache_x, 0, sgl);
        if (sh > 160 * 1024) return 1;
        pc_need_dyn_lds((const void *)k_consume<1024>, sh);
        hipLaunchKernelGGL((k_consume<1024>), dim3(1), dim3(1024), sh, st, *S, final_mode, cache_x, 0, sgl);
    } else {
        const int sgl = consume_slots_global(S, 64);
        int xr = S->Ncap;
        while (xr > 32 && consume_lds(S, 64, xr, 0, sgl) > 158 * 1024) xr = (xr + 1) / 2;
        const size_t sh = consume_lds(S, 64, xr, 0, sgl);
        if (sh > 160 * 1024) return 1;
        pc_need_dyn_lds((const void *)k_consume<64>, sh);
        hipLaunchKernelGGL((k_consume<64>), dim3(1), dim3(64), sh, st, *S, final_mode, xr, 0, sgl);
    }
    return 0;
}

extern "C" void pc_launch_apply(const PcState *S, unsigned batch, int nchains, hipStream_t st)
{
    if (S->pool) { hipLaunchKernelGGL(k_apply_pool, dim3((nchains + S->Ncap + 3) / 4), dim3(256), 0, st, *S, batch, nchains); return; }
    hipLaunchKernelGGL(k_apply_dead_ph, dim3(nchains), dim3(64 * PC_APPLY_WAVES), 0, st, *S, batch);
    hipLaunchKernelGGL(k_apply_live, dim3(S->Ncap), dim3(64), 0, st, *S);
}

extern "C" int pc_launch_apply_many(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st)
{
    if (!S->pool) {      // (the runs of a launch share the layout: Cohort::flush groups them by it)
        hipLaunchKernelGGL(k_apply_dead_ph_many, dim3(nchains, R), dim3(64 * PC_APPLY_WAVES), 0, st, dR);
        hipLaunchKernelGGL(k_apply_live_many, dim3(S->Ncap, R), dim3(64), 0, st, dR);
        return 0;
    }
    static const int wpg = std::getenv("PC_APPLY_WAVES") ? std::max(1, std::min(4, std::atoi(std::getenv("PC_APPLY_WAVES")))) : 4;
    hipLaunchKernelGGL(k_apply_pool_many, dim3((nchains + S->Ncap + wpg - 1) / wpg, R), dim3(64 * wpg), 0, st, dR, nchains);      // (the nursery's number: each run's own, PcManyRec::ia[0])
    return 0;
}

// the phantom clean for R runs at once (pool compaction of runs in step): every run its own row count (PcManyRec::ia[1], blocks ia[2])
extern "C" int pc_launch_clean_many(const PcManyRec *dR, int R, int nblk_max, hipStream_t st)
{
    if (nblk_max < 1) return 1;
    hipLaunchKernelGGL(k_clean_flag_many, dim3(nblk_max, R), dim3(256), 0, st, dR);
    hipLaunchKernelGGL(k_scan_blocks_many, dim3(1, R), dim3(256), 0, st, dR);
    hipLaunchKernelGGL(k_clean_scatter_many, dim3(nblk_max, R), dim3(256), 0, st, dR);
    return 0;
}

extern "C" void pc_launch_install_live(const PcState *S, const double *rows, int n, hipStream_t st)
{
    if (n > 0) (void)hipMemcpyAsync(S->live, rows, sizeof(double) * (size_t)n * S->nT, hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(k_install_live, dim3(1), dim3(256), 0, st, *S, rows, n);
}

// phantom clean: returns nothing; *d_total (device int) receives the surviving count
extern "C" void pc_launch_clean(const PcState *S, int nph, unsigned char *keep, int *blk, int *d_total, double *ph2,
                                double *phL2, unsigned *phC2, unsigned long long *phU2, int *dst_index, hipStream_t st)
{
    const int nblk = (nph + 255) / 256;
    if (nblk > 0) {
        hipLaunchKernelGGL(k_clean_flag, dim3(nblk), dim3(256), 0, st, *S, nph, keep, blk);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, blk, nblk, d_total, &S->ctl->nphantom);
        hipLaunchKernelGGL(k_clean_scatter, dim3(nblk), dim3(256), 0, st, *S, nph, keep, blk, ph2, phL2, phC2, phU2, dst_index);
    } else {
        hipMemsetAsync(d_total, 0, sizeof(int), st);
        hipMemsetAsync(&S->ctl->nphantom, 0, sizeof(int), st);
    }
}

extern "C" void pc_launch_reset_thresholds(const PcState *S, hipStream_t st)
{
    hipLaunchKernelGGL(k_reset_thresholds, dim3(1), dim3(S->maxc <= 1024 ? ((S->maxc + 63) / 64) * 64 : 1024), 0, st, *S);
}

extern "C" int pc_launch_reset_thresholds_many(const PcState *S, const PcManyRec *dR, int R, hipStream_t st)
{
    (void)S;
    hipLaunchKernelGGL(k_reset_thresholds_many, dim3(1, R), dim3(256), 0, st, dR);
    return 0;
}

static int cov_use_mfma(const PcState *S) { return S->D >= 32; }
static int cov_tile_stride(const PcState *S) { return cov_use_mfma(S) ? ((S->D + 15) & ~15) + 1 : S->D + 1; }
static int cov_rows(const PcState *S)
{   // rows per chunk: the centred tile [rows][stride] must fit in LDS
    int r = PC_COV_ROWS;
    while (r > 8 && sizeof(double) * (size_t)r * cov_tile_stride(S) + sizeof(int) * r > 120 * 1024) r >>= 1;
    return r;
}
extern "C" int pc_cov_nchunk(const PcState *S, int nph) { const int r = cov_rows(S); return (S->Ncap + nph + r - 1) / r; }

extern "C" int pc_launch_covmats(const PcState *S, int nph, int nc, double *psum, int *pcnt, double *mean, int *count,
                                 double *pcov, hipStream_t st)
{
    const int CR = cov_rows(S);
    const int nrows = S->Ncap + nph, nchunk = (nrows + CR - 1) / CR, D = S->D;
    hipLaunchKernelGGL(k_cov_mean_partial, dim3(nchunk, nc), dim3(256), 0, st, *S, nrows, nph, psum, pcnt, CR);
    const int NFOLD = 64;
    int nred = nchunk;                                           // partial sums the later stages read
    if (nchunk > 2 * NFOLD) {
        hipLaunchKernelGGL(k_fold_partials, dim3(NFOLD, nc), dim3(256), 0, st, psum, pcnt, nchunk, nc, D);
        nred = NFOLD;
    }
    const int TS = cov_tile_stride(S);
    const size_t sh = sizeof(double) * ((size_t)CR * TS + D + 256) + sizeof(int) * CR;
    if (sh > 160 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_cov_partial, sh);
    int ncov = nred;                                             // partial matrices the Cholesky kernel adds up
    if (cov_use_mfma(S)) {
        // persistent workgroups (one per CU at nDims = 100), each walks its share of the chunks and writes one partial matrix
        const int G = nchunk < 256 ? nchunk : 256;
        hipLaunchKernelGGL(k_cov_partial, dim3(G, nc), dim3(256), sh, st, *S, nrows, nred, psum, pcnt, mean, count, pcov, CR, TS, 1, nchunk);
        ncov = G;
    } else {
        hipLaunchKernelGGL(k_cov_partial, dim3(nchunk, nc), dim3(256), sh, st, *S, nrows, nred, psum, pcnt, mean, count, pcov, CR, TS, 0, nchunk);
        if (nchunk > 2 * NFOLD) hipLaunchKernelGGL(k_fold_partials, dim3(NFOLD, nc), dim3(256), 0, st, pcov, (int *)nullptr, nchunk, nc, D * D);
    }
    size_t sh2 = sizeof(double) * (2 * (size_t)D * D + ((size_t)D * D >= PC_CHOL_NT ? 0 : PC_CHOL_NT));
    int a_global = 0;
    if (sh2 > 160 * 1024) { a_global = 1; sh2 = sizeof(double) * ((size_t)D * D + ((size_t)D * D >= PC_CHOL_NT ? 0 : PC_CHOL_NT)); }
    if (sh2 > 160 * 1024) { a_global = 2; sh2 = sizeof(double) * PC_CHOL_NT; }
    pc_need_dyn_lds((const void *)k_cov_final_chol, sh2);
    hipLaunchKernelGGL(k_cov_final_chol, dim3(nc), dim3(PC_CHOL_NT), sh2, st, *S, ncov, pcov, count, a_global, 0);
    return 0;
}

// calc_cholesky (utils.F90:621-649) for 32 <= nDims <= 128, one cluster, as a blocked right-looking factorisation: panels of
// sixteen columns -- the 16 x 16 diagonal block by one wave with a row per lane in registers, the rows below it by forward
// substitution (a thread per row), the trailing matrix A22 -= L21 L21^T in 16 x 16 tiles on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, four contraction steps per tile and panel) -- with the matrix in LDS throughout.  The reference
// sums its dot products in ascending k, one column at a time; this sums the same products panel by panel: the factor agrees to
// round-off.  Where that is not good enough -- a pivot that is not clearly positive, i.e. a covariance that is singular to
// working precision, where round-off decides whether the reference falls back to the scaled identity (utils.F90:633-638) --
// the kernel says so (PcCtl::chol_suspect) and the reference-order kernel launched behind it redoes the factorisation;
// otherwise that launch returns at once.  One-wave kernel: 0.35 ms per update at nDims = 100, this: 0.03.
template <int NT>
__global__ __launch_bounds__(256) void k_chol_blocked(PcState S, const double *ncov, const int *count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int NR = 16 * NT, NS = NR + 1;
    double *A = (double *)smem;                                         // [NR][NS]
    __shared__ int suspect;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4, D = S.D;
    const double n = (double)count[0];
    for (int e = tid; e < NR * NR; e += 256) {
        const int i = e / NR, j = e - i * NR;
        double v = (i == j) ? 1.0 : 0.0;                                // rows and columns beyond nDims: identity
        if (i < D && j < D) { v = ncov[(size_t)i * D + j] / n; S.cov[(size_t)i * D + j] = v; }
        A[i * NS + j] = v;
    }
    if (tid == 0) suspect = 0;
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NT; ++p) {
        const int c0 = 16 * p;
        // ---- diagonal block: lane = row, right-looking in registers
        if (wv == 0) {
            double a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = A[(c0 + li) * NS + c0 + k];
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double aii = readlane_f64(a[i], i);
                if (!(aii > 0.0)) bad = true;
                const double lii = sqrt(fabs(aii) + (aii > 0.0 ? 0.0 : 1.0));
                const double lji = (li == i) ? lii : a[i] / lii;
                if (li >= i) a[i] = lji;
#pragma unroll
                for (int k = i + 1; k < 16; ++k) { const double lki = readlane_f64(lji, k); if (li > i) a[k] -= lji * lki; }
            }
            if (lk == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) A[(c0 + li) * NS + c0 + k] = (k <= li) ? a[k] : 0.0;
            }
            if (bad && lane == 0) suspect = 1;
        }
        __syncthreads();
        // ---- rows below: x L11^T = A21, a thread per row
        {
            const int i = c0 + 16 + tid;
            if (i < NR) {
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double s = A[i * NS + c0 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= x[k] * A[(c0 + j) * NS + c0 + k];
                    x[j] = s / A[(c0 + j) * NS + c0 + j];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) A[i * NS + c0 + j] = x[j];
            }
        }
        __syncthreads();
        // ---- trailing matrix, lower triangle of tiles, dealt out to the four waves
        int q = 0;
        for (int ti = p + 1; ti < NT; ++ti)
            for (int tj = p + 1; tj <= ti; ++tj, ++q) {
                if ((q & 3) != wv) continue;
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = A[(16 * ti + lk + 4 * r) * NS + 16 * tj + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[(16 * ti + li) * NS + c0 + 4 * ks + lk], A[(16 * tj + li) * NS + c0 + 4 * ks + lk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(16 * ti + lk + 4 * r) * NS + 16 * tj + li] = acc[r];
            }
        __syncthreads();
    }
    // a pivot that is tiny against its own diagonal element is round-off's to decide as well
    for (int i = tid; i < D; i += 256) { const double l = A[i * NS + i]; if (!(l * l > 1e-9 * S.cov[(size_t)i * D + i])) suspect = 1; }
    __syncthreads();
    if (tid == 0) S.ctl->chol_suspect = suspect;
    if (suspect) return;
    for (int e = tid; e < D * D; e += 256) { const int i = e / D, j = e - i * D; S.chol[e] = (j <= i) ? A[i * NS + j] : 0.0; }
}

// pieces of the general update path used by the fused update of pc_update.hip
extern "C" void pc_launch_scan_blocks(int *blk, int nblk, int *total, int *total2, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, blk, nblk, total, total2);
}
extern "C" void pc_launch_chol_only(const PcState *S, const double *ncov, const int *count, hipStream_t st)
{   // one cluster, one "partial" = n times the covariance
    const int D = S->D;
    size_t sh2 = sizeof(double) * (2 * (size_t)D * D + ((size_t)D * D >= PC_CHOL_NT ? 0 : PC_CHOL_NT));
    int a_global = 0;
    if (sh2 > 160 * 1024) { a_global = 1; sh2 = sizeof(double) * ((size_t)D * D + ((size_t)D * D >= PC_CHOL_NT ? 0 : PC_CHOL_NT)); }
    if (sh2 > 160 * 1024) { a_global = 2; sh2 = sizeof(double) * PC_CHOL_NT; }
    pc_need_dyn_lds((const void *)k_cov_final_chol, sh2);
    static const bool blocked_off = std::getenv("PC_CHOL_BLOCKED_OFF") != nullptr;
    int guard = 0;
    if (!blocked_off && D >= 32 && D <= 128) {
        const int nt = (D + 15) / 16;
        const size_t shb = sizeof(double) * (size_t)(16 * nt) * (16 * nt + 1);
#define PC_CHOLB(NT) { pc_need_dyn_lds((const void *)k_chol_blocked<NT>, shb); \
            hipLaunchKernelGGL((k_chol_blocked<NT>), dim3(1), dim3(256), shb, st, *S, ncov, count); }
        switch (nt) { case 2: PC_CHOLB(2) break; case 3: PC_CHOLB(3) break; case 4: PC_CHOLB(4) break; case 5: PC_CHOLB(5) break;
                      case 6: PC_CHOLB(6) break; case 7: PC_CHOLB(7) break; default: PC_CHOLB(8) break; }
#undef PC_CHOLB
        guard = 1;
    }
    hipLaunchKernelGGL(k_cov_final_chol, dim3(1), dim3(PC_CHOL_NT), sh2, st, *S, 1, ncov, count, a_global, guard);
}

extern "C" void pc_launch_init_state(const PcState *S, double logzero, hipStream_t st)
{
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(256), 0, st, *S, logzero);
}

#define PC_POST_BLOCKS 512
extern "C" int pc_post_blocks(void) { return PC_POST_BLOCKS; }
extern "C" void pc_launch_post_moments(const PcState *S, int nd, double *pmax, double *part, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_max, dim3(PC_POST_BLOCKS), dim3(256), 0, st, *S, nd, pmax);
    hipLaunchKernelGGL(k_post_moments, dim3(PC_POST_BLOCKS), dim3(256), 0, st, *S, nd, pmax, part);
}
