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
//   * update_evidence is additive in exp-space: every accumulator travels as a pair (m, s) meaning m + log s, so an
//     accumulation costs one exp and no log, and the O(ncluster) cross terms of a death are one cluster per lane:
//     lane q carries X_q, <Z X_q> and the factors that X_p X_q picked up since the launch began -- the cross-volume
//     matrix itself is only READ during the launch (its row for the next cluster to die is requested one death ahead) and
//     rewritten once at the end;
//   * live_logZ and the update trigger are sums over clusters of quantities that change in the dying and the receiving
//     cluster only: kept per lane in linear space about references fixed at launch, one DPP wave sum per accepted chain;
//   * identify_cluster comes from the candidate lists of k_nn_lists, one baby per lane, the eight candidates' liveness
//     fetched side by side (two LDS round trips);
//   * nothing is stored to HBM inside the loop (on gfx9 a wait for a prefetched load also waits for every store issued
//     before it): plan records, phantom masks and slot sources collect in LDS and leave together.
//
// ONE wavefront runs the loop -- no barrier inside it; the other three waves of the workgroup stage the state before and
// write it back after.  Anything outside this kernel's envelope (dynamic nlive, the reference's list rule of the
// sequential-stream test mode, kill-off, live sets or cluster counts beyond the LDS) stays with k_consume, which is also the
// arbiter: settings.ablate bit 5 sends every launch there, and the two must produce the same run (tests/test_gpu_parity.py).
// A cluster that dies ends the fast loop: the deletion (delete_cluster, run_time_info.f90:507-598) is done by the
// workgroup on the way out and the rest of the nursery fails the epoch guard in the next launch, as in the reference.
#include "pc_state.h"
#include "pc_keys.h"
#include <cstdlib>

#define CL_NT 256
#define CL_J 2                    /* clusters per lane: ncluster <= 128 */
#define CL_MAXC (64 * CL_J)

struct ClLayout {                 // byte offsets into the dynamic LDS block; the same function sizes it on the host
    size_t sL, sortL, candL, cLast, hLogw, hPostXs, hZm, hZs, hContour, zp, logn, fg, masks;
    size_t sortS, sC, sP, sO, sSrc, sCS, candW, candRank, cNlike, cEpoch, cCa, lst, lstOff, hDeadIdx, hDeadSrc, hDeadCuid, hPhCuid, hPhBase, kmin, misc;
    size_t total;
};
__host__ __device__ inline ClLayout cl_layout(int Ncap, int B, int nr)
{
    ClLayout o{};
    size_t p = 0;
    const size_t NS = ((size_t)Ncap + 63) & ~(size_t)63, nw = ((size_t)nr + 63) / 64;
    auto take = [&](size_t &field, size_t bytes) { field = p; p += (bytes + 15) & ~(size_t)15; };
    take(o.sL, 8 * (size_t)Ncap); take(o.sortL, 8 * (NS + 1)); take(o.candL, 8 * ((size_t)B + 1)); take(o.cLast, 8 * (size_t)B);
    take(o.hLogw, 8 * (size_t)B); take(o.hPostXs, 8 * (size_t)B); take(o.hZm, 8 * (size_t)B); take(o.hZs, 8 * (size_t)B); take(o.hContour, 8 * (size_t)B);
    take(o.zp, 8 * 6 * CL_MAXC); take(o.logn, 8 * ((size_t)Ncap + 4)); take(o.fg, 8 * 2 * CL_MAXC); take(o.masks, 8 * (size_t)B * nw);
    take(o.kmin, 8 * CL_MAXC);
    take(o.sortS, 4 * (NS + 1)); take(o.sC, 4 * (size_t)Ncap); take(o.sP, 4 * (size_t)Ncap); take(o.sO, 4 * (size_t)Ncap); take(o.sSrc, 4 * (size_t)Ncap);
    take(o.sCS, 4 * (size_t)B); take(o.candW, 4 * ((size_t)B + 1)); take(o.candRank, 4 * (size_t)B); take(o.cNlike, 4 * (size_t)B); take(o.cEpoch, 4 * (size_t)B);
    take(o.cCa, 4 * (size_t)B); take(o.lst, 4 * ((size_t)Ncap + B)); take(o.lstOff, 4 * (CL_MAXC + 1));
    take(o.hDeadIdx, 4 * (size_t)B); take(o.hDeadSrc, 4 * (size_t)B); take(o.hDeadCuid, 4 * (size_t)B); take(o.hPhCuid, 4 * (size_t)B); take(o.hPhBase, 4 * (size_t)B);
    take(o.misc, 4 * 64);
    o.total = p;
    return o;
}

// (m, s) <- (m, s) (+) (m2, s2): one exp, no log.  Neutral element (NEGBIG, 0).
__device__ __forceinline__ void cl_comb(double &m, double &s, double m2, double s2)
{
    const double e = exp(-fabs(m - m2));
    s = (m >= m2) ? s + s2 * e : s * e + s2;
    m = fmax(m, m2);
}
// three pairs at once: the exponentials do not wait for each other
__device__ __forceinline__ void cl_comb3(double &m, double &s, double m1, double s1, double m2, double s2, bool fix0)
{
    const double M = fix0 ? 0.0 : fmax(m, fmax(m1, m2));
    const double e0 = exp(m - M), e1 = exp(m1 - M), e2 = exp(m2 - M);
    s = s * e0 + s1 * e1 + s2 * e2;
    m = M;
}
__device__ __forceinline__ double cl_val(double m, double s, double logzero)
{   // the logarithm a pair stands for; an accumulator nothing was added to keeps the reference's logzero
    const double v = s > 0.0 ? m + log(s) : NEGBIG;
    return v > logzero ? v : logzero;
}
__device__ __forceinline__ int cl_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double cl_unid(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double cl_get(const double (&v)[CL_J], int c) { return c < 64 ? readlane_f64(v[0], c) : readlane_f64(v[1], c - 64); }
__device__ __forceinline__ int cl_geti(const int (&v)[CL_J], int c) { return c < 64 ? __builtin_amdgcn_readlane(v[0], c) : __builtin_amdgcn_readlane(v[1], c - 64); }

__global__ __launch_bounds__(CL_NT) void k_consume_cl(PcState S)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ncap = S.Ncap, maxc = S.maxc, nr = S.nr, nT = S.nT, D = S.D;
    const int NS = (Ncap + 63) & ~63, nw = (nr + 63) / 64;
    const ClLayout Y = cl_layout(Ncap, S.B, nr);
    double *sL = (double *)(smem + Y.sL), *sortL = (double *)(smem + Y.sortL), *candL = (double *)(smem + Y.candL), *cLast = (double *)(smem + Y.cLast);
    double *hLogw = (double *)(smem + Y.hLogw), *hPostXs = (double *)(smem + Y.hPostXs), *hZm = (double *)(smem + Y.hZm), *hZs = (double *)(smem + Y.hZs);
    double *hContour = (double *)(smem + Y.hContour), *zp = (double *)(smem + Y.zp), *slogn = (double *)(smem + Y.logn), *fg = (double *)(smem + Y.fg);
    unsigned long long *masks = (unsigned long long *)(smem + Y.masks), *kmin = (unsigned long long *)(smem + Y.kmin);
    int *sortS = (int *)(smem + Y.sortS), *sC = (int *)(smem + Y.sC), *sP = (int *)(smem + Y.sP), *sO = (int *)(smem + Y.sO), *sSrc = (int *)(smem + Y.sSrc);
    int *sCS = (int *)(smem + Y.sCS), *candW = (int *)(smem + Y.candW), *candRank = (int *)(smem + Y.candRank), *cNlike = (int *)(smem + Y.cNlike);
    int *cEpoch = (int *)(smem + Y.cEpoch), *cCa = (int *)(smem + Y.cCa), *lst = (int *)(smem + Y.lst), *lstOff = (int *)(smem + Y.lstOff);
    int *hDeadIdx = (int *)(smem + Y.hDeadIdx), *hDeadSrc = (int *)(smem + Y.hDeadSrc), *hPhBase = (int *)(smem + Y.hPhBase), *misc = (int *)(smem + Y.misc);
    unsigned *hDeadCuid = (unsigned *)(smem + Y.hDeadCuid), *hPhCuid = (unsigned *)(smem + Y.hPhCuid);
    // own-state pairs of a cluster (touched only when the cluster itself loses a point): Zp, Zp2, ZpXp as (m, s)
    double *zpm = zp, *zps = zp + CL_MAXC, *zp2m = zp + 2 * CL_MAXC, *zp2s = zp + 3 * CL_MAXC, *zpxm = zp + 4 * CL_MAXC, *zpxs = zp + 5 * CL_MAXC;
    double *Fbuf = fg, *Gbuf = fg + CL_MAXC;

    PcCtl *ctl = S.ctl;
    const int T = ctl->i_nursery;                     // chains in the nursery at launch: w = T-1 ... 0
    int nc = ctl->ncluster;
    const int epoch0 = ctl->admin_epoch;
    // ------------------------------------------------------------------ stage (all waves)
    for (int s = tid; s < Ncap; s += CL_NT) { sL[s] = S.live_logL[s]; sC[s] = S.live_cluster[s]; sP[s] = S.live_pos[s]; sO[s] = S.nn_slot_owner[s]; sSrc[s] = S.slot_src[s]; }
    for (int i = tid; i <= NS; i += CL_NT) { const bool in = i < NS; sortS[i] = in ? S.sort_slot[i] : -1; sortL[i] = in ? key2d(S.sort_key[i]) : PC_HUGE; }
    for (int c = tid; c < S.B; c += CL_NT) sCS[c] = S.nn_chain_slot[c];
    for (int w = tid; w < T; w += CL_NT) { cLast[w] = S.baby_logL[(size_t)w * nr + nr - 1]; cNlike[w] = S.ch_nlike[w]; cEpoch[w] = S.ch_epoch[w]; cCa[w] = S.ch_cluster[w]; }
    for (int k = tid; k < Ncap + 4; k += CL_NT) slogn[k] = S.logn[k];
    for (int c = tid; c < CL_MAXC; c += CL_NT) {
        const bool in = c < nc;
        zpm[c] = in ? S.logZp[c] : NEGBIG; zps[c] = in ? 1.0 : 0.0; zp2m[c] = in ? S.logZp2[c] : NEGBIG; zp2s[c] = in ? 1.0 : 0.0;
        zpxm[c] = in ? S.logZpXp[c] : NEGBIG; zpxs[c] = in ? 1.0 : 0.0;
    }
    __syncthreads();
    // per-cluster lists with room for the chains that may join: a cluster's region = its points + the nursery's chains seeded in it
    if (tid < nc) { int a = 0; for (int w = 0; w < T; ++w) a += (cCa[w] == tid); kmin[tid] = (unsigned long long)a; }
    __syncthreads();
    if (tid == 0) { int o = 0; for (int c = 0; c < nc; ++c) { lstOff[c] = o; o += S.cl_n[c] + (int)kmin[c]; } lstOff[nc] = o; }
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] >= 0) lst[lstOff[sC[s]] + sP[s]] = s;
    // ranks of the candidates (last babies) among themselves: (logL, chain) ascending
    for (int w = tid; w < T; w += CL_NT) {
        const double x = cLast[w];
        int r = 0;
        for (int v = 0; v < T; ++v) { const double y = cLast[v]; r += (y < x) || (y == x && v < w); }
        candRank[w] = r; candL[r] = x; candW[r] = w;
    }
    if (tid == 0) { candL[T] = PC_HUGE; candW[T] = -1; }
    {   // The nursery's records and the cross-volume matrix were written by other XCDs: a first touch costs 1-2 us, and the loop
        // would pay that once per chain, serially.  Touch what it will read now, in bulk, so that its loads hit this XCD's L2.
        auto touch = [&](const void *base, size_t bytes) {
            const char *b = (const char *)base;
            for (size_t o = (size_t)tid * 64; o < bytes; o += (size_t)CL_NT * 64) { const int v = *(const volatile int *)(b + o); asm volatile("" :: "v"(v)); }
        };
        touch(S.baby_logL, sizeof(double) * (size_t)T * nr);
        touch(S.nn_list, sizeof(int) * (size_t)T * nr * PC_NN_K);
        for (int c = 0; c < nc; ++c) touch(S.XpXq + (size_t)c * maxc, sizeof(double) * (size_t)nc);
    }
    __syncthreads();

    // ------------------------------------------------------------------ the loop (wave 0)
    __shared__ int out_i[16];
    __shared__ double out_d[8];
    if (wv == 0) {
        int i_nursery = T, epoch = epoch0, failures = ctl->failures, ndead = ctl->ndead, nph = ctl->nphantom;
        long long nlike = ctl->nlike, niter = ctl->niter, nlike_failed = ctl->nlike_failed;
        double lx_last = ctl->logX_last_update;
        int status = PC_ST_RUNNING, error = PC_ERR_NONE, need_drop = 0;
        const double log2v = 0.6931471805599453, logzero = S.logzero;
        // ---- per-lane cluster state (cluster q = lane + 64 j)
        double Xp[CL_J], XL[CL_J], ZXm[CL_J], ZXs[CL_J], F[CL_J], G[CL_J], lref[CL_J], lsum[CL_J], Eq[CL_J], thr[CL_J], A[CL_J];
        int n[CL_J]; unsigned uid[CL_J];
        double lxm0 = -PC_HUGE, R0 = -PC_HUGE;
#pragma unroll
        for (int j = 0; j < CL_J; ++j) {
            const int q = lane + 64 * j; const bool in = q < nc;
            Xp[j] = in ? S.logXp[q] : -PC_HUGE; ZXm[j] = in ? S.logZXp[q] : NEGBIG; ZXs[j] = in ? 1.0 : 0.0; F[j] = 0.0; G[j] = 0.0;
            n[j] = in ? S.cl_n[q] : 0; lref[j] = in ? S.lse_ref[q] : -PC_HUGE; lsum[j] = in ? S.lse_sum[q] : 0.0; thr[j] = in ? S.death_thr[q] : -PC_HUGE;
            uid[j] = in ? S.cl_uid[q] : 0u;
            lxm0 = fmax(lxm0, Xp[j]); if (in && n[j] > 0) R0 = fmax(R0, lref[j]);
        }
        lxm0 = wave_max(lxm0); R0 = wave_max(R0);
        double sumXL = 0.0, acc = 0.0;
#pragma unroll
        for (int j = 0; j < CL_J; ++j) {
            const bool in = lane + 64 * j < nc;
            XL[j] = in ? exp(Xp[j] - lxm0) : 0.0;
            Eq[j] = (in && n[j] > 0) ? exp(lref[j] - R0) : 0.0;
            A[j] = (in && n[j] > 0) ? (lsum[j] / (double)n[j]) * XL[j] * Eq[j] : 0.0;
        }
        { double a = 0.0, b = 0.0;
#pragma unroll
          for (int j = 0; j < CL_J; ++j) { a += XL[j]; b += A[j]; }
          sumXL = wave_sum<4>(a); acc = wave_sum<4>(b); }
        double Zm = ctl->logZ, Zs = 1.0, Z2m = ctl->logZ2, Z2s = 1.0;
        double E0 = exp(S.log_prec + Zm - lxm0 - R0);
        const double UT = exp(lx_last + S.log_cf - lxm0);          // update trigger: sum_p X_p <= X_last_update * compression_factor
        // ---- death order: the sorted snapshot and the newcomers of this launch
        int ptr = 0;
        double curL = sortL[0]; int curS = sortS[0];
        int curC = (curS >= 0 && curL < PC_HUGE) ? sC[curS] : 0;
        curL = cl_unid(curL); curS = cl_uni(curS); curC = cl_uni(curC);
        unsigned long long accw = 0ull;                // lane l < 16: word l of the bitmap of accepted candidate ranks
        double nmL = PC_HUGE; int nmRank = -1;
        // the row of the cross-volume matrix for the cluster expected to lose a point next
        int predC = curC; double xrow[CL_J];
#pragma unroll
        for (int j = 0; j < CL_J; ++j) { const int q = lane + 64 * j; xrow[j] = (q < nc) ? S.XpXq[(size_t)predC * maxc + q] : 0.0; }
        // ---- the records of the next chain are requested while this one is processed
        double pf_blog = 0.0; int4 pf_a = make_int4(PC_NN_NONE, PC_NN_NONE, PC_NN_NONE, PC_NN_NONE), pf_b = pf_a;
        auto prefetch = [&](int wn) {
            if (wn < 0) return;
            if (lane < nr) {
                pf_blog = S.baby_logL[(size_t)wn * nr + lane];
                const int4 *L4 = (const int4 *)(S.nn_list + ((size_t)wn * nr + lane) * PC_NN_K);
                pf_a = L4[0]; pf_b = L4[1];
            }
        };
        prefetch(T - 1);
        const int seg_hi = T - 1;
        long long cyc0 = clock64(), walks = 0, fallbacks = 0, cyA = 0, cyB = 0, cyC = 0, cyD = 0;

        while (true) {
            // ---- more_samples_needed (nested_sampling.F90:514-543) + the failures guard (:239)
            bool more = true;
            if (S.max_ndead == 0) more = false;
            else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
            else if (S.use_prec) { if (!(acc > 0.0) || acc < Zs * E0) more = false; }
            if (!more || failures > S.nfail) { status = PC_ST_DONE; break; }
            if (i_nursery == 0) break;
            const long long q0 = clock64();
            const int w = i_nursery - 1;
            i_nursery--;
            const double my_blog = pf_blog; const int4 my_a = pf_a, my_b = pf_b;
            prefetch(w - 1);
            const int w_nlike = cNlike[w], w_epoch = cEpoch[w], ca = cl_uni(cCa[w]);
            nlike += w_nlike; niter++;
            if (lane == 0) { hDeadIdx[w] = -1; hPhBase[w] = nph; hContour[w] = logzero; hPhCuid[w] = 0u; }
            if (lane < nw) masks[(size_t)w * nw + lane] = 0ull;
            if (w_epoch != epoch) { nlike_failed += w_nlike; continue; }        // nested_sampling.F90:313
            // ---- replace_point (run_time_info.f90:716-787)
            const double Lg = fmin(curL, nmL);
            if (lane == 0) hContour[w] = Lg;
            int nph_add = 0, id_last = -1;
            for (int m = 0; m < nw; ++m) {
                const int i = m * 64 + lane;
                double bl = my_blog; int4 a = my_a, b = my_b;
                if (m > 0 && i < nr) {
                    bl = S.baby_logL[(size_t)w * nr + i];
                    const int4 *L4 = (const int4 *)(S.nn_list + ((size_t)w * nr + i) * PC_NN_K);
                    a = L4[0]; b = L4[1];
                }
                const bool need = (i < nr) && (bl > Lg);
                int res = -1;
                if (__ballot(need)) {
                    // identify_cluster from the candidate list: the first entry that is alive NOW is the nearest live point
                    const int codes[PC_NN_K] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
                    int sl[PC_NN_K], ow[PC_NN_K], cc[PC_NN_K];
#pragma unroll
                    for (int k = 0; k < PC_NN_K; ++k) {
                        const int code = codes[k];
                        sl[k] = (!need || code == PC_NN_NONE) ? -1 : (code >= 0 ? code : sCS[-(1 + code)]);
                    }
#pragma unroll
                    for (int k = 0; k < PC_NN_K; ++k) { const int s = sl[k] >= 0 ? sl[k] : 0; ow[k] = sO[s]; cc[k] = sC[s]; }
                    res = need ? -2 : -1;
#pragma unroll
                    for (int k = PC_NN_K - 1; k >= 0; --k) {
                        const int code = codes[k];
                        const bool alive = sl[k] >= 0 && (code >= 0 ? (ow[k] == -1) : (ow[k] == -(1 + code)));
                        if (alive) res = cc[k];
                    }
                    walks++;
                    unsigned long long unres = __ballot(res == -2);
                    if (unres) {
                        // a list without a living entry: the full search of identify_cluster (run_time_info.f90:913-949) over the
                        // live set as it is now, one baby at a time, the wave scanning the slots (rare: a handful of chains per run)
                        fallbacks++;
                        while (unres) {
                            const int bl_lane = __ffsll((long long)unres) - 1; unres &= unres - 1;
                            const int bi = m * 64 + bl_lane;
                            const double *x = S.babies + ((size_t)w * nr + bi) * nT;
                            vk_t best{PC_HUGE, 0x7fffffff};
                            for (int s = lane; s < Ncap; s += 64) {
                                const int c = sC[s];
                                if (c < 0) continue;
                                const int o = sO[s];
                                const double *y = (o >= 0) ? S.babies + ((size_t)o * nr + (nr - 1)) * nT : S.live + (size_t)s * nT;
                                double d2 = 0.0;
                                for (int d = 0; d < D; ++d) { const double t = x[d] - y[d]; d2 += t * t; }
                                best = vk_min(best, vk_t{d2, c * Ncap + sP[s]});
                            }
                            best = wave_argmin(best);
                            if (lane == bl_lane) res = (best.k == 0x7fffffff) ? -1 : best.k / Ncap;
                        }
                    }
                }
                // phantoms: babies 1 .. nr-1 above the contour and inside the seed cluster's cell
                const unsigned long long pm = __ballot((i < nr - 1) && need && res == ca);
                if (pm) { if (lane == 0) masks[(size_t)w * nw + m] = pm; nph_add += __popcll(pm); }
                if (m == (nr - 1) / 64) id_last = __builtin_amdgcn_readlane(res, (nr - 1) & 63);
            }
            const long long q1 = clock64(); cyA += q1 - q0;
            if (nph + nph_add > S.Pcap) { status = PC_ST_ERROR; error = PC_ERR_PHANTOM_CAP; break; }
            if (lane == 0) hPhCuid[w] = cl_geti((const int (&)[CL_J])uid, ca);
            nph += nph_add;
            const double Llast = cLast[w];
            bool replaced = false;
            if (Llast > Lg) {
                if (id_last == ca) {
                    if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
                    // ================= delete_outermost_point (run_time_info.f90:789-817): the global minimum dies
                    const bool from_snap = curL <= nmL;
                    int slot_del, cd;
                    if (from_snap) { slot_del = curS; cd = curC; }
                    else { slot_del = cl_uni(sCS[candW[nmRank]]); cd = cl_uni(sC[slot_del]); }
                    const double L = Lg;
                    const int nd = cl_geti(n, cd);
                    const double l0 = slogn[nd], l1 = slogn[nd + 1], l2 = slogn[nd + 2];
                    const double Xd = cl_get(Xp, cd), Fd = cl_get(F, cd), Gd = cl_get(G, cd);
                    if (cd != predC) {                 // (a newcomer died, or the prediction was made before a relabel: fetch the row now)
#pragma unroll
                        for (int j = 0; j < CL_J; ++j) { const int q = lane + 64 * j; xrow[j] = (q < nc) ? S.XpXq[(size_t)cd * maxc + q] : 0.0; }
                    }
                    const double XX = cl_get(xrow, cd) + Gd;                 // log <X_cd^2> now
                    const double logweight = Xd - l1;
                    const double zxm_d = cl_get(ZXm, cd), zxs_d = cl_get(ZXs, cd);
                    const double lref_d = cl_get(lref, cd), lref_a = cl_get(lref, ca);
                    // ---- update_evidence (run_time_info.f90:211-296).  Every accumulation reads the state before the death.
                    // (a) the cross terms, one cluster per lane: <Z X_q> += <X_cd X_q> L / (n+1); the dying cluster's own term
                    //     is scaled by n/(n+1) and gains <X_cd^2> L n / ((n+1)(n+2))
#pragma unroll
                    for (int j = 0; j < CL_J; ++j) {
                        const int q = lane + 64 * j;
                        if (j == 0 || nc > 64) {
                            const bool self = q == cd;
                            const double tm = self ? XX + L + l0 - l1 - l2 : xrow[j] + Fd + F[j] + L - l1;
                            double bm = ZXm[j] + (self ? l0 - l1 : 0.0), bs = ZXs[j];
                            cl_comb(bm, bs, tm, 1.0);
                            if (q < nc) { ZXm[j] = bm; ZXs[j] = bs; }
                        }
                    }
                    // (b) the accumulators of the evidence itself and of the dying cluster, one per lane; lanes 3, 6, 7 take the
                    //     plain exponentials the bookkeeping needs (threshold scale, live log-sum-exp of the two clusters)
                    double jm = NEGBIG, js = 0.0, t1m = NEGBIG, t1s = 0.0, t2m = NEGBIG, t2s = 0.0; bool fix0 = false;
                    const double zp_m = zpm[cd], zp_s = zps[cd], zp2_m = zp2m[cd], zp2_s = zp2s[cd], zpx_m = zpxm[cd], zpx_s = zpxs[cd];
                    const double tZ = Xd + L - l1, tXX = log2v + XX + 2.0 * L - l1 - l2, tS = XX + L + l0 - l1 - l2;
                    if (lane == 0) { jm = Zm; js = Zs; t1m = tZ; t1s = 1.0; }
                    else if (lane == 1) { jm = zp_m; js = zp_s; t1m = tZ; t1s = 1.0; }
                    else if (lane == 2) { jm = Z2m; js = Z2s; t1m = log2v + zxm_d + L - l1; t1s = zxs_d; t2m = tXX; t2s = 1.0; }
                    else if (lane == 3) { jm = S.log_prec + fmax(Zm, tZ) - lxm0 - R0; js = 1.0; fix0 = true; }
                    else if (lane == 4) { jm = zp2_m; js = zp2_s; t1m = log2v + zpx_m + L - l1; t1s = zpx_s; t2m = tXX; t2s = 1.0; }
                    else if (lane == 5) { jm = zpx_m + l0 - l1; js = zpx_s; t1m = tS; t1s = 1.0; }
                    else if (lane == 6) { jm = L - lref_d; js = 1.0; fix0 = true; }
                    else if (lane == 7) { jm = -fabs(Llast - lref_a); js = 1.0; fix0 = true; }
                    if (lane < 8) cl_comb3(jm, js, t1m, t1s, t2m, t2s, fix0);
                    Zm = readlane_f64(jm, 0); Zs = readlane_f64(js, 0); Z2m = readlane_f64(jm, 2); Z2s = readlane_f64(js, 2);
                    E0 = readlane_f64(js, 3);
                    if (lane == 1) { zpm[cd] = jm; zps[cd] = js; }
                    if (lane == 4) { zp2m[cd] = jm; zp2s[cd] = js; }
                    if (lane == 5) { zpxm[cd] = jm; zpxs[cd] = js; }
                    const double e_del = readlane_f64(js, 6), e_add = readlane_f64(js, 7);
                    const long long q2 = clock64(); cyB += q2 - q1;
                    // (E0 was formed with max(Zm, term) as the pair's new scale: that IS Zm now)
                    // ---- the dying cluster: volume, factors of its row of the cross-volume matrix, count, live log-sum-exp
                    const double ratio = (double)nd / ((double)nd + 1.0);
#pragma unroll
                    for (int j = 0; j < CL_J; ++j)
                        if (lane + 64 * j == cd) { Xp[j] += l0 - l1; F[j] += l0 - l1; G[j] += l0 - l2; XL[j] *= ratio; n[j] = nd - 1; lsum[j] -= e_del; thr[j] = L; }
                    // ---- plan record of this death
                    if (lane == 0) {
                        const int src = sSrc[slot_del];
                        hDeadIdx[w] = ndead; hDeadSrc[w] = (src >= 0) ? -(1 + src) : slot_del; hLogw[w] = logweight;
                        hDeadCuid[w] = cl_geti((const int (&)[CL_J])uid, cd);
                        hZm[w] = Zm; hZs[w] = Zs;
                    }
                    ndead++;
                    // ---- the order of deaths moves on
                    if (from_snap) {
                        ptr++;
                        double nl = sortL[ptr]; int ns = sortS[ptr];
                        int ncl = (ns >= 0 && nl < PC_HUGE) ? sC[ns] : 0;
                        curL = cl_unid(nl); curS = cl_uni(ns); curC = cl_uni(ncl);
                    } else {
                        if (lane == (nmRank >> 6)) accw &= ~(1ull << (nmRank & 63));
                        const unsigned long long nz = __ballot(lane < 16 && accw != 0ull);
                        if (nz) {
                            const int l = __ffsll((long long)nz) - 1;
                            const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(accw >> 32), l) << 32) |
                                                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned)accw, l);
                            nmRank = l * 64 + __ffsll((long long)word) - 1; nmL = cl_unid(candL[nmRank]);
                        } else { nmRank = -1; nmL = PC_HUGE; }
                    }
                    // ================= add_point: the newcomer takes the dead point's slot (static number of live points)
                    const int slot = slot_del;
                    const int na = cl_geti(n, ca);                               // (after the death: cd may be ca)
                    const double lref_a2 = cl_get(lref, ca), lsum_a = cl_get(lsum, ca);
                    double nref = lref_a2, nsum;
                    if (na == 0) { nref = Llast; nsum = 1.0; }
                    else if (Llast > lref_a2) { nsum = lsum_a * e_add + 1.0; nref = Llast; }
                    else nsum = lsum_a + e_add;
                    double nEq = cl_get(Eq, ca);
                    if (nref != lref_a2 || na == 0) {                           // the cluster's reference moved: its scale factor too
                        if (nref - R0 > 600.0) {                                 // (cannot be represented about the launch's reference: re-base every cluster)
                            const double R1 = nref;
#pragma unroll
                            for (int j = 0; j < CL_J; ++j) Eq[j] = (lane + 64 * j < nc && n[j] > 0) ? exp(lref[j] - R1) : 0.0;
                            E0 = exp(S.log_prec + Zm - lxm0 - R1);
                            R0 = R1;
                        }
                        nEq = exp(nref - R0);
                    }
#pragma unroll
                    for (int j = 0; j < CL_J; ++j) if (lane + 64 * j == ca) { n[j] = na + 1; lref[j] = nref; lsum[j] = nsum; Eq[j] = nEq; }
                    if (lane == 0) {
                        // list bookkeeping.  Engine rule (oracle keyed mode): a newcomer that replaces a death of its own cluster takes
                        // the dead point's position -- nothing moves.  Otherwise the dying cluster's last entry fills the hole
                        // (delete_point, array_utils.f90:433-458) and the newcomer is appended to its own cluster's list
                        if (cd != ca) {
                            const int p = sP[slot_del], od = lstOff[cd], oa = lstOff[ca];
                            const int last = lst[od + nd - 1];
                            if (p != nd - 1) { lst[od + p] = last; sP[last] = p; }
                            lst[oa + na] = slot; sP[slot] = na;
                        }
                        sL[slot] = Llast; sC[slot] = ca; sO[slot] = w; sCS[w] = slot; sSrc[slot] = w;
                    }
                    { const int r = candRank[w];
                      if (lane == (r >> 6)) accw |= 1ull << (r & 63);
                      if (Llast < nmL) { nmL = Llast; nmRank = r; } }
                    // ---- sums over the clusters: volumes (update trigger, posterior stack) and the live evidence (termination)
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int j = 0; j < CL_J; ++j) {
                        A[j] = (n[j] > 0) ? (lsum[j] / (double)n[j]) * XL[j] * Eq[j] : 0.0;
                        a += XL[j]; b += A[j];
                    }
                    sumXL = wave_sum<4>(a); acc = wave_sum<4>(b);
                    if (lane == 0) hPostXs[w] = sumXL;
                    // the row of the cluster that is expected to lose the next point
                    predC = (curL <= nmL) ? curC : -1;
                    if (predC >= 0) {
#pragma unroll
                        for (int j = 0; j < CL_J; ++j) { const int q = lane + 64 * j; xrow[j] = (q < nc) ? S.XpXq[(size_t)predC * maxc + q] : 0.0; }
                    }
                    replaced = true;
                    cyC += clock64() - q2;
                    if (nd - 1 == 0 && cd != ca) need_drop = 1;                 // a cluster died (delete_cluster): the workgroup takes over
                }
            } else {
                // failed spawn: the last baby is recorded as dead with zero weight (run_time_info.f90:781-785)
                if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
                if (lane == 0) { hDeadIdx[w] = ndead; hDeadSrc[w] = -(1 + w); hLogw[w] = logzero; hPostXs[w] = 1.0; hZm[w] = 0.0; hZs[w] = -1.0; hDeadCuid[w] = 0xFFFFFFFFu; }
                ndead++;
            }
            failures = replaced ? 0 : failures + 1;
            if (!replaced) nlike_failed += w_nlike;
            // ---- update trigger (nested_sampling.F90:321), delete_cluster (:339)
            const bool update = sumXL <= UT;
            if (update) lx_last = lxm0 + log(sumXL);
            if (need_drop) { if (update) status = PC_ST_UPDATE; break; }
            if (update) { status = PC_ST_UPDATE; break; }
        }
        if (need_drop && status == PC_ST_RUNNING) {
            // A cluster died: the administrator's epoch moves on (nested_sampling.F90:339-341) and what is left of the nursery
            // fails the guard of :313 -- after the loop's termination test, which sees the same state before each of them
            bool more = true;
            if (S.max_ndead == 0) more = false;
            else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
            else if (S.use_prec) { if (!(acc > 0.0) || acc < Zs * E0) more = false; }
            if (!more || failures > S.nfail) status = PC_ST_DONE;
            else {
                double nl = 0.0;
                for (int w = lane; w < i_nursery; w += 64) {
                    nl += (double)cNlike[w];
                    hDeadIdx[w] = -1; hPhBase[w] = nph; hContour[w] = logzero; hPhCuid[w] = 0u;
                    for (int m = 0; m < nw; ++m) masks[(size_t)w * nw + m] = 0ull;
                }
                const long long tot = (long long)wave_sum<4>(nl);                 // (exact: counts far below 2^53)
                nlike += tot; nlike_failed += tot; niter += i_nursery;
                i_nursery = 0;
            }
        }
        // ---- hand the state to the workgroup
        if (lane == 0) {
            out_i[0] = status; out_i[1] = error; out_i[2] = i_nursery; out_i[3] = epoch; out_i[4] = failures; out_i[5] = ndead; out_i[6] = nph;
            out_i[7] = need_drop; out_i[8] = seg_hi;
            out_d[0] = cl_val(Zm, Zs, logzero); out_d[1] = cl_val(Z2m, Z2s, logzero); out_d[2] = lx_last;
            { const double v = acc > 0.0 ? log(acc) + lxm0 + R0 : logzero; out_d[3] = (v > logzero + 800.0) ? v : pc_logaddexp(logzero, v); }
            out_d[4] = lxm0;
            ctl->nlike = nlike; ctl->niter = niter; ctl->nlike_failed = nlike_failed;
            ctl->gen_cyc[0] += clock64() - cyc0; ctl->gen_cyc[1] += cyA; ctl->gen_cyc[2] += cyB; ctl->gen_cyc[3] += cyC; ctl->nn_walks += walks; ctl->nn_fallbacks += fallbacks;
            (void)cyD;
        }
#pragma unroll
        for (int j = 0; j < CL_J; ++j) {
            const int q = lane + 64 * j;
            if (q < nc) {
                S.logXp[q] = Xp[j]; S.logZXp[q] = cl_val(ZXm[j], ZXs[j], logzero); S.cl_n[q] = n[j]; S.lse_ref[q] = lref[j]; S.lse_sum[q] = lsum[j];
                S.death_thr[q] = thr[j]; Fbuf[q] = F[j]; Gbuf[q] = G[j];
            }
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ write back (all waves)
    int status = out_i[0];
    const int i_nursery = out_i[2], need_drop = out_i[7], seg_hi = out_i[8];
    int epoch = out_i[3];
    const double lxm0 = out_d[4];
    for (int s = tid; s < Ncap; s += CL_NT) {
        S.live_logL[s] = sL[s]; S.live_cluster[s] = sC[s]; S.live_pos[s] = sP[s]; S.nn_slot_owner[s] = sO[s]; S.slot_src[s] = sSrc[s];
        if (sC[s] >= 0) S.cl_list[(size_t)sC[s] * Ncap + sP[s]] = s;
    }
    for (int c = tid; c < S.B; c += CL_NT) S.nn_chain_slot[c] = sCS[c];
    for (int c = tid; c < nc; c += CL_NT) { S.logZp[c] = cl_val(zpm[c], zps[c], S.logzero); S.logZp2[c] = cl_val(zp2m[c], zp2s[c], S.logzero); S.logZpXp[c] = cl_val(zpxm[c], zpxs[c], S.logzero); }
    // the cross-volume matrix picks up the factors of the launch's deaths: X_p X_q *= f_p f_q, X_p^2 *= g_p
    for (int e = tid; e < nc * nc; e += CL_NT) {
        const int p = e / nc, q = e % nc;
        const double add = (p == q) ? Gbuf[p] : Fbuf[p] + Fbuf[q];
        if (add != 0.0) S.XpXq[(size_t)p * maxc + q] += add;
    }
    // plan records of the chains this launch consumed
    for (int w = i_nursery + tid; w <= seg_hi; w += CL_NT) {
        PcPlanHead h;
        h.dead_idx = hDeadIdx[w]; h.dead_src = hDeadSrc[w]; h.ph_base = hPhBase[w]; h.dead_cuid = hDeadCuid[w]; h.ph_cuid = hPhCuid[w]; h.ph_count = 0;
        h.logw = hLogw[w]; h.contour = hContour[w];
        const bool spawn_failed = hZs[w] < 0.0;
        h.postX = spawn_failed ? 0.0 : lxm0; h.postXs = hPostXs[w];
        h.postZ = spawn_failed ? 0.0 : cl_val(hZm[w], hZs[w], S.logzero);
        if (h.dead_idx < 0) { h.dead_src = 0; h.dead_cuid = 0u; h.logw = 0.0; h.postX = 0.0; h.postXs = 1.0; h.postZ = 0.0; }
        *(PcPlanHead *)&S.plan[w] = h;
        for (int m = 0; m < nw; ++m) S.plan[w].ph_mask[m] = masks[(size_t)w * nw + m];
    }
    // find_min_loglikelihoods (run_time_info.f90:883-909), once: lowest (logL, list position) of every cluster
    for (int c = tid; c < CL_MAXC; c += CL_NT) { kmin[c] = KEY_HUGE; lstOff[c] = 0x7fffffff; }
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] >= 0) atomicMin(&kmin[sC[s]], d2key(sL[s]));
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] >= 0 && d2key(sL[s]) == kmin[sC[s]]) atomicMin(&lstOff[sC[s]], sP[s]);
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] >= 0 && d2key(sL[s]) == kmin[sC[s]] && sP[s] == lstOff[sC[s]]) { S.imin_slot[sC[s]] = s; S.logLp[sC[s]] = sL[s]; }
    for (int c = tid; c < nc; c += CL_NT) if (kmin[c] == KEY_HUGE) { S.imin_slot[c] = -1; S.logLp[c] = PC_HUGE; }
    __syncthreads();
    int ncd = ctl->ncluster_dead, cluster_deleted = 0;
    if (need_drop) {
        // delete_cluster (run_time_info.f90:507-598): drop the first empty cluster, keep the others' order
        __threadfence_block();
        __syncthreads();
        int p = -1;
        for (int c = 0; c < nc && p < 0; ++c) if (S.cl_n[c] == 0) p = c;
        if (p >= 0) {
            if (tid == 0 && ncd < S.maxc_dead) { S.logZp_dead[ncd] = S.logZp[p]; S.logZp2_dead[ncd] = S.logZp2[p]; S.cl_uid_dead[ncd] = S.cl_uid[p]; }
            __syncthreads();
            ncd++;
            if (tid == 0) {
                for (int a = 0, na = 0; a < nc; ++a) {
                    if (a == p) continue;
                    for (int b = 0, nb = 0; b < nc; ++b) { if (b == p) continue; S.XpXq[(size_t)na * maxc + nb] = S.XpXq[(size_t)a * maxc + b]; nb++; }
                    na++;
                }
                for (int c = p; c < nc - 1; ++c) {
                    S.logLp[c] = S.logLp[c + 1]; S.logXp[c] = S.logXp[c + 1]; S.logZp[c] = S.logZp[c + 1]; S.logZXp[c] = S.logZXp[c + 1];
                    S.logZp2[c] = S.logZp2[c + 1]; S.logZpXp[c] = S.logZpXp[c + 1]; S.lse_ref[c] = S.lse_ref[c + 1]; S.lse_sum[c] = S.lse_sum[c + 1];
                    S.death_thr[c] = S.death_thr[c + 1]; S.cl_n[c] = S.cl_n[c + 1]; S.imin_slot[c] = S.imin_slot[c + 1]; S.cl_uid[c] = S.cl_uid[c + 1];
                }
            }
            const int DD = D * D;
            for (int c = p; c < nc - 1; ++c) {
                for (int e = tid; e < DD; e += CL_NT) { S.chol[(size_t)c * DD + e] = S.chol[(size_t)(c + 1) * DD + e]; S.cov[(size_t)c * DD + e] = S.cov[(size_t)(c + 1) * DD + e]; }
                __syncthreads();
            }
            for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] > p) { sC[s] -= 1; S.live_cluster[s] = sC[s]; }
            __syncthreads();
            // (the lists of the clusters behind the deleted one move up a row)
            for (int s = tid; s < Ncap; s += CL_NT) if (sC[s] >= p) S.cl_list[(size_t)sC[s] * Ncap + sP[s]] = s;
            nc--; epoch++; cluster_deleted = 1;
            if (nc == 0) status = PC_ST_DONE;
        }
    }
    if (tid == 0) {
        ctl->status = status; ctl->error = out_i[1]; ctl->i_nursery = i_nursery; ctl->admin_epoch = epoch; ctl->failures = out_i[4];
        ctl->ncluster = nc; ctl->ncluster_dead = ncd; ctl->ndead = out_i[5]; ctl->nphantom = out_i[6];
        ctl->seg_hi = seg_hi; ctl->seg_lo = i_nursery; ctl->cluster_deleted = cluster_deleted;
        ctl->logZ = out_d[0]; ctl->logZ2 = out_d[1]; ctl->logX_last_update = out_d[2]; ctl->live_logZ = out_d[3];
    }
    __syncthreads();
    pc_publish_ctl(S);
}

extern "C" int pc_consume_cl_fits(const PcState *S, int nc)
{
    if (nc < 2 || nc > CL_MAXC || S->B > 1024 || S->nr > 64 * PC_MASK_WORDS) return 0;
    return cl_layout(S->Ncap, S->B, S->nr).total + 512 <= (size_t)160 * 1024;
}

extern "C" int pc_launch_consume_cl(const PcState *S, hipStream_t st)
{
    const size_t sh = cl_layout(S->Ncap, S->B, S->nr).total;
    static size_t done = 0;
    if (sh > done) { (void)hipFuncSetAttribute((const void *)k_consume_cl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); done = sh; }
    hipLaunchKernelGGL(k_consume_cl, dim3(1), dim3(CL_NT), sh, st, *S);
    return 0;
}
