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

struct ClLayout {                 // byte offsets into the dynamic LDS block; the same function sizes it on the host
    size_t slot, sL, sorted, cand, chain, head, own, logn, rcp, fg, masks, kmin, sCS, lst, lstOff, tag, total;
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

template <int J>
__global__ __launch_bounds__(CL_NT) void k_consume_cl(PcState S)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ncap = S.Ncap, maxc = S.maxc, nr = S.nr, nT = S.nT, D = S.D;
    const int NS = (Ncap + 63) & ~63, nw = (nr + 63) / 64;
    const ClLayout Y = cl_layout(Ncap, S.B, nr);
    ClSlot *sS = (ClSlot *)(smem + Y.slot); double *sL = (double *)(smem + Y.sL);
    ClSorted *sSort = (ClSorted *)(smem + Y.sorted); ClCand *sCand = (ClCand *)(smem + Y.cand);
    ClChain *sCh = (ClChain *)(smem + Y.chain); ClHead *sHead = (ClHead *)(smem + Y.head); ClOwn *sOwn = (ClOwn *)(smem + Y.own);
    double *slogn = (double *)(smem + Y.logn), *srcp = (double *)(smem + Y.rcp), *fg = (double *)(smem + Y.fg);
    unsigned long long *masks = (unsigned long long *)(smem + Y.masks), *kmin = (unsigned long long *)(smem + Y.kmin);
    int *sCS = (int *)(smem + Y.sCS), *lst = (int *)(smem + Y.lst), *lstOff = (int *)(smem + Y.lstOff), *tag = (int *)(smem + Y.tag);
    double *Fbuf = fg, *Gbuf = fg + CL_MAXC;
    __shared__ int out_i[16];
    __shared__ double out_d[8];
    __shared__ double ref_d[8];        // launch-wide references: Lhi, lxm0

    PcCtl *ctl = S.ctl;
    const long long t_start = clock64();
    const int T = ctl->i_nursery;                     // chains in the nursery at launch: w = T-1 ... 0
    int nc = ctl->ncluster;
    const int epoch0 = ctl->admin_epoch;
    // ------------------------------------------------------------------ stage (all waves)
    for (int s = tid; s < Ncap; s += CL_NT) { sL[s] = S.live_logL[s]; sS[s] = ClSlot{S.live_cluster[s], S.live_pos[s], S.nn_slot_owner[s], S.slot_src[s]}; }
    for (int i = tid; i <= NS; i += CL_NT) { const bool in = i < NS; sSort[i] = ClSorted{in ? key2d(S.sort_key[i]) : PC_HUGE, 0.0, in ? S.sort_slot[i] : -1, 0}; }
    for (int c = tid; c < S.B; c += CL_NT) sCS[c] = S.nn_chain_slot[c];
    for (int w = tid; w < T; w += CL_NT) sCh[w] = ClChain{S.baby_logL[(size_t)w * nr + nr - 1], S.ch_nlike[w], S.ch_epoch[w], S.ch_cluster[w], 0};
    for (int k = tid; k < Ncap + 4; k += CL_NT) { slogn[k] = S.logn[k]; srcp[k] = k > 0 ? 1.0 / (double)k : 0.0; }
    __syncthreads();
    // per-cluster lists with room for the chains that may join: a cluster's region = its points + the nursery's chains seeded in it
    if (tid < nc) { int a = 0; for (int w = 0; w < T; ++w) a += (sCh[w].ca == tid); kmin[tid] = (unsigned long long)a; }
    if (tid == CL_NT - 1) {
        // the window of this launch: its k-th death lies below the k-th smallest point of the snapshot
        int nlv = 0;
        for (int c = 0; c < nc; ++c) nlv += S.cl_n[c];
        const int kth = (T < nlv ? T : nlv) - 1;
        const double L0 = sSort[0].L, Lk = sSort[kth > 0 ? kth : 0].L;
        ref_d[0] = fmin(Lk, L0 + CL_WINDOW);
        double m = -PC_HUGE;
        for (int c = 0; c < nc; ++c) m = fmax(m, S.logXp[c]);
        ref_d[1] = m;
    }
    __syncthreads();
    const double Lhi = ref_d[0], lxm0 = ref_d[1];
    if (tid == 0) { int o = 0; for (int c = 0; c < nc; ++c) { lstOff[c] = o; o += S.cl_n[c] + (int)kmin[c]; } lstOff[nc] = o; }
    if (tid < nc) {
        // own accumulators of a cluster: each about the larger of its value now and the largest term this launch can add
        ClOwn o;
        const double zp0 = S.logZp[tid], zpx0 = S.logZpXp[tid], zp20 = S.logZp2[tid];
        o.rzp = fmax(zp0, lxm0 + Lhi); o.zp = exp(zp0 - o.rzp); o.kzp = exp(lxm0 + Lhi - o.rzp);
        o.rzpx = fmax(zpx0, 2.0 * lxm0 + Lhi); o.zpx = exp(zpx0 - o.rzpx); o.kzpx = exp(2.0 * lxm0 + Lhi - o.rzpx);
        o.rzp2 = fmax(zp20, fmax(o.rzpx + Lhi, 2.0 * lxm0 + 2.0 * Lhi)); o.zp2 = exp(zp20 - o.rzp2);
        o.kp2a = exp(o.rzpx + Lhi - o.rzp2); o.kp2b = exp(2.0 * lxm0 + 2.0 * Lhi - o.rzp2);
        o.touched = 0; o.pad = 0;
        sOwn[tid] = o;
    }
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sS[s].c >= 0) lst[lstOff[sS[s].c] + sS[s].p] = s;
    // ranks of the candidates (last babies) among themselves: (logL, chain) ascending
    // (by integer key: a strict total order whatever the values -- a NaN logL, which a singular covariance can produce, must not
    //  make two candidates share a rank)
    for (int w = tid; w < T; w += CL_NT) {
        const double x = sCh[w].last;
        const unsigned long long kx = d2key(x);
        int r = 0;
        for (int v = 0; v < T; ++v) { const unsigned long long ky = d2key(sCh[v].last); r += (ky < kx) || (ky == kx && v < w); }
        sCh[w].rank = r; sCand[r] = ClCand{x, exp(x - Lhi), w, 0};
    }
    if (tid == 0) sCand[T] = ClCand{PC_HUGE, 0.0, -1, 0};
    // the exponential of every death this launch can make (its first T snapshot points), and the liveness tags of the candidate
    // lists: entry s < Ncap = cluster of the point that occupied slot s when the lists were made, while it lives; entry
    // Ncap + w = cluster of the last baby of chain w, from its acceptance to its death; -1 otherwise (the last entry: no candidate)
    for (int i = tid; i <= T && i <= NS; i += CL_NT) sSort[i].e = exp(sSort[i].L - Lhi);
    for (int s = tid; s < Ncap; s += CL_NT) tag[s] = (sS[s].o == -1) ? sS[s].c : -1;
    for (int c = tid; c <= S.B; c += CL_NT) { const int sl = c < S.B ? sCS[c] : -1; tag[Ncap + c] = (sl >= 0 && sS[sl].o == c) ? sS[sl].c : -1; }
    {   // The nursery's records and the cross-volume matrix were written by other XCDs: a first touch costs 1-2 us, and the loop
        // would pay that once per chain, serially.  Touch what it will read now, in bulk, so that its loads hit this XCD's L2.
        auto touch = [&](const void *base, size_t bytes) {           // (eight lines in flight per thread; nothing waits until the end)
            const char *b = (const char *)base;
            int acc = 0;
            for (size_t o0 = (size_t)tid * 64; o0 < bytes; o0 += (size_t)CL_NT * 64 * 8) {
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const size_t o = o0 + (size_t)u * CL_NT * 64; v[u] = (o < bytes) ? *(const volatile int *)(b + o) : 0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
            }
            asm volatile("" :: "v"(acc));
        };
        touch(S.baby_logL, sizeof(double) * (size_t)T * nr);
        touch(S.nn_list, sizeof(int) * (size_t)T * nr * PC_NN_K);
        touch(S.XpXq, sizeof(double) * (size_t)nc * maxc);            // (one sweep: a call per row would wait for each row's miss in turn)
    }
    __syncthreads();

    // ------------------------------------------------------------------ the loop (wave 0)
    if (wv == 0) {
        int i_nursery = T, failures = ctl->failures, ndead = ctl->ndead, nph = ctl->nphantom;
        const int epoch = epoch0;
        long long nlike = ctl->nlike, niter = ctl->niter, nlike_failed = ctl->nlike_failed;
        double lx_last = ctl->logX_last_update;
        int status = PC_ST_RUNNING, error = PC_ERR_NONE, need_drop = 0, any_death = 0;
        const double logzero = S.logzero, logZ0 = ctl->logZ, logZ20 = ctl->logZ2;
        // ---- per-lane cluster state (cluster q = lane + 64 j): log volume, linear volume / factors / <Z X_q>, live log-sum-exp
        double Xp[J], XL[J], Fl[J], Gl[J], Flog[J], Glog[J], ZX[J], kZX[J], rZX[J], k2a[J], lref[J], lsum[J], Eq[J], kd[J], thr[J];
        int n[J]; unsigned uid[J];
        double R0 = -PC_HUGE, rZXmax = -PC_HUGE;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int q = lane + 64 * j; const bool in = q < nc;
            Xp[j] = in ? S.logXp[q] : -PC_HUGE; Fl[j] = 1.0; Gl[j] = 1.0; Flog[j] = 0.0; Glog[j] = 0.0;
            n[j] = in ? S.cl_n[q] : 0; lref[j] = in ? S.lse_ref[q] : -PC_HUGE; lsum[j] = in ? S.lse_sum[q] : 0.0; thr[j] = in ? S.death_thr[q] : -PC_HUGE;
            uid[j] = in ? S.cl_uid[q] : 0u;
            const double zx0 = in ? S.logZXp[q] : NEGBIG;
            rZX[j] = fmax(zx0, 2.0 * lxm0 + Lhi); ZX[j] = in ? exp(zx0 - rZX[j]) : 0.0; kZX[j] = exp(2.0 * lxm0 + Lhi - rZX[j]);
            if (in) rZXmax = fmax(rZXmax, rZX[j]);
            if (in && n[j] > 0) R0 = fmax(R0, lref[j]);
        }
        R0 = wave_max(R0); rZXmax = wave_max(rZXmax);
        const double rZ = fmax(logZ0, lxm0 + Lhi), kZ = exp(lxm0 + Lhi - rZ);
        const double rZ2 = fmax(logZ20, fmax(rZXmax + Lhi, 2.0 * lxm0 + 2.0 * Lhi)), k2b = exp(2.0 * lxm0 + 2.0 * Lhi - rZ2);
        double Zl = exp(logZ0 - rZ), Z2l = exp(logZ20 - rZ2);
        double sumXL = 0.0, acc = 0.0;
        {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const bool in = lane + 64 * j < nc;
                XL[j] = in ? exp(Xp[j] - lxm0) : 0.0;
                k2a[j] = exp(rZX[j] + Lhi - rZ2);
                Eq[j] = (in && n[j] > 0) ? exp(lref[j] - R0) : 0.0;
                kd[j] = in ? exp(Lhi - lref[j]) : 0.0;
                a += XL[j]; b += (in && n[j] > 0) ? (lsum[j] * srcp[n[j]]) * XL[j] * Eq[j] : 0.0;
            }
            sumXL = wave_sum<4>(a); acc = wave_sum<4>(b);
        }
        double E0 = exp(S.log_prec + rZ - lxm0 - R0);              // more_samples_needed in linear space: acc < Zl * E0
        double kR = exp(Lhi - R0);
        // (the launch's constants live in vector registers: the loop has more wave-uniform values than scalar registers, and a
        //  spilled scalar costs a v_readlane and its hazard slots at every use)
        double kZv = kZ, k2bv = k2b, UTv = exp(lx_last + S.log_cf - lxm0), Lhiv = Lhi, lxm0v = lxm0, lprec = S.log_prec, rZv = rZ;
        asm volatile("" : "+v"(kZv), "+v"(k2bv), "+v"(UTv), "+v"(Lhiv), "+v"(lxm0v), "+v"(lprec), "+v"(rZv));
        asm volatile("" : "+v"(E0), "+v"(kR), "+v"(R0));
        // update trigger: sum_p X_p <= X_last_update * compression_factor (UTv)
        // ---- death order: the sorted snapshot and the newcomers of this launch
        int ptr = 0;
        double curL; int curS, curC; double eCur;
        { const ClSorted e = sSort[0]; curL = cl_unid(e.L); curS = cl_uni(e.slot); curC = cl_uni((e.slot >= 0 && e.L < PC_HUGE) ? sS[e.slot >= 0 ? e.slot : 0].c : 0); eCur = cl_unid(e.e); }
        unsigned long long accw = 0ull;                // lane l < 16: word l of the bitmap of accepted candidate ranks
        double nmL = PC_HUGE, eNm = 0.0; int nmRank = -1;
        // the row of the cross-volume matrix for the cluster expected to lose a point next, in linear space about X_max^2
        int predC = curC; double xlin[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { const int q = lane + 64 * j; const double xr = (q < nc) ? S.XpXq[(size_t)predC * maxc + q] : NEGBIG; xlin[j] = exp(xr - 2.0 * lxm0); }
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
        long long cyc0 = clock64(), walks = 0, fallbacks = 0;
        if (lane == 0) ctl->gen_cyc[1] += cyc0 - t_start;

        while (true) {
            // ---- more_samples_needed (nested_sampling.F90:514-543) + the failures guard (:239)
            bool more = true;
            if (S.max_ndead == 0) more = false;
            else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
            else if (S.use_prec) { if (!(acc > 0.0) || acc < Zl * E0) more = false; }
            if (!more || failures > S.nfail) {
                status = PC_ST_DONE;
                break;
            }
            if (i_nursery == 0) break;
            const double Lg = fmin(curL, nmL);
            if (Lg > Lhiv) break;                                   // the next death leaves the launch's window: new references (host relaunches)
            const int w = i_nursery - 1;
            i_nursery--;
            const double my_blog = pf_blog; const int4 my_a = pf_a, my_b = pf_b;
            prefetch(w - 1);
            const ClChain ch = sCh[w];
            const int w_nlike = ch.nlike, ca = cl_uni(ch.ca);
            const double Llast = ch.last;
            nlike += w_nlike; niter++;
            if (lane < nw) masks[(size_t)w * nw + lane] = 0ull;
            if (ch.epoch != epoch) {                                // nested_sampling.F90:313
                nlike_failed += w_nlike;
                if (lane == 0) { ClHead h{}; h.dead_idx = -1; h.ph_base = nph; h.contour = logzero; sHead[w] = h; }
                continue;
            }
            // ---- replace_point (run_time_info.f90:716-787)
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
                int res;
                {
                    // identify_cluster from the candidate list: the first entry that is alive NOW is the nearest live point
                    const int codes[PC_NN_K] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
                    int tg[PC_NN_K];
#pragma unroll
                    for (int k = 0; k < PC_NN_K; ++k) {
                        const int code = codes[k];
                        tg[k] = tag[code == PC_NN_NONE ? Ncap + S.B : (code >= 0 ? code : Ncap - 1 - code)];
                    }
                    res = -2;
#pragma unroll
                    for (int k = PC_NN_K - 1; k >= 0; --k) res = tg[k] >= 0 ? tg[k] : res;
                    if (!need) res = -1;
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
                                const ClSlot r = sS[s];
                                if (r.c < 0) continue;
                                const double *y = (r.o >= 0) ? S.babies + ((size_t)r.o * nr + (nr - 1)) * nT : S.live + (size_t)s * nT;
                                double d2 = 0.0;
                                for (int d = 0; d < D; ++d) { const double t = x[d] - y[d]; d2 += t * t; }
                                best = vk_min(best, vk_t{d2, r.c * Ncap + r.p});
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
            if (nph + nph_add > S.Pcap) { status = PC_ST_ERROR; error = PC_ERR_PHANTOM_CAP; break; }
            ClHead hd{};
            hd.dead_idx = -1; hd.ph_base = nph; hd.contour = Lg; hd.ph_cuid = (unsigned)cl_geti<J>((const int (&)[J])uid, ca);
            nph += nph_add;
            bool replaced = false;
            if (Llast > Lg) {
                if (id_last == ca) {
                    if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
                    // ================= delete_outermost_point (run_time_info.f90:789-817): the global minimum dies
                    const bool from_snap = curL <= nmL;
                    int slot_del, cd;
                    if (from_snap) { slot_del = curS; cd = curC; }
                    else { slot_del = cl_uni(sCS[sCand[nmRank].w]); cd = cl_uni(sS[slot_del].c); }
                    const double L = Lg, eL = from_snap ? eCur : eNm;              // exp(L - Lhi), formed when the point became the next to die
                    const int nd = cl_geti<J>(n, cd);
                    const double l0 = slogn[nd], l1 = slogn[nd + 1], l2 = slogn[nd + 2], r1 = srcp[nd + 1], r2 = srcp[nd + 2];
                    if (cd != predC) {                 // (a newcomer died where a snapshot point was expected: fetch the row now)
#pragma unroll
                        for (int j = 0; j < J; ++j) { const int q = lane + 64 * j; const double xr = (q < nc) ? S.XpXq[(size_t)cd * maxc + q] : NEGBIG; xlin[j] = exp(xr - 2.0 * lxm0v); }
                    }
                    const ClOwn own = sOwn[cd];
                    const double XLd = cl_get<J>(XL, cd), Fd = cl_get<J>(Fl, cd), Gd = cl_get<J>(Gl, cd), ZXd = cl_get<J>(ZX, cd), k2ad = cl_get<J>(k2a, cd);
                    const double Xd = cl_get<J>(Xp, cd), kdd = cl_get<J>(kd, cd);
                    const double rat = (double)nd * r1, c1 = eL * r1, c2 = eL * rat * r2;
                    const double XXs = cl_get<J>(xlin, cd) * Gd;                  // <X_cd^2> now, about X_max^2
                    const double logweight = Xd - l1;
                    // ---- update_evidence (run_time_info.f90:211-296); every accumulation reads the state before the death
                    const double tZ = XLd * c1;
                    const double tXX = XXs * (eL * c1 * r2);                       // <X^2> L^2 / ((n+1)(n+2))
                    Zl += tZ * kZv;                                                // <Z> += X L / (n+1)
                    Z2l += 2.0 * (ZXd * c1 * k2ad + tXX * k2bv);                    // <Z^2> += 2 <Z X> L/(n+1) + 2 <X^2> L^2/((n+1)(n+2))
#pragma unroll
                    for (int j = 0; j < J; ++j) {                                  // <Z X_q>, one cluster per lane
                        const bool self = lane + 64 * j == cd;
                        const double cross = xlin[j] * Fd * Fl[j] * c1 * kZX[j];
                        ZX[j] = self ? ZX[j] * rat + XXs * c2 * kZX[j] : ZX[j] + cross;
                    }
                    if (lane == 0) {                                               // the dying cluster's own accumulators
                        ClOwn o2 = own;
                        o2.zp = own.zp + tZ * own.kzp;
                        o2.zp2 = own.zp2 + 2.0 * (own.zpx * c1 * own.kp2a + tXX * own.kp2b);
                        o2.zpx = own.zpx * rat + XXs * c2 * own.kzpx;
                        o2.touched = 1;
                        sOwn[cd] = o2;
                    }
                    const double e_del = eL * kdd;                                 // exp(L - lse_ref[cd])
                    // ---- the dying cluster: volume, the factors of its row of the cross-volume matrix, count, live log-sum-exp
#pragma unroll
                    for (int j = 0; j < J; ++j)
                        if (lane + 64 * j == cd) { Xp[j] += l0 - l1; Flog[j] += l0 - l1; Glog[j] += l0 - l2; XL[j] *= rat; Fl[j] *= rat; Gl[j] *= (double)nd * r2;
                                                   n[j] = nd - 1; lsum[j] -= e_del; thr[j] = L; }
                    // ---- plan record of this death
                    { const ClSlot dr = sS[slot_del]; const int src = dr.src;
                      if (lane == 0) tag[dr.o == -1 ? slot_del : Ncap + dr.o] = -1;          // (static live count: the slot is never empty, o is -1 or a chain)
                      hd.dead_idx = ndead; hd.dead_src = (src >= 0) ? -(1 + src) : slot_del; hd.logw = logweight;
                      hd.dead_cuid = (unsigned)cl_geti<J>((const int (&)[J])uid, cd); hd.zl = Zl; }
                    ndead++; any_death = 1;
                    // ---- the order of deaths moves on
                    if (from_snap) {
                        ptr++;
                        const ClSorted e = sSort[ptr];
                        curL = cl_unid(e.L); curS = cl_uni(e.slot); curC = cl_uni((e.slot >= 0 && e.L < PC_HUGE) ? sS[e.slot >= 0 ? e.slot : 0].c : 0);
                        eCur = cl_unid(e.e);
                    } else {
                        if (lane == (nmRank >> 6)) accw &= ~(1ull << (nmRank & 63));
                        const unsigned long long nz = __ballot(lane < 16 && accw != 0ull);
                        if (nz) {
                            const int l = __ffsll((long long)nz) - 1;
                            const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(accw >> 32), l) << 32) |
                                                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned)accw, l);
                            nmRank = l * 64 + __ffsll((long long)word) - 1; { const ClCand ce = sCand[nmRank]; nmL = cl_unid(ce.L); eNm = cl_unid(ce.e); }
                        } else { nmRank = -1; nmL = PC_HUGE; eNm = 0.0; }
                    }
                    // the row of the cross-volume matrix for the cluster of the next snapshot point: requested now, turned into
                    // linear space at the end of the step, when it has arrived (nothing below waits for it)
                    double xr[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) { const int q = lane + 64 * j; xr[j] = (q < nc) ? S.XpXq[(size_t)curC * maxc + q] : NEGBIG; }
                    // ================= add_point: the newcomer takes the dead point's slot (static number of live points)
                    const int slot = slot_del;
                    const int na = cl_geti<J>(n, ca);                               // (after the death: cd may be ca)
                    const double lref_a = cl_get<J>(lref, ca), lsum_a = cl_get<J>(lsum, ca);
                    const ClCand me = sCand[ch.rank];                            // exp(Llast - Lhi), made with the ranks
                    double nref = lref_a, nsum, nEq = cl_get<J>(Eq, ca), nkd = cl_get<J>(kd, ca);
                    if (na == 0 || Llast > lref_a) {                             // the cluster's reference moves to the newcomer (utils.F90 logsumexp bookkeeping)
                        nsum = (na == 0) ? 1.0 : lsum_a * exp(lref_a - Llast) + 1.0;
                        nref = Llast;
                        if (nref - R0 > 600.0) {                                 // (cannot be represented about the launch's reference: re-base every cluster)
                            const double R1 = nref;
#pragma unroll
                            for (int j = 0; j < J; ++j) Eq[j] = (lane + 64 * j < nc && n[j] > 0) ? exp(lref[j] - R1) : 0.0;
                            E0 = exp(lprec + rZv - lxm0v - R1); kR = exp(Lhiv - R1);
                            R0 = R1;
                        }
                        nEq = exp(nref - R0); nkd = exp(Lhiv - nref);            // (a newcomer may sit far above the window: no products of precomputed factors here)
                    } else {
                        // lsum + exp(Llast - lref): the product of the two precomputed factors where both are ordinary numbers
                        const bool plain = fabs(Llast - Lhiv) < 600.0 && fabs(Lhiv - lref_a) < 600.0;
                        nsum = lsum_a + (plain ? me.e * nkd : exp(Llast - lref_a));
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) if (lane + 64 * j == ca) { n[j] = na + 1; lref[j] = nref; lsum[j] = nsum; Eq[j] = nEq; kd[j] = nkd; }
                    if (lane == 0) {
                        // list bookkeeping.  Engine rule (oracle keyed mode): a newcomer that replaces a death of its own cluster takes
                        // the dead point's position -- nothing moves.  Otherwise the dying cluster's last entry fills the hole
                        // (delete_point, array_utils.f90:433-458) and the newcomer is appended to its own cluster's list
                        ClSlot ns = sS[slot];
                        if (cd != ca) {
                            const int p = ns.p, od = lstOff[cd], oa = lstOff[ca];
                            const int last = lst[od + nd - 1];
                            if (p != nd - 1) { lst[od + p] = last; sS[last].p = p; }
                            lst[oa + na] = slot; ns.p = na;
                        }
                        ns.c = ca; ns.o = w; ns.src = w;
                        sS[slot] = ns; sL[slot] = Llast; sCS[w] = slot; tag[Ncap + w] = ca;
                    }
                    { const int r = ch.rank;
                      if (lane == (r >> 6)) accw |= 1ull << (r & 63);
                      if (Llast < nmL) { nmL = Llast; nmRank = r; eNm = me.e; } }
                    // ---- sums over the clusters: volumes (update trigger, posterior stack) and the live evidence (termination)
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int j = 0; j < J; ++j) { a += XL[j]; b += (n[j] > 0) ? (lsum[j] * srcp[n[j]]) * XL[j] * Eq[j] : 0.0; }
                    sumXL = wave_sum<4>(a); acc = wave_sum<4>(b);
                    hd.postXs = sumXL;
                    // the row of the cluster that is expected to lose the next point
                    predC = (curL <= nmL) ? curC : -1;                            // (a newcomer that dies next has its row fetched when it does)
                    if (predC >= 0) {
#pragma unroll
                        for (int j = 0; j < J; ++j) xlin[j] = exp(xr[j] - 2.0 * lxm0v);
                    }
                    replaced = true;
                    if (nd - 1 == 0 && cd != ca) need_drop = 1;                 // a cluster died (delete_cluster): the workgroup takes over
                }
            } else {
                // failed spawn: the last baby is recorded as dead with zero weight (run_time_info.f90:781-785)
                if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
                hd.dead_idx = ndead; hd.dead_src = -(1 + w); hd.logw = logzero; hd.postXs = 1.0; hd.zl = -1.0; hd.dead_cuid = 0xFFFFFFFFu;
                ndead++;
            }
            if (lane == 0) sHead[w] = hd;
            failures = replaced ? 0 : failures + 1;
            if (!replaced) nlike_failed += w_nlike;
            // ---- update trigger (nested_sampling.F90:321), delete_cluster (:339)
            const bool update = sumXL <= UTv;
            if (update) lx_last = lxm0v + log(sumXL);
            if (need_drop) { if (update) status = PC_ST_UPDATE; break; }
            if (update) { status = PC_ST_UPDATE; break; }
        }
        if (need_drop && status == PC_ST_RUNNING) {
            // A cluster died: the administrator's epoch moves on (nested_sampling.F90:339-341) and what is left of the nursery
            // fails the guard of :313 -- after the loop's termination test, which sees the same state before each of them
            bool more = true;
            if (S.max_ndead == 0) more = false;
            else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
            else if (S.use_prec) { if (!(acc > 0.0) || acc < Zl * E0) more = false; }
            if (!more || failures > S.nfail) status = PC_ST_DONE;
            else {
                double nl = 0.0;
                for (int w = lane; w < i_nursery; w += 64) {
                    nl += (double)sCh[w].nlike;
                    ClHead h{}; h.dead_idx = -1; h.ph_base = nph; h.contour = logzero; sHead[w] = h;
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
            out_i[7] = need_drop; out_i[8] = seg_hi; out_i[9] = any_death;
            out_d[0] = any_death ? cl_log(rZ, Zl, logzero) : logZ0; out_d[1] = any_death ? cl_log(rZ2, Z2l, logzero) : logZ20; out_d[2] = lx_last;
            { const double v = acc > 0.0 ? log(acc) + lxm0 + R0 : logzero; out_d[3] = (v > logzero + 800.0) ? v : pc_logaddexp(logzero, v); }
            out_d[4] = rZ;
            ctl->nlike = nlike; ctl->niter = niter; ctl->nlike_failed = nlike_failed;
            ctl->gen_cyc[0] += clock64() - cyc0; ctl->nn_walks += walks; ctl->nn_fallbacks += fallbacks;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int q = lane + 64 * j;
            if (q < nc) {
                S.logXp[q] = Xp[j]; if (any_death) S.logZXp[q] = cl_log(rZX[j], ZX[j], logzero);
                S.cl_n[q] = n[j]; S.lse_ref[q] = lref[j]; S.lse_sum[q] = lsum[j]; S.death_thr[q] = thr[j]; Fbuf[q] = Flog[j]; Gbuf[q] = Glog[j];
            }
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ write back (all waves)
    int status = out_i[0];
    const int i_nursery = out_i[2], need_drop = out_i[7], seg_hi = out_i[8];
    int epoch = out_i[3];
    const double rZ = out_d[4];
    for (int s = tid; s < Ncap; s += CL_NT) {
        const ClSlot r = sS[s];
        S.live_logL[s] = sL[s]; S.live_cluster[s] = r.c; S.live_pos[s] = r.p; S.nn_slot_owner[s] = r.o; S.slot_src[s] = r.src;
        if (r.c >= 0) S.cl_list[(size_t)r.c * Ncap + r.p] = s;
    }
    for (int c = tid; c < S.B; c += CL_NT) S.nn_chain_slot[c] = sCS[c];
    for (int c = tid; c < nc; c += CL_NT) {
        const ClOwn o = sOwn[c];
        if (o.touched) { S.logZp[c] = cl_log(o.rzp, o.zp, S.logzero); S.logZp2[c] = cl_log(o.rzp2, o.zp2, S.logzero); S.logZpXp[c] = cl_log(o.rzpx, o.zpx, S.logzero); }
    }
    // the cross-volume matrix picks up the factors of the launch's deaths: X_p X_q *= f_p f_q, X_p^2 *= g_p
    for (int e0 = tid; e0 < nc * nc; e0 += 4 * CL_NT) {               // (four entries in flight per thread: the loads before the stores)
        double v[4], add[4]; size_t at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * CL_NT; const bool in = e < nc * nc;
            const int p = in ? e / nc : 0, q = in ? e % nc : 0;
            at[u] = (size_t)p * maxc + q; add[u] = !in ? 0.0 : ((p == q) ? Gbuf[p] : Fbuf[p] + Fbuf[q]);
            v[u] = S.XpXq[at[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (add[u] != 0.0) S.XpXq[at[u]] = v[u] + add[u];
    }
    // plan records of the chains this launch consumed
    for (int w = i_nursery + tid; w <= seg_hi; w += CL_NT) {
        const ClHead r = sHead[w];
        PcPlanHead h;
        h.dead_idx = r.dead_idx; h.dead_src = r.dead_src; h.ph_base = r.ph_base; h.dead_cuid = r.dead_cuid; h.ph_cuid = r.ph_cuid; h.ph_count = 0;
        h.logw = r.logw; h.contour = r.contour;
        const bool spawn_failed = r.zl < 0.0;
        h.postX = spawn_failed ? 0.0 : ref_d[1]; h.postXs = r.postXs;
        h.postZ = spawn_failed ? 0.0 : cl_log(rZ, r.zl, S.logzero);
        if (h.dead_idx < 0) { h.dead_src = 0; h.dead_cuid = 0u; h.logw = 0.0; h.postX = 0.0; h.postXs = 1.0; h.postZ = 0.0; }
        *(PcPlanHead *)&S.plan[w] = h;
        for (int m = 0; m < nw; ++m) S.plan[w].ph_mask[m] = masks[(size_t)w * nw + m];
    }
    // find_min_loglikelihoods (run_time_info.f90:883-909), once: lowest (logL, list position) of every cluster
    for (int c = tid; c < CL_MAXC; c += CL_NT) { kmin[c] = KEY_HUGE; lstOff[c] = 0x7fffffff; }
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sS[s].c >= 0) atomicMin(&kmin[sS[s].c], d2key(sL[s]));
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) if (sS[s].c >= 0 && d2key(sL[s]) == kmin[sS[s].c]) atomicMin(&lstOff[sS[s].c], sS[s].p);
    __syncthreads();
    for (int s = tid; s < Ncap; s += CL_NT) { const ClSlot r = sS[s]; if (r.c >= 0 && d2key(sL[s]) == kmin[r.c] && r.p == lstOff[r.c]) { S.imin_slot[r.c] = s; S.logLp[r.c] = sL[s]; } }
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
            // Everything moves up by one cluster, in place: every thread takes its share of the sources into registers, the
            // workgroup meets, then the stores (a thread that walks the arrays alone pays a memory round trip per element: 2 ms for
            // the cross-volume matrix of 70 clusters, a hundred times per run)
            {
                const int m1 = nc - 1;
                double v[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    const int e = tid + u * CL_NT;
                    if (e < m1 * m1) { const int na = e / m1, nb = e % m1; v[u] = S.XpXq[(size_t)(na + (na >= p)) * maxc + nb + (nb >= p)]; }
                }
                double t[9]; int ti[2]; unsigned tu = 0u;
                const int c = p + tid;
                if (c < m1) {
                    t[0] = S.logLp[c + 1]; t[1] = S.logXp[c + 1]; t[2] = S.logZp[c + 1]; t[3] = S.logZXp[c + 1]; t[4] = S.logZp2[c + 1]; t[5] = S.logZpXp[c + 1];
                    t[6] = S.lse_ref[c + 1]; t[7] = S.lse_sum[c + 1]; t[8] = S.death_thr[c + 1]; ti[0] = S.cl_n[c + 1]; ti[1] = S.imin_slot[c + 1]; tu = S.cl_uid[c + 1];
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    const int e = tid + u * CL_NT;
                    if (e < m1 * m1) S.XpXq[(size_t)(e / m1) * maxc + e % m1] = v[u];
                }
                if (c < m1) {
                    S.logLp[c] = t[0]; S.logXp[c] = t[1]; S.logZp[c] = t[2]; S.logZXp[c] = t[3]; S.logZp2[c] = t[4]; S.logZpXp[c] = t[5];
                    S.lse_ref[c] = t[6]; S.lse_sum[c] = t[7]; S.death_thr[c] = t[8]; S.cl_n[c] = ti[0]; S.imin_slot[c] = ti[1]; S.cl_uid[c] = tu;
                }
            }
            {   // Cholesky factors and covariance matrices: groups of clusters whose elements fit sixteen to a thread
                const int DD = D * D;
                int G = (16 * CL_NT) / DD; if (G < 1) G = 1;
                for (int c0 = p; c0 < nc - 1; c0 += G) {
                    const int g = (nc - 1 - c0 < G) ? nc - 1 - c0 : G, tot = g * DD;
                    for (int e0 = 0; e0 < tot; e0 += 16 * CL_NT) {               // (one pass unless a single matrix exceeds the share)
                        double a[16], b[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { const int e = e0 + tid + u * CL_NT; if (e < tot) { a[u] = S.chol[(size_t)(c0 + 1) * DD + e]; b[u] = S.cov[(size_t)(c0 + 1) * DD + e]; } }
                        __syncthreads();
#pragma unroll
                        for (int u = 0; u < 16; ++u) { const int e = e0 + tid + u * CL_NT; if (e < tot) { S.chol[(size_t)c0 * DD + e] = a[u]; S.cov[(size_t)c0 * DD + e] = b[u]; } }
                        if (tot > 16 * CL_NT) __syncthreads();                   // (the next pass reads what this one's neighbours write)
                    }
                }
            }
            for (int s = tid; s < Ncap; s += CL_NT) if (sS[s].c > p) { sS[s].c -= 1; S.live_cluster[s] = sS[s].c; }
            __syncthreads();
            // (the lists of the clusters behind the deleted one move up a row)
            for (int s = tid; s < Ncap; s += CL_NT) if (sS[s].c >= p) S.cl_list[(size_t)sS[s].c * Ncap + sS[s].p] = s;
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
    if (tid == 0) ctl->gen_cyc[3] += clock64() - t_start;            // (developer counters: staging, loop, whole kernel)
    pc_publish_ctl(S);
}

extern "C" int pc_consume_cl_fits(const PcState *S, int nc)
{
    if (nc < 2 || nc > CL_MAXC || S->B > 1024 || S->nr > 64 * PC_MASK_WORDS) return 0;
    return cl_layout(S->Ncap, S->B, S->nr).total + 1024 <= (size_t)160 * 1024;
}

extern "C" int pc_launch_consume_cl(const PcState *S, int nc, hipStream_t st)
{
    const size_t sh = cl_layout(S->Ncap, S->B, S->nr).total;
    static size_t done1 = 0, done2 = 0;
    if (nc <= 64) {
        if (sh > done1) { (void)hipFuncSetAttribute((const void *)k_consume_cl<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); done1 = sh; }
        hipLaunchKernelGGL(k_consume_cl<1>, dim3(1), dim3(CL_NT), sh, st, *S);
    } else {
        if (sh > done2) { (void)hipFuncSetAttribute((const void *)k_consume_cl<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); done2 = sh; }
        hipLaunchKernelGGL(k_consume_cl<2>, dim3(1), dim3(CL_NT), sh, st, *S);
    }
    return 0;
}
