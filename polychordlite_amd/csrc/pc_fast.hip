// pc_fast.hip -- the contraction for the common case: ONE cluster, static number of live points.
//
// Same decisions as k_consume (replace_point / delete_outermost_point / update_evidence /
// more_samples_needed, src/polychord/run_time_info.f90:716-817,211-296, nested_sampling.F90:514-543),
// organised around what the hardware can do (single-wave latencies measured with tools/ubench.hip:
// an instruction every 5-6 cycles on a wave alone on its SIMD, exp ~120, log ~460, LDS round trip 70 -- round 6's numbers;
// "dependent fp64 op 32 cycles" in earlier rounds included 24 cycles of the measuring loop):
//
//   k_sort_live     bitonic sort of the live slots by (logL, list position) in LDS: deaths happen in
//                   ascending logL, so the serial pass below only walks a pointer.
//   k_consume_fast  ONE wavefront.
//       pass A (serial, compares only): contour = min(next sorted snapshot point, min of the points
//              inserted during this launch); accept / reject; list bookkeeping with the reference's
//              append / swap-with-last semantics (array_utils.f90:396-458); the inserted points keep a
//              per-lane cached minimum so that the common step is O(1).  No transcendental, no global
//              memory read: the chains' inputs are staged in LDS once.
//       pass B (every 64 deaths): the evidence recursion of update_evidence is affine in exp-space,
//              so the 64 deaths of a chunk are evaluated by lane-parallel log-space prefix scans
//              (6 levels of logaddexp instead of 64 serial ones).
//       The termination test cannot fire inside a chunk as long as the live evidence is more than
//       e^(1+8G/n) times the threshold (bound in DESIGN.md); inside that margin the kernel falls
//       back to one exact evaluation per death.
//       Kill-off (nested_sampling.F90:381-384) is the same machinery with lane = sorted rank.
//   k_ph_prepare    phantom masks / counts / base offsets from the recorded contour of every chain
//                   (run_time_info.f90:747-757), in consumption order: deterministic layout.
//
// Exact ties in logL are broken by (sorted snapshot first, then lane order) instead of list position.
#include "pc_state.h"

#include "pc_keys.h"

__device__ __forceinline__ double wave_min_f64(double v)
{
    v = fmin(v, dpp_f64<PC_DPP_XOR1>(v));
    v = fmin(v, dpp_f64<PC_DPP_XOR2>(v));
    v = fmin(v, dpp_f64<PC_DPP_HALF_MIRROR>(v));
    v = fmin(v, dpp_f64<PC_DPP_MIRROR>(v));
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmin(fmin(r0, r1), fmin(r2, r3));
}

// inclusive prefix of logaddexp over the wave (Hillis-Steele, 6 levels)
// two independent scans interleaved (their exp/log chains overlap)
__device__ __forceinline__ void scan_lae2(double &v, double &u, int lane, int m)
{
    for (int k = 1; k < m; k <<= 1) {      // m = number of live records: ceil(log2 m) levels
        const double ov = __shfl_up(v, k), ou = __shfl_up(u, k);
        const double nv = lae2(v, ov), nu = lae2(u, ou);
        if (lane >= k) { v = nv; u = nu; }
    }
}
__device__ __forceinline__ double scan_add(double v, int lane, int m)
{
    for (int k = 1; k < m; k <<= 1) {
        const double o = __shfl_up(v, k);
        if (lane >= k) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// sort of the live slots by (logL, list position); free slots last
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sort_live_body(const PcState &S, int npow2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *kv = (double *)smem;            // [npow2]
    int *kp = (int *)(kv + npow2);          // [npow2] list position
    int *ks = kp + npow2;                   // [npow2] slot
    const int tid = threadIdx.x;
    for (int i = tid; i < npow2; i += 1024) {
        const bool used = i < S.Ncap && S.live_cluster[i] >= 0;
        kv[i] = used ? S.live_logL[i] : PC_HUGE; kp[i] = used ? S.live_pos[i] : 0x7fffffff; ks[i] = i < S.Ncap ? i : -1;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 1024) {
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
    for (int i = tid; i < NS; i += 1024) { S.sort_slot[i] = (i < npow2) ? ks[i] : -1; S.sort_key[i] = (i < npow2) ? d2key(kv[i]) : KEY_HUGE; }
    // the rank of every live point in this order = how many deaths of the snapshot come before its own: k_nn_lists_d tells by it which
    // candidates are certainly alive when a chain is looked at (behind the candidates' codes: [Ncap + B] codes, [Ncap] ranks)
    if (S.nn_code) {
        int *rank = S.nn_code + (size_t)S.Ncap + S.B;
        for (int i = tid; i < npow2; i += 1024) {
            const int sl = ks[i]; if (sl >= 0 && sl < S.Ncap) rank[sl] = (kv[i] < PC_HUGE) ? i : 0x7fffffff;
            // the number of live points, behind the ranks (free slots sort last)
            if (kv[i] < PC_HUGE && (i + 1 == npow2 || !(kv[i + 1] < PC_HUGE))) rank[S.Ncap] = i + 1;
            if (i == 0 && !(kv[0] < PC_HUGE)) rank[S.Ncap] = 0;
        }
    }
}
__global__ __launch_bounds__(1024) void k_sort_live(PcState S, int npow2) { sort_live_body(S, npow2); }
__global__ __launch_bounds__(1024) void k_sort_live_many(const PcManyRec *R, int npow2) { sort_live_body(pc_many_state(R, blockIdx.y), npow2); }


// ------------------------------------------------------------------------------------------
// the contraction
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_consume_fast(PcState S, int final_mode)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int Ncap = S.Ncap, nr = S.nr, nT = S.nT;
    const int NS = (Ncap + 63) & ~63;
    PcCtl *ctl = S.ctl;
    const int nchain = ctl->i_nursery;
    // LDS carve: 8-byte arrays first, 4-byte arrays after (typed pointer arithmetic only, so that the
    // compiler keeps the LDS address space)
    double *sL = (double *)smem;               // [NS] logL by slot (+HUGE free)
    double *cLast = sL + NS;                   // [B] logL of the last baby of chain w
    unsigned long long *sK = (unsigned long long *)(cLast + S.B);   // [NS] sortable keys by slot
    unsigned long long *sSortK = sK + NS;      // [NS] keys in sorted order
    unsigned long long *cLastK = sSortK + NS;  // [B]
    unsigned long long *rLgK = cLastK + S.B, *rAddK = rLgK + 64;   // step records of the chunk, [64] each
    int *sSrc = (int *)(rAddK + 64);           // [NS] -1 or chain that inserted the point in this launch
    int *sSort = sSrc + NS;                    // [NS] slots in ascending (logL, pos)
    int *cNlike = sSort + NS;                  // [B]
    int *cEpoch = cNlike + S.B;                // [B]
    int *rW = cEpoch + S.B, *rKind = rW + 64, *rSrc = rKind + 64;   // [64] each

    for (int s = lane; s < NS; s += 64) {
        const bool used = s < Ncap && S.live_cluster[s] >= 0;
        sL[s] = used ? S.live_logL[s] : PC_HUGE;
        sSrc[s] = -1; sSort[s] = S.sort_slot[s];
    }
    int n = S.cl_n[0];
    for (int w = lane; w < nchain; w += 64) {
        cLast[w] = S.baby_logL[(size_t)w * nr + nr - 1]; cNlike[w] = S.ch_nlike[w]; cEpoch[w] = S.ch_epoch[w];
    }
    int i_nursery = nchain, failures = ctl->failures, ndead = ctl->ndead;
    const int epoch = ctl->admin_epoch;
    int nc_dead = ctl->ncluster_dead, nc = ctl->ncluster;
    long long nlike = ctl->nlike, niter = ctl->niter, nlike_failed = ctl->nlike_failed;
    double logZ = ctl->logZ, logZ2 = ctl->logZ2, lx_last = ctl->logX_last_update;
    double Xp = S.logXp[0], Zp = S.logZp[0], ZXp = S.logZXp[0], Zp2 = S.logZp2[0], ZpXp = S.logZpXp[0], XX = S.XpXq[0];
    double lseRef = S.lse_ref[0], lseSum = S.lse_sum[0], thr = S.death_thr[0];
    const unsigned cuid = S.cl_uid[0];
    int status = PC_ST_RUNNING, error = PC_ERR_NONE;
    const int seg_hi = i_nursery - 1;
    double live_logZ_val = S.logzero;
    const double log2v = 0.6931471805599453;
    const double l0 = log((double)n + 0.0), l1 = log((double)n + 1.0), l2 = log((double)n + 2.0), d01 = l0 - l1;
    __syncthreads();

    // ================= pass B: evidence of the deaths among the m recorded steps =================
    // lane i < m owns step i.  isdeath lanes carry (L, Ladd, Xb, XXb, a0..a2 = log n, log(n+1), log(n+2));
    // the other lanes are neutral elements of the scans.  Returns logZ after the lane's step.
    auto evidence_chunk = [&](int m, bool isdeath, double L, double Ladd, double Xb, double XXb,
                              double a0, double a1, double a2) __attribute__((always_inline)) -> double {
        const double e01 = isdeath ? a0 - a1 : 0.0;
        const double T = isdeath ? Xb + L - a1 : NEGBIG;                  // log of the evidence increment
        const double U = isdeath ? XXb + L + a0 - a1 - a2 : NEGBIG;       // increment of <Z X>
        // <ZX>_i = e01_i <ZX>_{i-1} + U_i is affine: with S_i = sum_{j<=i} e01_j,
        // ZX_i = S_i + lae(ZX_0, prefix_lae(U_j - S_j))
        const double Sd = scan_add(e01, lane, m);
        double PT = T, PV = isdeath ? U - Sd : NEGBIG;
        scan_lae2(PT, PV, lane, m);
        const double Zi = lae2(logZ, PT), Zpi = lae2(Zp, PT);
        const double ZXi = Sd + lae2(ZXp, PV), ZpXpi = Sd + lae2(ZpXp, PV);
        double ZXprev = __shfl_up(ZXi, 1), ZpXpprev = __shfl_up(ZpXpi, 1);
        if (lane == 0) { ZXprev = ZXp; ZpXpprev = ZpXp; }
        const double cz = log2v + XXb + 2 * L - a1 - a2;
        double W = isdeath ? lae2(log2v + ZXprev + L - a1, cz) : NEGBIG;
        double Wp = isdeath ? lae2(log2v + ZpXpprev + L - a1, cz) : NEGBIG;
        scan_lae2(W, Wp, lane, m);
        const double Z2i = lae2(logZ2, W), Zp2i = lae2(Zp2, Wp);
        const int last = m - 1;
        logZ = readlane_f64(Zi, last); Zp = readlane_f64(Zpi, last); ZXp = readlane_f64(ZXi, last);
        ZpXp = readlane_f64(ZpXpi, last); logZ2 = readlane_f64(Z2i, last); Zp2 = readlane_f64(Zp2i, last);
        // live log-sum-exp bookkeeping (run_time_info.f90:683-709), rebased on the new maximum
        const double mx = wave_max(isdeath ? Ladd : NEGBIG);
        if (mx > lseRef) { lseSum *= exp(lseRef - mx); lseRef = mx; }
        const double de = isdeath ? ((Ladd > NEGBIG ? exp(Ladd - lseRef) : 0.0) - exp(L - lseRef)) : 0.0;
        lseSum += wave_sum<4>(de);
        return Zi;
    };

    if (final_mode) {
        // nested_sampling.F90:381-384: every remaining live point dies, lowest first: death i of a
        // chunk is the i-th sorted slot, n shrinks by one per death.
        const int n0 = n;
        for (int base = 0; base < n0 && status == PC_ST_RUNNING; base += 64) {
            const int m = min(64, n0 - base);
            if (ndead + m > S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
            const bool on = lane < m;
            const int slot = on ? sSort[base + lane] : 0;
            const int myn = n0 - base - lane;                               // live points before my death
            const double L = on ? sL[slot] : NEGBIG;
            const double a0 = on ? log((double)myn + 0.0) : 0.0, a1 = on ? log((double)myn + 1.0) : 0.0,
                         a2 = on ? log((double)myn + 2.0) : 0.0;
            const double e01 = a0 - a1, e02 = a0 - a2;
            const double sx = scan_add(e01, lane, m), sxx = scan_add(e02, lane, m);
            const double Xb = Xp + (sx - e01), XXb = XX + (sxx - e02);      // volumes before my death
            const double Zi = evidence_chunk(m, on, L, NEGBIG, Xb, XXb, a0, a1, a2);
            if (on) {
                const double *row = S.live + (size_t)slot * nT;
                double *dst = S.dead + (size_t)(ndead + lane) * nT;
                for (int e = 0; e < nT; ++e) dst[e] = row[e];
                S.dead_logw[ndead + lane] = Xb - a1; S.dead_postX[ndead + lane] = Xb + e01; S.dead_postZ[ndead + lane] = Zi;
                S.dead_cuid[ndead + lane] = cuid; S.dead_entry[ndead + lane] = S.live_entry[slot];
            }
            Xp = Xp + readlane_f64(sx, m - 1); XX = XX + readlane_f64(sxx, m - 1);
            thr = readlane_f64(L, m - 1);
            ndead += m;
        }
        if (status == PC_ST_RUNNING) {
            for (int s = lane; s < NS; s += 64) sL[s] = PC_HUGE;
            n = 0;
            if (lane == 0 && nc_dead < S.maxc_dead) { S.logZp_dead[nc_dead] = Zp; S.logZp2_dead[nc_dead] = Zp2; S.cl_uid_dead[nc_dead] = cuid; }
            nc_dead++; nc = 0;
            status = PC_ST_DONE;
        }
    }

    // ================= pass A: the serial walk (integer only) =================
    // keys of the live points replace their logL from here on
    for (int i = lane; i < NS; i += 64) { sK[i] = d2key(sL[i]); const int ss = sSort[i]; sSortK[i] = (i < n && ss >= 0) ? d2key(sL[ss]) : KEY_HUGE; }
    for (int w = lane; w < nchain; w += 64) cLastK[w] = d2key(cLast[w]);
    __syncthreads();
    int ptr = 0;
    int snapSlot = __builtin_amdgcn_readfirstlane(sSort[0]);
    unsigned long long snapK = (n > 0) ? uni64(sSortK[0]) : KEY_HUGE;
    int nxtSlot = __builtin_amdgcn_readfirstlane(sSort[1 < NS ? 1 : 0]);
    unsigned long long nxtK = (1 < n) ? uni64(sSortK[1]) : KEY_HUGE;
    unsigned long long im_k = KEY_HUGE; int im_s = -1;                      // per-lane min of inserted points
    unsigned long long insK = KEY_HUGE; int ins_lane = 0;
    unsigned long long lastDeathK = d2key(thr);
    int m = 0, mdeaths = 0, ndead0 = ndead, deaths_total = 0;
    const double XpL0 = Xp, XXL0 = XX, d02 = l0 - l2;
    long long cyB = 0, nFlush = 0, nSlow = 0, cyIns = 0, nIns = 0, cyCommon = 0, nCommon = 0, cyRej = 0, nRej = 0;
    const long long cy0 = clock64();
    const int G = (n >= 1024) ? 64 : max(1, n / 16);
    // the precision criterion cannot fire within G deaths while
    //   live_logZ - log(prec) - logZ >= -log( exp(-2G/n) - prec G/(n+1) )      
    double gthr = PC_HUGE;
    if (S.use_prec && n > 0) {
        const double arg = exp(-2.0 * G / (double)n) - exp(S.log_prec) * (double)G / ((double)n + 1.0);
        if (arg > 0.0) gthr = -log(arg) + 0.05;
    }
    // deaths until the update trigger  logXp <= logX_last_update + log(compression)  (nested_sampling.F90:321)
    int kupd = 0x7fffffff;
    if (n > 0) {
        const double tx = lx_last + S.log_cf;
        double kf = ceil((XpL0 - tx) / (-d01));
        if (kf < 1.0) kf = 1.0;
        if (kf < 2.0e9) {
            kupd = (int)kf;
            while (kupd > 1 && XpL0 + (double)(kupd - 1) * d01 <= tx) kupd--;
            while (XpL0 + (double)kupd * d01 > tx) kupd++;
        }
    }
    bool fastphase = false;
    auto eval_guard = [&]() __attribute__((always_inline)) {
        if (!S.use_prec) { fastphase = true; return; }
        live_logZ_val = lseRef + log(lseSum) - l0 + Xp;
        fastphase = (live_logZ_val - (S.log_prec + logZ)) > gthr;
    };
    if (!final_mode && n > 0) eval_guard();

    auto flush = [&]() __attribute__((always_inline)) {
        if (m == 0) return;
        const bool on = lane < m;
        const int kind = on ? rKind[lane] : 0, w = on ? rW[lane] : 0;
        const bool isdeath = kind == 2;
        const double Lg = on ? key2d(rLgK[lane]) : NEGBIG, Ladd = isdeath ? key2d(rAddK[lane]) : NEGBIG;
        // volumes before my death from the number of deaths before it (Xp only moves by log(n/(n+1)))
        const unsigned long long km = __ballot(isdeath);
        const int nb = (deaths_total - mdeaths) + __popcll(km & ((1ull << lane) - 1ull));
        const double Xb = XpL0 + (double)nb * d01, XXb = XXL0 + (double)nb * d02;
        const double Zi = evidence_chunk(m, isdeath, isdeath ? Lg : NEGBIG, Ladd, Xb, XXb, l0, l1, l2);
        const unsigned long long dm = __ballot(on && kind >= 1);
        const int didx = ndead0 + __popcll(dm & ((1ull << lane) - 1ull));
        if (on) {
            PcPlan *pw = S.plan + w;
            pw->ph_cuid = cuid; pw->ph_count = -1; pw->ph_base = 0;
            pw->contour = (kind == 0) ? PC_HUGE : Lg;                       // dropped chains get no phantoms
            pw->dead_idx = (kind >= 1) ? didx : -1;
            if (kind == 2) {
                pw->dead_src = rSrc[lane]; pw->logw = Xb - l1; pw->postX = Xb + d01; pw->postXs = 1.0; pw->postZ = Zi; pw->dead_cuid = cuid;
            } else if (kind == 1) {
                pw->dead_src = -(1 + w); pw->logw = S.logzero; pw->postX = 0.0; pw->postXs = 1.0; pw->postZ = 0.0; pw->dead_cuid = 0xFFFFFFFFu;
            }
        }
        ndead0 += __popcll(dm);
        Xp = XpL0 + (double)deaths_total * d01; XX = XXL0 + (double)deaths_total * d02;
        m = 0; mdeaths = 0;
    };

    while (status == PC_ST_RUNNING) {
        // ---- more_samples_needed (nested_sampling.F90:514-543); inside a chunk of the fast phase the
        //      precision criterion cannot fire, the integer criteria are always exact
        bool more = true;
        if (S.max_ndead == 0) more = false;
        else if (S.max_ndead > 0 && ndead >= S.max_ndead) more = false;
        else if (S.use_prec && !fastphase) {
            live_logZ_val = lseRef + log(lseSum) - l0 + Xp;
            more = !(live_logZ_val < S.log_prec + logZ);
        }
        if (!more || failures > S.nfail) { status = PC_ST_DONE; break; }
        if (i_nursery == 0) break;

        const int w = i_nursery - 1;
        i_nursery--;
        const unsigned long long LlastK = uni64(cLastK[w]);
        nlike += __builtin_amdgcn_readfirstlane(cNlike[w]); niter++;
        const int ep = __builtin_amdgcn_readfirstlane(cEpoch[w]);
        const unsigned long long LgK = snapK < insK ? snapK : insK;
        if (ep != epoch) {                             // nested_sampling.F90:313: only nlike is counted
            nlike_failed += __builtin_amdgcn_readfirstlane(cNlike[w]);
            if (lane == 0) { rW[m] = w; rKind[m] = 0; rLgK[m] = LgK; }
            m++;
            if (m == 64) { __builtin_amdgcn_wave_barrier(); flush(); if (fastphase) eval_guard(); }
            continue;
        }
        if (ndead >= S.Dcap) { status = PC_ST_ERROR; error = PC_ERR_DEAD_CAP; break; }
        bool replaced = false;
        if (LlastK > LgK) {
            const bool from_snap = snapK <= insK;
            const int slot = from_snap ? snapSlot : __builtin_amdgcn_readlane(im_s, ins_lane);
            const int src = from_snap ? -1 : __builtin_amdgcn_readfirstlane(sSrc[slot]);
            if (lane == 0) {
                rW[m] = w; rKind[m] = 2; rSrc[m] = (src >= 0) ? -(1 + src) : slot;
                rLgK[m] = LgK; rAddK[m] = LlastK;
                sK[slot] = LlastK; sSrc[slot] = w;    // the baby takes the dead point's slot and list position
            }
            lastDeathK = LgK;
            ndead++; m++; mdeaths++; deaths_total++;
            const int own = slot & 63;
            if (from_snap) {
                ptr++;
                snapSlot = nxtSlot; snapK = (ptr < n) ? nxtK : KEY_HUGE;
                const int pn = (ptr + 1 < NS) ? ptr + 1 : ptr;
                nxtSlot = __builtin_amdgcn_readfirstlane(sSort[pn]);
                nxtK = (ptr + 1 < n) ? uni64(sSortK[pn]) : KEY_HUGE;
                if (lane == own && LlastK < im_k) { im_k = LlastK; im_s = slot; }
                if (LlastK < insK) { insK = LlastK; ins_lane = own; }
            } else {
                // the minimum of the inserted points died and its slot holds the new baby: recompute
                // that stride's inserted minimum cooperatively, then the wave minimum
                __builtin_amdgcn_wave_barrier();
                const long long ti = clock64(); nIns++;
                double bv = PC_HUGE; int bs = -1;
                for (int j0 = 0; j0 < NS / 64; j0 += 64) {
                    const int j = j0 + lane, sidx = own + 64 * j;
                    if (j < NS / 64 && sSrc[sidx] >= 0) { const double v = key2d(sK[sidx]); if (v < bv) { bv = v; bs = sidx; } }
                }
                const double mv = wave_min_f64(bv);
                const unsigned long long mm = __ballot(bv == mv && bs >= 0);
                const int wl = mm ? __ffsll((long long)mm) - 1 : 0;
                const int ws = __builtin_amdgcn_readlane(bs, wl);
                if (lane == own) { im_k = d2key(mv); im_s = ws; }
                const double gm = wave_min_f64(key2d(im_k));
                insK = d2key(gm);
                const unsigned long long ml = __ballot(im_k == insK);
                ins_lane = ml ? __ffsll((long long)ml) - 1 : 0;
                cyIns += clock64() - ti;
            }
            replaced = true;
        } else {
            // failed spawn (run_time_info.f90:781-785): the baby is recorded dead with zero weight
            if (lane == 0) { rW[m] = w; rKind[m] = 1; rLgK[m] = LgK; }
            ndead++; m++;
            nlike_failed += __builtin_amdgcn_readfirstlane(cNlike[w]);
        }
        const bool upd = replaced && deaths_total == kupd;
        if (m == 64 || mdeaths == G || upd || (!fastphase && replaced)) {
            __builtin_amdgcn_wave_barrier();
            const long long tb = clock64();
            if (!fastphase) nSlow++;
            flush();
            if (fastphase) eval_guard();
            cyB += clock64() - tb; nFlush++;
        }
        failures = replaced ? 0 : failures + 1;
        // ---- update trigger (nested_sampling.F90:321); one cluster: logsumexp(logXp) = logXp
        if (upd) { lx_last = Xp; status = PC_ST_UPDATE; }
    }
    if (m > 0) { __builtin_amdgcn_wave_barrier(); flush(); }
    if (!final_mode) { thr = key2d(lastDeathK); for (int i = lane; i < NS; i += 64) sL[i] = key2d(sK[i]); }
    if (S.use_prec && n > 0) live_logZ_val = lseRef + log(lseSum) - l0 + Xp;

    // ---- write back
    const long long cy1 = clock64();
    __syncthreads();
    double Lmin = PC_HUGE; int minSlot = -1;
    if (n > 0) {   // contour and its slot for the seed kernels / the next launch
        const bool from_snap = snapK <= insK;
        Lmin = key2d(from_snap ? snapK : insK);
        minSlot = from_snap ? snapSlot : __builtin_amdgcn_readlane(im_s, ins_lane);
    }
    for (int s = lane; s < Ncap; s += 64) {
        S.live_logL[s] = sL[s]; S.slot_src[s] = sSrc[s];
        if (final_mode && status == PC_ST_DONE) S.live_cluster[s] = -1;
    }
    if (lane == 0) {
        S.logLp[0] = Lmin; S.imin_slot[0] = minSlot; S.logXp[0] = Xp; S.logZp[0] = Zp; S.logZXp[0] = ZXp;
        S.logZp2[0] = Zp2; S.logZpXp[0] = ZpXp; S.XpXq[0] = XX; S.lse_ref[0] = lseRef; S.lse_sum[0] = lseSum;
        S.death_thr[0] = thr; S.cl_n[0] = n;
        ctl->status = status; ctl->error = error; ctl->i_nursery = i_nursery; ctl->failures = failures;
        ctl->ndead = ndead; ctl->seg_hi = seg_hi; ctl->seg_lo = i_nursery; ctl->cluster_deleted = 0;
        ctl->ncluster = nc; ctl->ncluster_dead = nc_dead;
        ctl->nlike = nlike; ctl->niter = niter; ctl->nlike_failed = nlike_failed; ctl->logZ = logZ; ctl->logZ2 = logZ2; ctl->logX_last_update = lx_last;
        ctl->live_logZ = live_logZ_val;
        ctl->dbg[0] += cy1 - cy0; ctl->dbg[1] += cyB; ctl->dbg[2] += cyCommon; ctl->dbg[3] += nCommon; ctl->dbg[4] += nFlush;
        ctl->dbg[5] += cyIns; ctl->dbg[6] += nIns; ctl->dbg[7] += cyRej;
    }
}

// ------------------------------------------------------------------------------------------
// phantoms of the consumed chains from the recorded contour: masks, counts, bases (one workgroup,
// chains in consumption order = descending chain index)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ph_prepare(PcState S)
{
    __shared__ int tmp[1024];
    __shared__ int carry;
    PcCtl *ctl = S.ctl;
    const int hi = ctl->seg_hi, lo = ctl->seg_lo, nr = S.nr, tid = threadIdx.x;
    if (tid == 0) carry = ctl->nphantom;
    __syncthreads();
    for (int base = 0; base <= hi - lo; base += 1024) {
        const int t = base + tid, w = hi - t;
        int cnt = 0;
        const bool mine = (w >= lo) && S.plan[w].ph_count < 0;
        if (mine) {
            const double Lg = S.plan[w].contour;
            const double *b = S.baby_logL + (size_t)w * nr;
            for (int mw = 0; mw < (nr + 62) / 64; ++mw) {
                unsigned long long mask = 0ull;
                for (int i = mw * 64; i < min(nr - 1, mw * 64 + 64); ++i) if (b[i] > Lg) mask |= 1ull << (i & 63);
                S.plan[w].ph_mask[mw] = mask;
                cnt += __popcll(mask);
            }
        } else if (w >= lo) cnt = S.plan[w].ph_count;
        tmp[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = tid >= off ? tmp[tid - off] : 0;
            __syncthreads();
            tmp[tid] += v;
            __syncthreads();
        }
        if (w >= lo) { S.plan[w].ph_base = carry + tmp[tid] - cnt; S.plan[w].ph_count = cnt; }
        __syncthreads();
        if (tid == 1023) carry += tmp[1023];
        __syncthreads();
    }
    if (tid == 0) ctl->nphantom = carry;
    pc_publish_ctl(S);                                              // the serial kernel's round ends here
}

// ------------------------------------------------------------------------------------------
static size_t fast_lds(const PcState *S)
{
    const size_t NS = ((size_t)S->Ncap + 63) & ~(size_t)63;
    return 8 * (3 * NS + 2 * (size_t)S->B + 128) + sizeof(int) * (2 * NS + 2 * (size_t)S->B + 192) + 64;
}

extern "C" int pc_fast_fits(const PcState *S) { return fast_lds(S) <= 160 * 1024 && S->Ncap <= 32768; }

extern "C" int pc_launch_sort_live(const PcState *S, hipStream_t st)
{
    int npow2 = 64;
    while (npow2 < S->Ncap) npow2 <<= 1;
    const size_t shs = (size_t)npow2 * 16;
    if (shs > 160 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_sort_live, shs);
    hipLaunchKernelGGL(k_sort_live, dim3(1), dim3(1024), shs, st, *S, npow2);
    return 0;
}

extern "C" int pc_launch_sort_live_many(const PcState *S, const PcManyRec *dR, int R, hipStream_t st)
{
    int npow2 = 64;
    while (npow2 < S->Ncap) npow2 <<= 1;
    const size_t shs = (size_t)npow2 * 16;
    if (shs > 160 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_sort_live_many, shs);
    hipLaunchKernelGGL(k_sort_live_many, dim3(1, R), dim3(1024), shs, st, dR, npow2);
    return 0;
}

extern "C" int pc_launch_consume_fast(const PcState *S, int final_mode, hipStream_t st)
{
    const size_t sh = fast_lds(S);
    if (sh > 160 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_consume_fast, sh);
    if (pc_launch_sort_live(S, st)) return 1;
    hipLaunchKernelGGL(k_consume_fast, dim3(1), dim3(64), sh, st, *S, final_mode);
    return 0;
}

extern "C" void pc_launch_ph_prepare(const PcState *S, hipStream_t st)
{
    hipLaunchKernelGGL(k_ph_prepare, dim3(1), dim3(1024), 0, st, *S);
}
