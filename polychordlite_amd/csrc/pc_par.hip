// pc_par.hip -- the single-cluster contraction as a PARALLEL computation over a whole nursery.
//
// Same decisions as the serial walk of k_consume_fast (replace_point / delete_outermost_point /
// update_evidence / more_samples_needed: src/polychord/run_time_info.f90:716-817,211-296,
// nested_sampling.F90:262-321,514-543), but nothing in it is a loop over deaths.  With one cluster
// and a static number of live points the sequential process
//      step t:  g_t = min(live);  if c_t > g_t: the minimum dies, the candidate c_t takes its slot
// has a closed form in terms of order statistics (S = sorted snapshot of the live set, A_t = the
// candidates accepted before step t, k_t = |A_t|):
//   * every inserted point is larger than everything that died before it, so the points that died
//     before step t are the k_t smallest of S u A_t, and g_t is the (k_t+1)-th smallest;
//   * hence  accept_t  <=>  #{e in S: e < c_t} > #{s < t accepted: c_s >= c_t}: acceptance depends on
//     RANKS only.  The left side is a binary search, the right side a popcount over a bitmap of
//     accepted ranks.  Steps are resolved 64 at a time (one wavefront, a fixpoint iteration over the
//     in-chunk dependencies that settles in 2-3 rounds);
//   * points accepted later are larger than g_t, so g_t = u[k_t] with u = sorted(S u A_final): the
//     j-th death of the launch is u[j], and the slot a candidate inherits is found by pointer
//     jumping over "who died for whom";
//   * the evidence recursion over the deaths is affine in exp-space (pc_fast.hip pass B): block-wide
//     log-space prefix scans; the termination / update / max_ndead / nfail triggers are evaluated
//     for EVERY step from the prefix state, the first one that fires truncates the launch exactly
//     where the reference's loop would have stopped.
// One workgroup of 1024 threads, thread = nursery step (B <= 1024).  Ties in logL: snapshot points
// before candidates, earlier steps count as larger (what the strict `>` of run_time_info.f90:733 needs).
#include "pc_state.h"
#include "pc_keys.h"

typedef unsigned long long u64;
#define PAR_NT 1024
#define PAR_W 16      /* bitmap words = PAR_NT / 64 */

// number of set bits at positions < t in a 1024-bit map (t in [0, 1024])
__device__ __forceinline__ int prefix_bits(const u64 *words, int t)
{
    int c = 0;
    const int wq = t >> 6;
    #pragma unroll
    for (int x = 0; x < PAR_W; ++x) {
        u64 v = words[x];
        if (x > wq) v = 0ull; else if (x == wq) v &= (1ull << (t & 63)) - 1ull;
        c += __popcll(v);
    }
    return c;
}

// Sums of exponentials are carried as (m, s) pairs meaning m + log(s): combining two pairs costs one
// exp and no log (the log is taken once, where a value is needed).  Neutral element: (NEGBIG, 0).
__device__ __forceinline__ void ls_comb(double &m, double &s, double m2, double s2)
{
    const double e = exp(-fabs(m - m2));
    s = (m >= m2) ? s + s2 * e : s * e + s2;
    m = fmax(m, m2);
}
__device__ __forceinline__ double ls_val(double m, double s) { return s > 0.0 ? m + log(s) : NEGBIG; }

// inclusive block-wide prefix of two independent pair sequences (their exp chains overlap).
// nact = number of leading threads that carry data (the rest hold the neutral element).
__device__ __forceinline__ void block_scan_ls2(double &am, double &as, double &bm, double &bs, int lane, int wv, int nact, double *wtot)
{
    if (wv * 64 < nact) {
        for (int k = 1; k < 64; k <<= 1) {
            const double oam = __shfl_up(am, k), oas = __shfl_up(as, k), obm = __shfl_up(bm, k), obs = __shfl_up(bs, k);
            double nam = am, nas = as, nbm = bm, nbs = bs;
            ls_comb(nam, nas, oam, oas); ls_comb(nbm, nbs, obm, obs);
            if (lane >= k) { am = nam; as = nas; bm = nbm; bs = nbs; }
        }
    }
    if (lane == 63) { wtot[wv] = am; wtot[PAR_W + wv] = as; wtot[2 * PAR_W + wv] = bm; wtot[3 * PAR_W + wv] = bs; }
    __syncthreads();
    const int nw = (nact + 63) >> 6;
    if (wv > 0 && wv < nw) {
        // every wave reduces the totals of the waves before it (<= 15 pairs, 4 levels)
        const int l16 = lane & 15;
        const bool on = l16 < wv;
        double tam = on ? wtot[l16] : NEGBIG, tas = on ? wtot[PAR_W + l16] : 0.0;
        double tbm = on ? wtot[2 * PAR_W + l16] : NEGBIG, tbs = on ? wtot[3 * PAR_W + l16] : 0.0;
        for (int k = 1; k < PAR_W; k <<= 1) {
            const double oam = __shfl_xor(tam, k), oas = __shfl_xor(tas, k), obm = __shfl_xor(tbm, k), obs = __shfl_xor(tbs, k);
            ls_comb(tam, tas, oam, oas); ls_comb(tbm, tbs, obm, obs);
        }
        ls_comb(am, as, tam, tas); ls_comb(bm, bs, tbm, tbs);
    }
    __syncthreads();
}

__device__ __forceinline__ double block_scan_add(double v, int lane, int wv, double *wtot)
{
    for (int k = 1; k < 64; k <<= 1) { const double o = __shfl_up(v, k); if (lane >= k) v += o; }
    if (lane == 63) wtot[wv] = v;
    __syncthreads();
    double p = 0.0;
    for (int x = 0; x < wv; ++x) p += wtot[x];
    __syncthreads();
    return v + p;
}

__global__ __launch_bounds__(PAR_NT) void k_consume_par(PcState S)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ncap = S.Ncap, nr = S.nr, NS = (Ncap + 63) & ~63;
    PcCtl *ctl = S.ctl;
    const int T = ctl->i_nursery;            // steps of this launch: step t consumes chain T-1-t
    const int n = S.cl_n[0];
    // LDS carve (8-byte arrays first; typed pointer arithmetic only)
    u64 *sSortK = (u64 *)smem;               // [NS] snapshot keys, ascending
    u64 *cK = sSortK + NS;                   // [1024] candidate keys by step
    u64 *srtK = cK + PAR_NT;                 // [1024] sort buffer, then the accepted keys in ascending order
    u64 *uKey = srtK + PAR_NT;               // [1088] the K+1 smallest of snapshot u accepted
    u64 *Gm = uKey + PAR_NT + 64;            // [1024] in-chunk "earlier and larger" masks; then live log-sum-exp prefix
    u64 *accR = Gm + PAR_NT;                 // [16] accepted candidates, by rank
    u64 *amask = accR + PAR_W;               // [16] accepted candidates, by step
    u64 *vmask = amask + PAR_W;              // [16] steps of the current epoch
    double *sZi = (double *)(vmask + PAR_W); // [1024] logZ after death j
    double *wtot = sZi + PAR_NT;             // [64] scan scratch
    double *fin = wtot + 64;                 // [16] state after the last death of the launch
    int *sSort = (int *)(fin + 16);          // [NS] slots of the snapshot, ascending
    int *srtT = sSort + NS;                  // [1024]
    int *rnk = srtT + PAR_NT;                // [1024] rank of the candidate among the candidates
    int *rlo = rnk + PAR_NT;                 // [1024] snapshot points below the candidate
    int *uSrc = rlo + PAR_NT;                // [1088] >=0 snapshot slot, <0 -(1+step)
    int *slotA = uSrc + PAR_NT + 64;         // [1024] slot inherited by an accepted step
    int *parA = slotA + PAR_NT;              // [1024]
    int *accStep = parA + PAR_NT;            // [1024] step of the j-th acceptance
    int *ish = accStep + PAR_NT;             // [16]
    double *sLse = (double *)Gm;

    const int epoch = ctl->admin_epoch;
    const int ndead0 = ctl->ndead, fail0 = ctl->failures;
    const double logZ0 = ctl->logZ, logZ20 = ctl->logZ2;
    const double Xp0 = S.logXp[0], Zp0 = S.logZp[0], ZXp0 = S.logZXp[0], Zp20 = S.logZp2[0], ZpXp0 = S.logZpXp[0], XX0 = S.XpXq[0];
    const double lseRef0 = S.lse_ref[0], lseSum0 = S.lse_sum[0];
    const unsigned cuid = S.cl_uid[0];
    const double log2v = 0.6931471805599453;
    const double l0 = log((double)n + 0.0), l1 = log((double)n + 1.0), l2 = log((double)n + 2.0), d01 = l0 - l1, d02 = l0 - l2;

    long long cyc[9]; int ncy = 0;
    cyc[ncy++] = clock64();
    // ---- phase 0: stage the sorted snapshot and the candidates
    for (int i = tid; i < NS; i += PAR_NT) { sSortK[i] = (i < n) ? S.sort_key[i] : KEY_HUGE; sSort[i] = S.sort_slot[i]; }
    const bool inT = tid < T;
    const int w = T - 1 - tid;
    u64 ck = KEY_HUGE; bool valid = false; int nl = 0;
    if (inT) { ck = d2key(S.baby_logL[(size_t)w * nr + nr - 1]); valid = S.ch_epoch[w] == epoch; nl = S.ch_nlike[w]; }
    cK[tid] = ck; srtK[tid] = ck; srtT[tid] = tid;
    {
        const u64 vm = __ballot(valid);
        if (lane == 0) { vmask[wv] = vm; accR[wv] = 0ull; amask[wv] = 0ull; }
    }
    if (tid == 0) { ish[0] = (T << 2) | 3; ish[1] = 0; }
    __syncthreads();

    // ---- phase 1: snapshot points strictly below / not above the candidate
    int rl = 0, rp = 0;
    if (inT) {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sSortK[mid] < ck) lo = mid + 1; else hi = mid; }
        rl = lo; rp = lo;
        while (rp < n && sSortK[rp] == ck) rp++;
    }
    rlo[tid] = rl;

    cyc[ncy++] = clock64();
    // ---- phase 2: rank of every candidate (bitonic sort of (key, step); equal keys: earlier step = larger)
    for (int k = 2; k <= PAR_NT; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int l = tid ^ j;
            if (l > tid) {
                const bool up = (tid & k) == 0;
                const u64 a = srtK[tid], b = srtK[l]; const int ta = srtT[tid], tb = srtT[l];
                const bool gt = (a > b) || (a == b && ta < tb);
                if (gt == up) { srtK[tid] = b; srtK[l] = a; srtT[tid] = tb; srtT[l] = ta; }
            }
            __syncthreads();
        }
    rnk[srtT[tid]] = tid;
    __syncthreads();
    const int rho = rnk[tid];

    cyc[ncy++] = clock64();
    // ---- phase 3: acceptance.  In-chunk dependency masks by every wave, then wave 0 resolves the
    //      chunks in step order against the bitmap of accepted ranks.
    {
        u64 G = 0ull;
        #pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int rj = __builtin_amdgcn_readlane(rho, j);
            if (j < lane && rj > rho) G |= 1ull << j;
        }
        Gm[tid] = G;
    }
    __syncthreads();
    if (wv == 0) {
        const int nch = (T + 63) >> 6;
        volatile u64 *vacc = accR;
        for (int c = 0; c < nch; ++c) {
            const int t = c * 64 + lane;
            const int rho_t = rnk[t], r = rlo[t];
            const u64 Gt = Gm[t];
            const bool v = (vmask[c] >> lane) & 1ull;
            const int wq = rho_t >> 6, bq = rho_t & 63;
            int P = 0;                                    // accepted in earlier chunks with a larger rank
            #pragma unroll
            for (int x = 0; x < PAR_W; ++x) {
                const u64 word = vacc[x];
                const u64 m = (x > wq) ? ~0ull : ((x == wq) ? ((~0ull << bq) << 1) : 0ull);
                P += __popcll(word & m);
            }
            u64 am = __ballot(v && r > P);
            for (int it = 0; it < 66; ++it) {
                const bool a = v && r > P + __popcll(Gt & am);
                const u64 nm = __ballot(a);
                if (nm == am) break;
                am = nm;
            }
            if ((am >> lane) & 1ull) atomicOr(&accR[wq], 1ull << bq);
            if (lane == 0) amask[c] = am;
            __threadfence_block();
        }
    }
    __syncthreads();

    cyc[ncy++] = clock64();
    // ---- phase 4: counts
    const bool acc = (amask[wv] >> lane) & 1ull;
    const int kt = prefix_bits(amask, tid);               // acceptances (= deaths) before my step
    const int K = prefix_bits(amask, PAR_NT);
    const int vp = prefix_bits(vmask, tid);               // dead records written before my step
    int pos = 0;
    if (acc) {
        const int q = prefix_bits(accR, rho);             // rank among the accepted
        pos = q + rp;                                     // index in sorted(snapshot u accepted)
        accStep[kt] = tid; srtK[q] = ck;
    }
    __syncthreads();
    const u64 *aK = srtK;

    // ---- phase 5: u[0..K] = the K+1 smallest of snapshot u accepted
    for (int idx = tid; idx <= K && idx < n; idx += PAR_NT) {
        const u64 sk = sSortK[idx];
        int lo = 0, hi = K;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (aK[mid] < sk) lo = mid + 1; else hi = mid; }
        const int p2 = idx + lo;
        if (p2 <= K) { uKey[p2] = sk; uSrc[p2] = sSort[idx]; }
    }
    if (acc && pos <= K) { uKey[pos] = ck; uSrc[pos] = -(1 + tid); }
    __syncthreads();

    // ---- phase 6: contour of every step, dying point of every accepted step, inherited slot
    const u64 gk = inT ? uKey[kt] : KEY_HUGE;
    const int src = acc ? uSrc[kt] : 0;
    slotA[tid] = (acc && src >= 0) ? src : -1;
    parA[tid] = (acc && src < 0) ? (-src - 1) : -1;
    __syncthreads();
    for (int it = 0; it < 12; ++it) {
        const int p = parA[tid];
        int ns = -1, np = -1;
        const bool act = p >= 0;
        if (act) { const int pp = parA[p]; if (pp < 0) ns = slotA[p]; else np = pp; }
        if (!__syncthreads_or(act)) break;
        if (act) { if (np < 0) { slotA[tid] = ns; parA[tid] = -1; } else parA[tid] = np; }
        __syncthreads();
    }

    cyc[ncy++] = clock64();
    // ---- phase 7: evidence of the K deaths (thread j = j-th death), update_evidence (run_time_info.f90:211-296)
    const bool isd = tid < K;
    double L = NEGBIG, Ladd = NEGBIG;
    if (isd) { L = key2d(uKey[tid]); Ladd = key2d(cK[accStep[tid]]); }
    const double jd = (double)tid;
    const double Xb = Xp0 + jd * d01, XXb = XX0 + jd * d02;           // volumes before my death
    const double Sd = (jd + 1.0) * d01;
    // increments of logZ and of <Z X> (decay factored out) as pairs
    double tM = isd ? Xb + L - l1 : NEGBIG, tS = isd ? 1.0 : 0.0;
    double vM = isd ? (XXb + L + l0 - l1 - l2) - Sd : NEGBIG, vS = tS;
    block_scan_ls2(tM, tS, vM, vS, lane, wv, K, wtot);
    double ziM = tM, ziS = tS;
    ls_comb(ziM, ziS, logZ0, 1.0);
    const double Zi = ls_val(ziM, ziS);                                // logZ after my death
    double zxM = vM, zxS = vS, zpxM = vM, zpxS = vS;                   // <Z X> = Sd + (zxM + log zxS)
    ls_comb(zxM, zxS, ZXp0, 1.0); ls_comb(zpxM, zpxS, ZpXp0, 1.0);
    if (lane == 63) { wtot[wv] = zxM; wtot[PAR_W + wv] = zxS; wtot[2 * PAR_W + wv] = zpxM; wtot[3 * PAR_W + wv] = zpxS; }
    __syncthreads();
    double pzxM = __shfl_up(zxM, 1), pzxS = __shfl_up(zxS, 1), pzpxM = __shfl_up(zpxM, 1), pzpxS = __shfl_up(zpxS, 1);
    if (lane == 0) {
        pzxM = wv ? wtot[wv - 1] : ZXp0; pzxS = wv ? wtot[PAR_W + wv - 1] : 1.0;
        pzpxM = wv ? wtot[2 * PAR_W + wv - 1] : ZpXp0; pzpxS = wv ? wtot[3 * PAR_W + wv - 1] : 1.0;
    }
    __syncthreads();
    const double cz = log2v + XXb + 2 * L - l1 - l2;
    const double cw = log2v + L - l1 + jd * d01;                        // + <Z X> before my death
    double wM = isd ? cw + pzxM : NEGBIG, wS = isd ? pzxS : 0.0;
    double wpM = isd ? cw + pzpxM : NEGBIG, wpS = isd ? pzpxS : 0.0;
    if (isd) { ls_comb(wM, wS, cz, 1.0); ls_comb(wpM, wpS, cz, 1.0); }
    block_scan_ls2(wM, wS, wpM, wpS, lane, wv, K, wtot);
    // live log-sum-exp after every death (run_time_info.f90:683-709), one reference for the launch
    double refp;
    {
        const double mx = wave_max(Ladd);
        if (lane == 0) wtot[wv] = mx;
        __syncthreads();
        refp = lseRef0;
        for (int x = 0; x < PAR_W; ++x) refp = fmax(refp, wtot[x]);
        __syncthreads();
    }
    const double lse0 = lseSum0 * exp(lseRef0 - refp);
    const double de = isd ? exp(Ladd - refp) - exp(L - refp) : 0.0;
    const double lsei = lse0 + block_scan_add(de, lane, wv, wtot);
    sZi[tid] = Zi; sLse[tid] = lsei;
    __syncthreads();

    cyc[ncy++] = clock64();
    // ---- phase 8: the first step at which the reference's loop would have stopped
    int kupd = 0x7fffffff;                                // deaths until logXp <= logX_last_update + log(compression)
    {
        const double tx = ctl->logX_last_update + S.log_cf;
        double kf = ceil((Xp0 - tx) / (-d01));
        if (kf < 1.0) kf = 1.0;
        if (kf < 2.0e9) {
            kupd = (int)kf;
            while (kupd > 1 && Xp0 + (double)(kupd - 1) * d01 <= tx) kupd--;
            while (Xp0 + (double)kupd * d01 > tx) kupd++;
        }
    }
    auto check = [&](int t) __attribute__((always_inline)) {
        const int kb = prefix_bits(amask, t), vb = prefix_bits(vmask, t);
        const int ndead_b = ndead0 + vb;
        int tl = -1;                                      // last accepted step before t
        for (int x = PAR_W - 1; x >= 0; --x) {
            u64 word = amask[x];
            if (x > (t >> 6)) word = 0ull; else if (x == (t >> 6)) word &= (1ull << (t & 63)) - 1ull;
            if (word) { tl = x * 64 + 63 - __clzll((long long)word); break; }
        }
        const int fb = (tl >= 0) ? vb - prefix_bits(vmask, tl) - 1 : fail0 + vb;
        bool more = true;                                 // more_samples_needed (nested_sampling.F90:514-543)
        if (S.max_ndead == 0) more = false;
        else if (S.max_ndead > 0 && ndead_b >= S.max_ndead) more = false;
        else if (S.use_prec) {
            const double lse_b = kb ? sLse[kb - 1] : lse0, Zb = kb ? sZi[kb - 1] : logZ0;
            const double live = refp + log(lse_b) - l0 + Xp0 + (double)kb * d01;
            more = !(live < S.log_prec + Zb);
        }
        int code = 0x7fffffff;
        const bool tv = t < T && ((vmask[t >> 6] >> (t & 63)) & 1ull), ta = t < T && ((amask[t >> 6] >> (t & 63)) & 1ull);
        if (!more || fb > S.nfail) code = (t << 2) | 1;
        else if (tv && ndead_b >= S.Dcap) code = (t << 2) | 2;
        else if (ta && kb + 1 == kupd) code = ((t + 1) << 2) | 0;
        if (code != 0x7fffffff) atomicMin(&ish[0], code);
        return fb;
    };
    if (tid <= T) check(tid);
    if (T == PAR_NT && tid == 0) check(PAR_NT);
    __syncthreads();

    cyc[ncy++] = clock64();
    // ---- phase 9: truncate at the trigger and publish
    const int code = ish[0];
    const int ts = code >> 2, pri = code & 3;
    const int Kp = prefix_bits(amask, ts), vps = prefix_bits(vmask, ts);
    const int status = (pri == 0) ? PC_ST_UPDATE : (pri == 1) ? PC_ST_DONE : (pri == 2) ? PC_ST_ERROR : PC_ST_RUNNING;
    if (Kp > 0 && tid == Kp - 1) {                        // state after the last death of the launch
        double m = tM, q = tS;
        ls_comb(m, q, Zp0, 1.0);
        fin[0] = Zi; fin[1] = ls_val(m, q); fin[2] = Sd + ls_val(zxM, zxS); fin[3] = Sd + ls_val(zpxM, zpxS);
        ls_comb(wM, wS, logZ20, 1.0); ls_comb(wpM, wpS, Zp20, 1.0);
        fin[4] = ls_val(wM, wS); fin[5] = ls_val(wpM, wpS); fin[6] = lsei; fin[7] = L;
    }
    if (inT && tid < ts) {
        atomicAdd(&ish[1], nl);
        PcPlan *pw = S.plan + w;
        const double Lg = key2d(gk);
        pw->ph_cuid = cuid; pw->ph_count = -1; pw->ph_base = 0;
        pw->contour = valid ? Lg : PC_HUGE;               // chains of an old epoch get no phantoms
        pw->dead_idx = valid ? ndead0 + vp : -1;
        if (acc) {
            const double Xd = Xp0 + (double)kt * d01;
            pw->dead_src = (src >= 0) ? src : -(1 + (T - 1 - (-src - 1)));
            pw->logw = Xd - l1; pw->postX = Xd + d01; pw->postZ = sZi[kt]; pw->dead_cuid = cuid;
        } else if (valid) {                               // failed spawn (run_time_info.f90:781-785)
            pw->dead_src = -(1 + w); pw->logw = S.logzero; pw->postX = 0.0; pw->postZ = 0.0; pw->dead_cuid = 0xFFFFFFFFu;
        }
    }
    for (int s = tid; s < Ncap; s += PAR_NT) S.slot_src[s] = -1;
    if (lane == 0) accR[wv] = 0ull;
    __syncthreads();
    const bool accT = acc && tid < ts;
    if (accT) atomicOr(&accR[rho >> 6], 1ull << (rho & 63));
    __syncthreads();
    int pos2 = 0;
    if (accT) { const int q2 = prefix_bits(accR, rho); pos2 = q2 + rp; srtK[q2] = ck; }
    __syncthreads();
    if (accT && pos2 >= Kp) {                             // accepted and still alive at the end of the launch
        const int sl = slotA[tid];
        S.live_logL[sl] = key2d(ck); S.slot_src[sl] = w;
        S.sort_key[pos2 - Kp] = ck; S.sort_slot[pos2 - Kp] = sl;
    }
    // the sorted order of the new live set: the survivors merged into what is left of the snapshot
    for (int idx = tid; idx < n; idx += PAR_NT) {
        const u64 sk = sSortK[idx];
        int lo = 0, hi = Kp;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (aK[mid] < sk) lo = mid + 1; else hi = mid; }
        const int p2 = idx + lo;
        if (p2 >= Kp) { S.sort_key[p2 - Kp] = sk; S.sort_slot[p2 - Kp] = sSort[idx]; }
    }
    if (tid == 0) {
        const double Xp = Xp0 + (double)Kp * d01, XX = XX0 + (double)Kp * d02;
        const double lse_e = Kp ? fin[6] : lse0;
        const int usrc = uSrc[Kp];
        S.logLp[0] = key2d(uKey[Kp]); S.imin_slot[0] = (usrc >= 0) ? usrc : slotA[-usrc - 1];
        S.logXp[0] = Xp; S.XpXq[0] = XX;
        if (Kp) { S.logZp[0] = fin[1]; S.logZXp[0] = fin[2]; S.logZpXp[0] = fin[3]; S.logZp2[0] = fin[5]; S.death_thr[0] = fin[7]; }
        S.lse_ref[0] = refp; S.lse_sum[0] = lse_e;
        // consecutive failed spawns at the end of the launch
        int tl = -1;
        for (int x = PAR_W - 1; x >= 0; --x) {
            u64 word = amask[x];
            if (x > (ts >> 6)) word = 0ull; else if (x == (ts >> 6)) word &= (1ull << (ts & 63)) - 1ull;
            if (word) { tl = x * 64 + 63 - __clzll((long long)word); break; }
        }
        ctl->failures = (tl >= 0) ? vps - prefix_bits(vmask, tl) - 1 : fail0 + vps;
        ctl->status = status; ctl->error = (pri == 2) ? PC_ERR_DEAD_CAP : PC_ERR_NONE;
        ctl->i_nursery = T - ts; ctl->ndead = ndead0 + vps; ctl->seg_hi = T - 1; ctl->seg_lo = T - ts; ctl->cluster_deleted = 0;
        ctl->nlike += ish[1]; ctl->niter += ts;
        if (Kp) { ctl->logZ = fin[0]; ctl->logZ2 = fin[4]; }
        if (pri == 0) ctl->logX_last_update = Xp;
        if (S.use_prec) ctl->live_logZ = refp + log(lse_e) - l0 + Xp;
        cyc[ncy++] = clock64();
        for (int x = 0; x + 1 < ncy && x < 8; ++x) ctl->dbg[x] += cyc[x + 1] - cyc[x];
    }
}

static size_t par_lds(const PcState *S)
{
    const size_t NS = ((size_t)S->Ncap + 63) & ~(size_t)63;
    return 8 * (NS + 4 * PAR_NT + 64 + 3 * PAR_W + PAR_NT + 64 + 16) + 4 * (NS + 7 * PAR_NT + 64 + 16) + 64;
}

extern "C" int pc_par_fits(const PcState *S) { return par_lds(S) <= 160 * 1024 && S->B <= PAR_NT; }

extern "C" int pc_launch_consume_par(const PcState *S, hipStream_t st)
{
    const size_t sh = par_lds(S);
    if (sh > 160 * 1024 || S->B > PAR_NT) return 1;
    static size_t d = 0;
    if (sh > d) { (void)hipFuncSetAttribute((const void *)k_consume_par, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); d = sh; }
    hipLaunchKernelGGL(k_consume_par, dim3(1), dim3(PAR_NT), sh, st, *S);
    return 0;
}
