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

// the same from the exclusive per-word counts cum[0..PAR_W] (cum[PAR_W] = all bits)
__device__ __forceinline__ int prefix_cum(const u64 *words, const int *cum, int t)
{
    const int wq = t >> 6;
    return (wq >= PAR_W) ? cum[PAR_W] : cum[wq] + __popcll(words[wq] & ((1ull << (t & 63)) - 1ull));
}
// orders the LDS operations of ONE wave for the compiler; the hardware executes them in issue order
#define PAR_WAVE_ORDER() asm volatile("" ::: "memory")

// Sums of exponentials are carried as (m, s) pairs meaning m + log(s): combining two pairs costs one
// exp and no log (the log is taken once, where a value is needed).  Neutral element: (NEGBIG, 0).
__device__ __forceinline__ void ls_comb(double &m, double &s, double m2, double s2)
{
    const double e = exp(-fabs(m - m2));
    s = (m >= m2) ? s + s2 * e : s * e + s2;
    m = fmax(m, m2);
}
__device__ __forceinline__ double ls_val(double m, double s) { return s > 0.0 ? m + log(s) : NEGBIG; }

// Inclusive block-wide prefix of two independent pair sequences (their exp chains overlap), work
// efficient: 256 threads own four consecutive elements each (O(n) combines; a Hillis-Steele scan over
// 1024 threads costs O(n log n) exps and is bound by the fp64 rate of the one CU this kernel runs on).
// In-wave levels move data with DPP only (row_shr 1/2/4/8, then row_bcast15 / row_bcast31).
// Elements at and after nact hold the neutral element.  X0..X3: [1024] doubles of scratch each.
#define PAR_SCAN_LEVEL(AM, AS, BM, BS, CTRL, PRED) { \
        const double oam = dpp_f64<CTRL>(AM), oas = dpp_f64<CTRL>(AS), obm = dpp_f64<CTRL>(BM), obs = dpp_f64<CTRL>(BS); \
        double nam = AM, nas = AS, nbm = BM, nbs = BS; \
        ls_comb(nam, nas, oam, oas); ls_comb(nbm, nbs, obm, obs); \
        if (PRED) { AM = nam; AS = nas; BM = nbm; BS = nbs; } }
// Three sequences per pass: waves 0-3 scan the pair (a, b), waves 4-7 scan c at the same time (the other waves
// only park and fetch).  Xc / Xcs: scratch of the third sequence; pass nullptr to scan two sequences only.
#define PAR_SCAN_LEVEL1(AM, AS, CTRL, PRED) { \
        const double oam = dpp_f64<CTRL>(AM), oas = dpp_f64<CTRL>(AS); \
        double nam = AM, nas = AS; \
        ls_comb(nam, nas, oam, oas); \
        if (PRED) { AM = nam; AS = nas; } }
__device__ __forceinline__ void block_scan_ls3(double &am, double &as, double &bm, double &bs, double &cm, double &cs, int tid, int nact,
                                               double *X0, double *X1, double *X2, double *X3, double *Xc, double *Xcs, double *wtot)
{
    const int lane = tid & 63, wv = tid >> 6;
    const bool three = Xc != nullptr;
    X0[tid] = am; X1[tid] = as; X2[tid] = bm; X3[tid] = bs;
    if (three) { Xc[tid] = cm; Xcs[tid] = cs; }
    __syncthreads();
    double a_m[4], a_s[4], b_m[4], b_s[4];
    double tam = NEGBIG, tas = 0.0, tbm = NEGBIG, tbs = 0.0;
    const int t4 = (tid & 255) * 4;                          // my four elements (waves 4-7 mirror waves 0-3)
    const bool has = t4 < nact;
    const int lr = lane & 15;
    if (wv < 4) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            a_m[u] = has ? X0[t4 + u] : NEGBIG; a_s[u] = has ? X1[t4 + u] : 0.0;
            b_m[u] = has ? X2[t4 + u] : NEGBIG; b_s[u] = has ? X3[t4 + u] : 0.0;
        }
        #pragma unroll
        for (int u = 1; u < 4; ++u) { ls_comb(a_m[u], a_s[u], a_m[u - 1], a_s[u - 1]); ls_comb(b_m[u], b_s[u], b_m[u - 1], b_s[u - 1]); }
        tam = a_m[3]; tas = a_s[3]; tbm = b_m[3]; tbs = b_s[3];
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x111, lr >= 1)
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x112, lr >= 2)
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x114, lr >= 4)
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x118, lr >= 8)
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x142, (lane >> 4) & 1)      // row_bcast15: lane 15 of the previous row
        PAR_SCAN_LEVEL(tam, tas, tbm, tbs, 0x143, lane >= 32)           // row_bcast31: lane 31
        if (lane == 63) { wtot[wv] = tam; wtot[4 + wv] = tas; wtot[8 + wv] = tbm; wtot[12 + wv] = tbs; }
    } else if (three && wv < 8) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) { a_m[u] = has ? Xc[t4 + u] : NEGBIG; a_s[u] = has ? Xcs[t4 + u] : 0.0; }
        #pragma unroll
        for (int u = 1; u < 4; ++u) ls_comb(a_m[u], a_s[u], a_m[u - 1], a_s[u - 1]);
        tam = a_m[3]; tas = a_s[3];
        PAR_SCAN_LEVEL1(tam, tas, 0x111, lr >= 1)
        PAR_SCAN_LEVEL1(tam, tas, 0x112, lr >= 2)
        PAR_SCAN_LEVEL1(tam, tas, 0x114, lr >= 4)
        PAR_SCAN_LEVEL1(tam, tas, 0x118, lr >= 8)
        PAR_SCAN_LEVEL1(tam, tas, 0x142, (lane >> 4) & 1)
        PAR_SCAN_LEVEL1(tam, tas, 0x143, lane >= 32)
        if (lane == 63) { wtot[16 + wv - 4] = tam; wtot[20 + wv - 4] = tas; }
    }
    __syncthreads();
    if (wv < 4) {
        // exclusive prefix of my segment: the previous lane's inclusive value, then the waves before mine
        double eam = __shfl_up(tam, 1), eas = __shfl_up(tas, 1), ebm = __shfl_up(tbm, 1), ebs = __shfl_up(tbs, 1);
        if (lane == 0) { eam = NEGBIG; eas = 0.0; ebm = NEGBIG; ebs = 0.0; }
        for (int x = 0; x < wv; ++x) { ls_comb(eam, eas, wtot[x], wtot[4 + x]); ls_comb(ebm, ebs, wtot[8 + x], wtot[12 + x]); }
        if (has) {
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                ls_comb(a_m[u], a_s[u], eam, eas); ls_comb(b_m[u], b_s[u], ebm, ebs);
                X0[t4 + u] = a_m[u]; X1[t4 + u] = a_s[u]; X2[t4 + u] = b_m[u]; X3[t4 + u] = b_s[u];
            }
        }
    } else if (three && wv < 8) {
        double eam = __shfl_up(tam, 1), eas = __shfl_up(tas, 1);
        if (lane == 0) { eam = NEGBIG; eas = 0.0; }
        for (int x = 0; x < wv - 4; ++x) ls_comb(eam, eas, wtot[16 + x], wtot[20 + x]);
        if (has) {
            #pragma unroll
            for (int u = 0; u < 4; ++u) { ls_comb(a_m[u], a_s[u], eam, eas); Xc[t4 + u] = a_m[u]; Xcs[t4 + u] = a_s[u]; }
        }
    }
    __syncthreads();
    am = X0[tid]; as = X1[tid]; bm = X2[tid]; bs = X3[tid];
    if (three) { cm = Xc[tid]; cs = Xcs[tid]; }
    __syncthreads();
}
__device__ __forceinline__ void block_scan_ls2(double &am, double &as, double &bm, double &bs, int tid, int nact,
                                               double *X0, double *X1, double *X2, double *X3, double *wtot)
{
    double cm = NEGBIG, cs = 0.0;
    block_scan_ls3(am, as, bm, bs, cm, cs, tid, nact, X0, X1, X2, X3, nullptr, nullptr, wtot);
}

// inclusive prefix sums over a wavefront by DPP (row shifts, then the two row broadcasts): no LDS crossbar round trips
__device__ __forceinline__ int wave_scan_i32(int v, int lane)
{
    v += dpp_i32<0x111>(v); v += dpp_i32<0x112>(v); v += dpp_i32<0x114>(v); v += dpp_i32<0x118>(v);
    { const int t = dpp_i32<0x142>(v); if ((lane >> 4) & 1) v += t; }
    { const int t = dpp_i32<0x143>(v); if (lane >= 32) v += t; }
    return v;
}
__device__ __forceinline__ double wave_scan_f64(double v, int lane)
{
    v += dpp_f64<0x111>(v); v += dpp_f64<0x112>(v); v += dpp_f64<0x114>(v); v += dpp_f64<0x118>(v);
    { const double t = dpp_f64<0x142>(v); if ((lane >> 4) & 1) v += t; }
    { const double t = dpp_f64<0x143>(v); if (lane >= 32) v += t; }
    return v;
}
// The same prefixes in LINEAR space, for the launches (nearly all of them) whose terms lie within PAR_LIN_SPAN nats of
// the first one: every term is rescaled to that reference with ONE exp, the prefixes are plain sums, and the pair
// (reference, sum) is handed back -- a pair means m + log(s) whatever its scale, so nothing downstream can tell.  The pair
// scan above costs 16 wave-level log-add-exps per sequence on each SIMD (180 cycles of fp64 issue each: 24 k cycles of a
// launch for the five sequences); this one costs one exp.  Early in a run the newcomers of a launch can be thousands of
// nats above its first deaths: then the terms do not fit one scale, this function says so (false, nothing changed) and the
// caller scans pairs.
#define PAR_LIN_SPAN 300.0
// NC > 0: the prefixes are combined afterwards with launch-wide values cv[c] (the state before the launch) -- with ONE
// reference per sequence exp(-|reference - cv[c]|) is the same number in every thread: worked out here by NC lanes of the
// last wave between the two barriers and left in uexp[c] (cq[c] = the sequence cv[c] goes with).
template <int NSEQ, int NC>
__device__ __forceinline__ bool block_scan_lin(double (&m)[NSEQ], double (&s)[NSEQ], int tid, double *sc /* [(NSEQ + 1) * 16] */,
                                               const double *cv = nullptr, const int *cq = nullptr, double *uexp = nullptr)
{
    const int lane = tid & 63, wv = tid >> 6;
    double *ref = sc + NSEQ * PAR_W;                      // [NSEQ] references, then the verdict
    if (tid == 0) {
        #pragma unroll
        for (int q = 0; q < NSEQ; ++q) ref[q] = m[q];
        ref[NSEQ] = 0.0;
    }
    pc_lds_barrier();
    if (NC > 0 && wv == PAR_W - 1 && lane < NC) uexp[lane] = exp(-fabs(ref[cq[lane]] - cv[lane]));
    double inc[NSEQ], R[NSEQ];
    bool bad = false;
    #pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        R[q] = ref[q];
        const bool has = s[q] != 0.0 && m[q] > 0.5 * NEGBIG;
        const double d = m[q] - R[q];
        bad = bad || (has && !(fabs(d) <= PAR_LIN_SPAN));
        inc[q] = wave_scan_f64(has ? s[q] * exp(d) : 0.0, lane);
        if (lane == 63) sc[q * PAR_W + wv] = inc[q];
    }
    if (__ballot(bad) != 0ull && lane == 0) ref[NSEQ] = 1.0;
    pc_lds_barrier();
    const bool ok = ref[NSEQ] == 0.0;
    if (ok) {
        #pragma unroll
        for (int q = 0; q < NSEQ; ++q) {
            const double t = sc[q * PAR_W + (lane & 15)];
            double pre = t;
            pre += dpp_f64<0x111>(pre); pre += dpp_f64<0x112>(pre); pre += dpp_f64<0x114>(pre); pre += dpp_f64<0x118>(pre);
            s[q] = inc[q] + readlane_f64(pre - t, wv);
            m[q] = R[q];
        }
    }
    pc_lds_barrier();                                     // the scratch is free again
    return ok;
}

__device__ __forceinline__ double block_scan_add(double v, int lane, int wv, double *wtot)
{
    v = wave_scan_f64(v, lane);
    if (lane == 63) wtot[wv] = v;
    __syncthreads();
    double p = 0.0;
    for (int x = 0; x < wv; ++x) p += wtot[x];
    __syncthreads();
    return v + p;
}

__device__ __forceinline__ void consume_par_body(const PcState &S)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ncap = S.Ncap, nr = S.nr, NS = (Ncap + 63) & ~63;
    // The candidates sit at indices that depend on the number of chains left in the nursery, which is the whole batch for
    // every launch but a resumed one: requested now, next to the state words, instead of a memory round trip behind them.
    const int Bh = S.B;
    double bl = 0.0; int ep = 0, nl = 0;
    if (tid < Bh) { const int wh = Bh - 1 - tid; bl = S.baby_logL_T[(size_t)(nr - 1) * Bh + wh]; ep = S.ch_epoch[wh]; nl = S.ch_nlike[wh]; }
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
    int *hist = ish + 16;                    // [NS + 64] candidates per snapshot gap, then gap offsets
    double *X5 = (double *)(hist + NS + 64); // [1024] scan scratch (third sequence)
    double *sLse = (double *)Gm;

    const int epoch = ctl->admin_epoch;
    const int ndead0 = ctl->ndead, fail0 = ctl->failures, nph0 = ctl->nphantom;
    const long long nlike0 = ctl->nlike, niter0 = ctl->niter;
    const double logZ0 = ctl->logZ, logZ20 = ctl->logZ2;
    const double Xp0 = S.logXp[0], Zp0 = S.logZp[0], ZXp0 = S.logZXp[0], Zp20 = S.logZp2[0], ZpXp0 = S.logZpXp[0], XX0 = S.XpXq[0];
    const double lseRef0 = S.lse_ref[0], lseSum0 = S.lse_sum[0];
    const unsigned cuid = S.cl_uid[0];
    const double log2v = 0.6931471805599453;
    // exclusive per-word bit counts of the three bitmaps (a prefix count is then one word and one offset, not sixteen words)
    __shared__ int cumA[PAR_W + 1], cumV[PAR_W + 1], cumR[PAR_W + 1];
    // the launch-wide logarithms, one lane each (as per-thread constants they were 4 x 480 cycles of every wave, four waves to a SIMD)
    __shared__ double ulog[4];
    if (wv == PAR_W - 1 && lane < 4) ulog[lane] = log(lane == 3 ? lseSum0 : (double)n + (double)lane);

    long long cyc[9]; int ncy = 0;
    cyc[ncy++] = clock64();
    // ---- phase 0: stage the sorted snapshot and the candidates
    // (slot records of the launch before: reset now, a whole kernel ahead of the stores of phase 9 that follow them)
    for (int s = tid; s < Ncap; s += PAR_NT) { S.slot_src[s] = -1; if (S.defer_update) S.slot_step[s] = -1; }
    for (int i = tid; i < NS; i += PAR_NT) { sSortK[i] = (i < n) ? S.sort_key[i] : KEY_HUGE; sSort[i] = S.sort_slot[i]; }
    const bool inT = tid < T;
    const int w = T - 1 - tid;
    u64 ck = KEY_HUGE; bool valid = false;
    if (T != Bh && inT) { bl = S.baby_logL_T[(size_t)(nr - 1) * S.B + w]; ep = S.ch_epoch[w]; nl = S.ch_nlike[w]; }
    if (inT) { ck = d2key(bl); valid = ep == epoch; } else nl = 0;
    cK[tid] = ck;
    for (int i = tid; i < NS + 64; i += PAR_NT) hist[i] = 0;
    {
        const u64 vm = __ballot(valid);
        if (lane == 0) { vmask[wv] = vm; accR[wv] = 0ull; amask[wv] = 0ull; }
    }
    if (tid == 0) { ish[0] = (T << 2) | 3; ish[1] = 0; ish[3] = 0; }
    __syncthreads();
    const double l0 = ulog[0], l1 = ulog[1], l2 = ulog[2], d01 = l0 - l1, d02 = l0 - l2;

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
    // ---- phase 2: rank of every candidate among the candidates (equal keys: earlier step = larger).
    //      Candidates are already bucketed by the snapshot gap they fall in (rl): rank = candidates in
    //      lower gaps + candidates of the same gap that compare lower.
    int myoff = 0;
    if (inT) myoff = atomicAdd(&hist[rl], 1);            // arrival order inside a gap: placement only
    __syncthreads();
    {
        const int nb = n + 2;                             // gaps 0..n, plus one sentinel that receives T
        const int per = (nb + PAR_NT - 1) / PAR_NT, b0 = tid * per;
        int sum = 0;
        for (int x = 0; x < per; ++x) if (b0 + x < nb) sum += hist[b0 + x];
        int inc = sum;
        inc = wave_scan_i32(inc, lane);
        if (lane == 63) accStep[wv] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int x = 0; x < wv; ++x) run += accStep[x];
        for (int x = 0; x < per; ++x) if (b0 + x < nb) { const int cnt = hist[b0 + x]; hist[b0 + x] = run; run += cnt; }
    }
    __syncthreads();
    if (inT) srtT[hist[rl] + myoff] = tid;
    __syncthreads();
    int rho = tid;
    if (inT) {
        const int g0 = hist[rl], g1 = hist[rl + 1];
        int below = 0;
        for (int x = g0; x < g1; ++x) {
            const int o = srtT[x];
            const u64 ko = cK[o];
            below += (ko < ck) || (ko == ck && o > tid);
        }
        rho = g0 + below;
    }
    rnk[tid] = rho;

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
        int *cum = accStep;                               // [16] accepted candidates in the bitmap words before x
        if (lane < PAR_W) cum[lane] = 0;
        __threadfence_block();
        int rho_t = rnk[lane], r = rlo[lane], tot = 0;
        u64 Gt = Gm[lane];
        for (int c = 0; c < nch; ++c) {
            // operands of the next chunk travel while this one is resolved
            const int tn = (c + 1 < nch) ? (c + 1) * 64 + lane : lane;
            const int nrho = rnk[tn], nr2 = rlo[tn];
            const u64 nG = Gm[tn];
            const bool v = (vmask[c] >> lane) & 1ull;
            const int wq = rho_t >> 6, bq = rho_t & 63;
            // accepted in earlier chunks with a larger rank = all of them minus those at or below my rank
            const int P = tot - cum[wq] - __popcll(accR[wq] & (((1ull << bq) << 1) - 1ull));
            u64 am = __ballot(v && r > P);
            for (int it = 0; it < 66; ++it) {
                const bool a = v && r > P + __popcll(Gt & am);
                const u64 nm = __ballot(a);
                if (nm == am) break;
                am = nm;
            }
            if ((am >> lane) & 1ull) atomicOr(&accR[wq], 1ull << bq);
            if (lane == 0) amask[c] = am;
            tot += __popcll(am);
            PAR_WAVE_ORDER();                             // (LDS operations of one wave execute in issue order: nothing to wait for)
            {   // refresh the per-word offsets
                int cw = (lane < PAR_W) ? __popcll(accR[lane]) : 0, inc = cw;      // (sixteen words = one DPP row: row shifts, no LDS crossbar)
                inc += dpp_i32<0x111>(inc); inc += dpp_i32<0x112>(inc); inc += dpp_i32<0x114>(inc); inc += dpp_i32<0x118>(inc);
                if (lane < PAR_W) cum[lane] = inc - cw;
            }
            PAR_WAVE_ORDER();
            rho_t = nrho; r = nr2; Gt = nG;
        }
        {   // the offsets every thread counts from in the phases below
            const int ca = (lane < PAR_W) ? __popcll(amask[lane]) : 0, cv = (lane < PAR_W) ? __popcll(vmask[lane]) : 0,
                      cr = (lane < PAR_W) ? __popcll(accR[lane]) : 0;
            int ia = ca, iv = cv, ir = cr;
            ia += dpp_i32<0x111>(ia); ia += dpp_i32<0x112>(ia); ia += dpp_i32<0x114>(ia); ia += dpp_i32<0x118>(ia);
            iv += dpp_i32<0x111>(iv); iv += dpp_i32<0x112>(iv); iv += dpp_i32<0x114>(iv); iv += dpp_i32<0x118>(iv);
            ir += dpp_i32<0x111>(ir); ir += dpp_i32<0x112>(ir); ir += dpp_i32<0x114>(ir); ir += dpp_i32<0x118>(ir);
            if (lane < PAR_W) { cumA[lane] = ia - ca; cumV[lane] = iv - cv; cumR[lane] = ir - cr; }
            if (lane == PAR_W - 1) { cumA[PAR_W] = ia; cumV[PAR_W] = iv; cumR[PAR_W] = ir; }
        }
    }
    __syncthreads();

    cyc[ncy++] = clock64();
    // ---- phase 4: counts
    const u64 amw = amask[wv], ltm = (1ull << lane) - 1ull;
    const bool acc = (amw >> lane) & 1ull;
    const int kt = cumA[wv] + __popcll(amw & ltm);        // acceptances (= deaths) before my step
    const int K = cumA[PAR_W];
    const int vp = cumV[wv] + __popcll(vmask[wv] & ltm);  // dead records written before my step
    int pos = 0, q = 0;
    if (acc) {
        q = cumR[rho >> 6] + __popcll(accR[rho >> 6] & ((1ull << (rho & 63)) - 1ull));   // rank among the accepted
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
#ifdef PAR_DBG_EVID
    long long ecy[9]; ecy[0] = clock64();
#endif
    // ---- phase 7: evidence of the K deaths (thread j = j-th death), update_evidence (run_time_info.f90:211-296)
    const bool isd = tid < K;
    double L = NEGBIG, Ladd = NEGBIG;
    if (isd) { L = key2d(uKey[tid]); Ladd = key2d(cK[accStep[tid]]); }
    const double jd = (double)tid;
    const double Xb = Xp0 + jd * d01, XXb = XX0 + jd * d02;           // volumes before my death
    const double Sd = (jd + 1.0) * d01;
#ifdef PAR_DBG_EVID
    ecy[1] = clock64();
#endif
    // increments of logZ and of <Z X> (decay factored out) as pairs
    double tM = isd ? Xb + L - l1 : NEGBIG, tS = isd ? 1.0 : 0.0;
    double vM = isd ? (XXb + L + l0 - l1 - l2) - Sd : NEGBIG, vS = tS;
    double *X0 = (double *)srtK, *X1 = (double *)Gm, *X2 = sZi, *X3 = (double *)rnk;   // scratch until the end of this phase
    // live log-sum-exp after every death (run_time_info.f90:683-709) as a (max, sum) pair: a death removes
    // exp(L), the newcomer adds exp(Ladd).  A single reference for the whole launch underflows when the
    // newcomers are hundreds of nats above the points they replace (early in a run, narrow posteriors).
    // It rides in the same pass as the two evidence sequences, on waves 4-7.
    double lsM = NEGBIG, lsS = 0.0;
    if (isd) {                                        // (one of the two exponentials is exp(0))
        const double dl = L - Ladd, e = exp(-fabs(dl));
        lsM = fmax(Ladd, L); lsS = (dl <= 0.0) ? 1.0 - e : e - 1.0;
    }
    __shared__ double lsc[4 * PAR_W];
    const bool lin_on = !(S.ablate & 16);             // (bit 4: pair scans only -- tests compare the two paths on the same run)
    __shared__ double ucv[4], uexp[4];
    __shared__ int ucq[4];
    bool lin1 = false;
    // log-add-exp of a prefix pair with a launch-wide value: on the linear path the exponential is launch-wide too
    auto comb_u = [&](double &m, double &s, double m2, double s2, int c) __attribute__((always_inline)) {
        if (lin1) { const double e = uexp[c]; s = (m >= m2) ? s + s2 * e : s * e + s2; m = fmax(m, m2); }
        else ls_comb(m, s, m2, s2);
    };
    int npair = 0;                                    // scans of this launch that went the pair way
    {
        double mm[3] = {tM, vM, lsM}, ss[3] = {tS, vS, lsS};
        if (tid < 4) { ucv[tid] = tid == 0 ? logZ0 : tid == 1 ? ZXp0 : tid == 2 ? ZpXp0 : lseRef0; ucq[tid] = tid == 0 ? 0 : tid == 3 ? 2 : 1; }
        lin1 = lin_on && block_scan_lin<3, 4>(mm, ss, tid, lsc, ucv, ucq, uexp);     // (its first barrier publishes ucv / ucq)
        if (lin1) { tM = mm[0]; tS = ss[0]; vM = mm[1]; vS = ss[1]; lsM = mm[2]; lsS = ss[2]; }
        else {
            npair++;
            __syncthreads();                          // every thread has read its candidate key: cK becomes scratch
            block_scan_ls3(tM, tS, vM, vS, lsM, lsS, tid, K, X0, X1, X2, X3, (double *)cK, X5, wtot);
        }
    }
#ifdef PAR_DBG_EVID
    ecy[2] = clock64();
#endif
    const bool wact = wv * 64 < K;                    // waves past the last death have nothing to work out (their values are read by nobody)
    double Zi = NEGBIG;
    double zxM = vM, zxS = vS, zpxM = vM, zpxS = vS;                   // <Z X> = Sd + (zxM + log zxS)
    if (wact) {
        double ziM = tM, ziS = tS;
        comb_u(ziM, ziS, logZ0, 1.0, 0);
        Zi = ls_val(ziM, ziS);                                         // logZ after my death
        comb_u(zxM, zxS, ZXp0, 1.0, 1); comb_u(zpxM, zpxS, ZpXp0, 1.0, 2);
    }
    if (lane == 63) { wtot[wv] = zxM; wtot[PAR_W + wv] = zxS; wtot[2 * PAR_W + wv] = zpxM; wtot[3 * PAR_W + wv] = zpxS; }
    __syncthreads();
    double pzxM = __shfl_up(zxM, 1), pzxS = __shfl_up(zxS, 1), pzpxM = __shfl_up(zpxM, 1), pzpxS = __shfl_up(zpxS, 1);
    if (lane == 0) {
        pzxM = wv ? wtot[wv - 1] : ZXp0; pzxS = wv ? wtot[PAR_W + wv - 1] : 1.0;
        pzpxM = wv ? wtot[2 * PAR_W + wv - 1] : ZpXp0; pzpxS = wv ? wtot[3 * PAR_W + wv - 1] : 1.0;
    }
    __syncthreads();
#ifdef PAR_DBG_EVID
    ecy[3] = clock64();
#endif
    const double cz = log2v + XXb + 2 * L - l1 - l2;
    const double cw = log2v + L - l1 + jd * d01;                        // + <Z X> before my death
    double wM = isd ? cw + pzxM : NEGBIG, wS = isd ? pzxS : 0.0;
    double wpM = isd ? cw + pzpxM : NEGBIG, wpS = isd ? pzpxS : 0.0;
    if (isd) { ls_comb(wM, wS, cz, 1.0); ls_comb(wpM, wpS, cz, 1.0); }
    {
        double mm[2] = {wM, wpM}, ss[2] = {wS, wpS};
        if (lin_on && block_scan_lin<2, 0>(mm, ss, tid, lsc)) { wM = mm[0]; wS = ss[0]; wpM = mm[1]; wpS = ss[1]; }
        else { npair++; block_scan_ls2(wM, wS, wpM, wpS, tid, K, X0, X1, X2, X3, wtot); }
    }
#ifdef PAR_DBG_EVID
    ecy[4] = clock64();
#endif
    if (wact) comb_u(lsM, lsS, lseRef0, lseSum0, 3);  // + the live set before the launch
#ifdef PAR_DBG_EVID
    ecy[5] = clock64(); ecy[6] = ecy[5];
#endif
    const double lse_log0 = lseRef0 + ulog[3];
    const double lsei = wact ? lsM + log(lsS) : NEGBIG;
    sZi[tid] = Zi; sLse[tid] = lsei;
    __syncthreads();
#ifdef PAR_DBG_EVID
    ecy[7] = clock64();
#endif

    cyc[ncy++] = clock64();
    // ---- phase 8: the first step at which the reference's loop would have stopped
    const bool defer = S.defer_update != 0;
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
    // last accepted step before my wave's first step, dead records before every step
    int tlw = -1;
    for (int x = wv - 1; x >= 0; --x) { const u64 word = amask[x]; if (word) { tlw = x * 64 + 63 - __clzll((long long)word); break; } }
    parA[tid] = vp;
    __syncthreads();
    auto check = [&](int t, int kb, int vb, int tl) __attribute__((always_inline)) {
        const int ndead_b = ndead0 + vb;
        const int fb = (tl >= 0) ? vb - parA[tl] - 1 : fail0 + vb;      // consecutive failed spawns before t
        bool more = true;                                 // more_samples_needed (nested_sampling.F90:514-543)
        if (S.max_ndead == 0) more = false;
        else if (S.max_ndead > 0 && ndead_b >= S.max_ndead) more = false;
        else if (S.use_prec) {
            const double lse_b = kb ? sLse[kb - 1] : lse_log0, Zb = kb ? sZi[kb - 1] : logZ0;     // log of the sum
            const double live = lse_b - l0 + Xp0 + (double)kb * d01;
            more = !(live < S.log_prec + Zb);
        }
        int code = 0x7fffffff;
        const bool tv = t < T && ((vmask[t >> 6] >> (t & 63)) & 1ull), ta = t < T && ((amask[t >> 6] >> (t & 63)) & 1ull);
        if (!more || fb > S.nfail) code = (t << 2) | 1;
        else if (tv && ndead_b >= S.Dcap) code = (t << 2) | 2;
        else if (!defer && ta && kb + 1 == kupd) code = ((t + 1) << 2) | 0;
        if (code != 0x7fffffff) atomicMin(&ish[0], code);
    };
    {
        const u64 mw = amask[wv] & ((1ull << lane) - 1ull);
        const int tl = mw ? wv * 64 + 63 - __clzll((long long)mw) : tlw;
        if (tid < T) check(tid, kt, vp, tl);
        if (tid == T - 1) {                               // the state after the last step of the nursery
            const int tl2 = acc ? tid : tl;
            check(T, kt + (acc ? 1 : 0), vp + (valid ? 1 : 0), tl2);
        }
    }
    __syncthreads();

    cyc[ncy++] = clock64();
    // ---- phase 9: truncate at the trigger and publish
#ifdef PAR_DBG_PUBLISH
    long long pcy[8]; pcy[0] = clock64();
#endif
    const int code = ish[0];
    const int ts = code >> 2, pri = code & 3;
    const int Kp = prefix_cum(amask, cumA, ts), vps = prefix_cum(vmask, cumV, ts);
    const int status = (pri == 0) ? PC_ST_UPDATE : (pri == 1) ? PC_ST_DONE : (pri == 2) ? PC_ST_ERROR : PC_ST_RUNNING;
    // The host can have the outcome NOW: everything it decides on (status, chains left, counters, update marks) is known, and
    // what is left of this kernel (plans, merged order, state: 13 us) and the row kernel behind it need nothing from the host.
    // Stamped here, the next round's launches are in the queue by the time the device gets to them.
    __shared__ unsigned note_early[8];
    __shared__ int marks_sh[2];                           // last mark (deaths), number of marks
    if (tid == 0) {
        int Kl = 0, marks = 0;
        if (defer && Kp >= kupd) {
            // Update triggers passed by this launch (nested_sampling.F90:321: logXp <= logX_last_update + log(compression),
            // tested after every death): the first after kupd deaths, every later one relative to the volume at the one
            // before.  The update is made once, afterwards, for the state at the last of them.
            Kl = kupd; marks = 1;
            for (;;) {
                // the first kn > Kl whose volume is below the trigger: the test is monotone in kn, so start at the estimate
                // Kl + log(cf) / (l0 - l1) and settle with the test itself (a linear search from Kl + 1 was 20 k cycles of
                // this one lane per launch)
                const double txn = (Xp0 + (double)Kl * d01) + S.log_cf;
                const double est = (double)Kl + S.log_cf / d01;
                int kn = !(est < (double)(Kp + 1)) ? Kp + 1 : (int)est;
                if (kn < Kl + 1) kn = Kl + 1;
                while (kn > Kl + 1 && !(Xp0 + (double)(kn - 1) * d01 > txn)) kn--;
                while (kn <= Kp && Xp0 + (double)kn * d01 > txn) kn++;
                if (kn > Kp) break;
                Kl = kn; marks++;
            }
        }
        marks_sh[0] = Kl; marks_sh[1] = marks;
        const unsigned err = (pri == 2) ? PC_ERR_DEAD_CAP : PC_ERR_NONE;
        note_early[0] = (unsigned)status | (err << 8) | ((unsigned)(marks > 0) << 17) | ((unsigned)(marks > 0x3FFF ? 0x3FFF : marks) << 18);
        // lived deaths until the next update trigger, as the state stands after this launch (PcCtl::upd_in: the host decides by it
        // whether to enqueue the next nursery's sampling before it has seen that round's outcome)
        int upd_in_e;
        {
            const double Xe = Xp0 + (double)Kp * d01, lastu = marks > 0 ? Xp0 + (double)Kl * d01 : ((pri == 0) ? Xe : ctl->logX_last_update);
            const double left = (Xe - (lastu + S.log_cf)) / (-d01);
            upd_in_e = left > 0.0 ? (left < 65535.0 ? (int)ceil(left) : 65535) : 1;
        }
        note_early[5] = (unsigned)upd_in_e;
        note_early[1] = ((unsigned)(T - ts) & 0xFFFFu) | ((unsigned)upd_in_e << 16); note_early[2] = (unsigned)(ndead0 + vps);
        note_early[3] = (unsigned)(S.pool ? S.pool_base + S.pool_rows : nph0 + ts * nr);
        note_early[4] = 1u | ((unsigned)(ctl->ncluster_dead & 0xFFFF) << 16);
    }
    pc_lds_barrier();
    if (S.ctl_host && tid < PC_NOTE_WORDS)
        ((volatile unsigned long long *)S.ctl_host)[tid] = ((unsigned long long)S.notify_seq << 32) | note_early[tid];
    if (Kp > 0 && tid == Kp - 1) {                        // state after the last death of the launch
        double m = tM, q = tS;
        ls_comb(m, q, Zp0, 1.0);
        fin[0] = Zi; fin[1] = ls_val(m, q); fin[2] = Sd + ls_val(zxM, zxS); fin[3] = Sd + ls_val(zpxM, zpxS);
        ls_comb(wM, wS, logZ20, 1.0); ls_comb(wpM, wpS, Zp20, 1.0);
        fin[4] = ls_val(wM, wS); fin[5] = ls_val(wpM, wpS); fin[6] = lsM; fin[7] = L; fin[8] = lsS;
    }
    {
        int nls = (inT && tid < ts) ? nl : 0, nlf = (inT && tid < ts && !acc) ? nl : 0;
        nls = (int)wave_sum<4>((double)nls); nlf = (int)wave_sum<4>((double)nlf);       // (exact: counts below 2^31)
        if (lane == 0 && nls) atomicAdd(&ish[1], nls);
        if (lane == 0 && nlf) atomicAdd(&ish[3], nlf);
    }
    if (inT && tid < ts) {
        // the chain's record: failed spawn (run_time_info.f90:781-785) unless accepted; a chain of an old epoch has no dead
        // record at all (dead_idx -1: nobody reads the rest) and gets no phantoms
        PcPlanHead h;
        h.ph_cuid = cuid; h.contour = valid ? key2d(gk) : PC_HUGE; h.dead_idx = valid ? ndead0 + vp : -1;
        h.dead_src = -(1 + w); h.logw = S.logzero; h.postX = 0.0; h.postXs = 1.0; h.postZ = 0.0; h.dead_cuid = 0xFFFFFFFFu;
        if (acc) {
            const double Xd = Xp0 + (double)kt * d01;
            h.dead_src = (src >= 0) ? src : -(1 + (T - 1 - (-src - 1)));
            if (S.pool && src >= 0) S.slot_dead[src] = w;    // (the apply kernel moves that row out before the slot's new occupant moves in)
            h.logw = Xd - l1; h.postX = Xd + d01; h.postZ = sZi[kt]; h.dead_cuid = cuid;
        }
        // phantoms of the consumed chains (run_time_info.f90:747-757): the babies above the contour their chain was
        // consumed at.  Which babies those are is a comparison with the contour that the row-copy kernel, one workgroup per
        // chain on the whole chip, does for itself: every consumed chain gets a region of nr rows of the phantom array in
        // consumption order (ph_count = -2), its phantoms land in it at their own index and the other rows of the region
        // carry the cluster id PC_CUID_NONE, which no clean keeps -- a layout as deterministic as the packed one, without the
        // masks, the prefix sums and 39 strided loads per chain on this one CU (10 us of a 60 us launch).
        h.ph_base = S.pool ? S.pool_base + w * nr : nph0 + tid * nr; h.ph_count = -2;
        *static_cast<PcPlanHead *>(S.plan + w) = h;       // four 16-byte stores
    }
#ifdef PAR_DBG_PUBLISH
    pcy[1] = clock64();
#endif
    if (tid == 0) ish[2] = S.pool ? S.pool_base + S.pool_rows : nph0 + ts * nr;               // rows in use after this launch
    const bool accT = acc && tid < ts;
    int pos2 = pos;
    if (ts >= T) {
        // nothing truncated (every launch but a run's last, with the deferred update): the accepted keep the ranks of phase 4
        if (acc) srtK[q] = ck;
    } else {
        if (lane == 0) accR[wv] = 0ull;
        pc_lds_barrier();                                 // (LDS only: the plan's stores to HBM need not have landed)
        if (accT) atomicOr(&accR[rho >> 6], 1ull << (rho & 63));
        pc_lds_barrier();
        if (accT) { const int q2 = prefix_bits(accR, rho); pos2 = q2 + rp; srtK[q2] = ck; }
    }
#ifdef PAR_DBG_PUBLISH
    pcy[2] = clock64(); pcy[3] = pcy[2];
#endif
    pc_lds_barrier();                                     // (LDS only: the plan's stores to HBM need not have landed)
    if (accT && pos2 >= Kp) {                             // accepted and still alive at the end of the launch
        const int sl = slotA[tid];
        S.live_logL[sl] = key2d(ck); S.slot_src[sl] = w;
        if (defer) S.slot_step[sl] = tid;
        S.sort_key[pos2 - Kp] = ck; S.sort_slot[pos2 - Kp] = sl;
    }
#ifdef PAR_DBG_PUBLISH
    pcy[4] = clock64();
#endif
    // the sorted order of the new live set: the survivors merged into what is left of the snapshot
    for (int idx = tid; idx < n; idx += PAR_NT) {
        const u64 sk = sSortK[idx];
        int lo = 0, hi = Kp;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (aK[mid] < sk) lo = mid + 1; else hi = mid; }
        const int p2 = idx + lo;
        if (p2 >= Kp) { S.sort_key[p2 - Kp] = sk; S.sort_slot[p2 - Kp] = sSort[idx]; }
    }
#ifdef PAR_DBG_PUBLISH
    pcy[5] = clock64();
#endif
    if (tid == 0) {
        const double Xp = Xp0 + (double)Kp * d01, XX = XX0 + (double)Kp * d02;
        const double lse_m = Kp ? fin[6] : lseRef0, lse_s = Kp ? fin[8] : lseSum0;
        const int usrc = uSrc[Kp];
        S.logLp[0] = key2d(uKey[Kp]); S.imin_slot[0] = (usrc >= 0) ? usrc : slotA[-usrc - 1];
        S.logXp[0] = Xp; S.XpXq[0] = XX;
        if (Kp) { S.logZp[0] = fin[1]; S.logZXp[0] = fin[2]; S.logZpXp[0] = fin[3]; S.logZp2[0] = fin[5]; S.death_thr[0] = fin[7]; }
        S.lse_ref[0] = lse_m; S.lse_sum[0] = lse_s;
        // consecutive failed spawns at the end of the launch
        int tl = -1;
        for (int x = PAR_W - 1; x >= 0; --x) {
            u64 word = amask[x];
            if (x > (ts >> 6)) word = 0ull; else if (x == (ts >> 6)) word &= (1ull << (ts & 63)) - 1ull;
            if (word) { tl = x * 64 + 63 - __clzll((long long)word); break; }
        }
        ctl->failures = (tl >= 0) ? vps - prefix_cum(vmask, cumV, tl) - 1 : fail0 + vps;
        ctl->status = status; ctl->error = (pri == 2) ? PC_ERR_DEAD_CAP : PC_ERR_NONE;
        ctl->i_nursery = T - ts; ctl->ndead = ndead0 + vps; ctl->seg_hi = T - 1; ctl->seg_lo = T - ts; ctl->cluster_deleted = 0;
        ctl->nlike = nlike0 + ish[1]; ctl->niter = niter0 + ts; ctl->nphantom = ish[2]; ctl->nlike_failed += ish[3];
        if (Kp) { ctl->logZ = fin[0]; ctl->logZ2 = fin[4]; }
        if (pri == 0) ctl->logX_last_update = Xp;
        ctl->upd_pending = 0; ctl->upd_marks = 0;
        ctl->spec_ok = (status == PC_ST_RUNNING && marks_sh[1] == 0 && T - ts == 0 && pri != 2) ? (int)(ctl->batch_id + 1u) : -1;
        if (marks_sh[1] > 0) {
            const int Kl = marks_sh[0], marks = marks_sh[1];
            ctl->upd_pending = 1; ctl->upd_marks = marks;
            ctl->upd_tmark = accStep[Kl - 1] + 1; ctl->upd_T = T; ctl->upd_ts = ts; ctl->upd_nph0 = S.pool ? S.pool_base : nph0;
            ctl->upd_thr = key2d(uKey[Kl - 1]); ctl->upd_keep_thr = (Kp > Kl) ? 1 : 0;
            ctl->logX_last_update = Xp0 + (double)Kl * d01;
        }
        ctl->upd_in = (int)note_early[5];
        if (S.use_prec) ctl->live_logZ = (Kp ? sLse[Kp - 1] : lseRef0 + ulog[3]) - l0 + Xp;   // (= lse_m + log(lse_s), taken in phase 7)
        cyc[ncy++] = clock64();
#ifdef PAR_NO_DBG
#elif defined(PAR_DBG_EVID)
        for (int x = 0; x < 7; ++x) ctl->dbg[x] += ecy[x + 1] - ecy[x];
#elif defined(PAR_DBG_PUBLISH)
        pcy[6] = clock64();
        for (int x = 0; x < 6; ++x) ctl->dbg[x] += pcy[x + 1] - pcy[x];
#else
        for (int x = 0; x + 1 < ncy && x < 8; ++x) ctl->dbg[x] += cyc[x + 1] - cyc[x];
        ctl->dbg[7] += npair;                             // (PC_DEBUG=4: evidence scans that did not fit one scale)
#endif
    }
}
__global__ __launch_bounds__(PAR_NT) void k_consume_par(PcState S) { consume_par_body(S); }
__global__ __launch_bounds__(PAR_NT) void k_consume_par_many(const PcManyRec *R) { consume_par_body(pc_many_state(R, blockIdx.y)); }


// ------------------------------------------------------------------------------------------
// Kill-off (nested_sampling.F90:381-384): every remaining live point dies, lowest first.  Death i of
// the sorted live set leaves n-i points behind, so volumes are prefix sums of log((n-i)/(n-i+1)) and the
// evidence is the same pair scan as above, 1024 deaths per pass with the state carried between passes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void final_par_body(const PcState &S)
{
    __shared__ __attribute__((aligned(16))) double X0[PAR_NT], X1[PAR_NT], X2[PAR_NT], X3[PAR_NT];
    __shared__ double wtot[64];
    __shared__ double carry[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    PcCtl *ctl = S.ctl;
    const int n0 = S.cl_n[0], nT = S.nT, ndead0 = ctl->ndead;
    const unsigned cuid = S.cl_uid[0];
    const double log2v = 0.6931471805599453;
    if (ndead0 + n0 > S.Dcap) { if (tid == 0) { ctl->status = PC_ST_ERROR; ctl->error = PC_ERR_DEAD_CAP; } return; }
    if (tid == 0) {
        carry[0] = ctl->logZ; carry[1] = ctl->logZ2; carry[2] = S.logXp[0]; carry[3] = S.XpXq[0]; carry[4] = S.logZp[0];
        carry[5] = S.logZXp[0]; carry[6] = S.logZp2[0]; carry[7] = S.logZpXp[0]; carry[8] = S.death_thr[0];
    }
    __syncthreads();
    for (int base = 0; base < n0; base += PAR_NT) {
        const int m = min(PAR_NT, n0 - base), i = base + tid;
        const bool on = tid < m;
        const double logZ0 = carry[0], logZ20 = carry[1], Xp0 = carry[2], XX0 = carry[3], Zp0 = carry[4], ZXp0 = carry[5],
                     Zp20 = carry[6], ZpXp0 = carry[7];
        const int slot = on ? S.sort_slot[i] : 0;
        const double L = on ? key2d(S.sort_key[i]) : NEGBIG;
        const int myn = n0 - i;                                   // live points before my death
        const double a0 = on ? log((double)myn) : 0.0, a1 = on ? log((double)myn + 1.0) : 0.0, a2 = on ? log((double)myn + 2.0) : 0.0;
        const double e01 = a0 - a1, e02 = a0 - a2;
        const double sx = block_scan_add(e01, lane, wv, wtot), sxx = block_scan_add(e02, lane, wv, wtot);   // inclusive
        const double Xb = Xp0 + (sx - e01), XXb = XX0 + (sxx - e02), Sd = sx;
        double tM = on ? Xb + L - a1 : NEGBIG, tS = on ? 1.0 : 0.0;
        double vM = on ? (XXb + L + a0 - a1 - a2) - Sd : NEGBIG, vS = tS;
        block_scan_ls2(tM, tS, vM, vS, tid, m, X0, X1, X2, X3, wtot);
        double ziM = tM, ziS = tS;
        ls_comb(ziM, ziS, logZ0, 1.0);
        const double Zi = ls_val(ziM, ziS);
        double zxM = vM, zxS = vS, zpxM = vM, zpxS = vS;
        ls_comb(zxM, zxS, ZXp0, 1.0); ls_comb(zpxM, zpxS, ZpXp0, 1.0);
        // <Z X> before my death: the previous death's value (its own decay prefix), the carry for the first
        X0[tid] = zxM; X1[tid] = zxS; X2[tid] = zpxM; X3[tid] = zpxS;
        __syncthreads();
        const double SdPrev = Sd - e01;
        double pzxM = tid ? X0[tid - 1] : ZXp0, pzxS = tid ? X1[tid - 1] : 1.0;
        double pzpxM = tid ? X2[tid - 1] : ZpXp0, pzpxS = tid ? X3[tid - 1] : 1.0;
        __syncthreads();
        const double cz = log2v + XXb + 2 * L - a1 - a2;
        const double cw = log2v + L - a1 + SdPrev;
        double wM = on ? cw + pzxM : NEGBIG, wS = on ? pzxS : 0.0;
        double wpM = on ? cw + pzpxM : NEGBIG, wpS = on ? pzpxS : 0.0;
        if (on) { ls_comb(wM, wS, cz, 1.0); ls_comb(wpM, wpS, cz, 1.0); }
        block_scan_ls2(wM, wS, wpM, wpS, tid, m, X0, X1, X2, X3, wtot);
        if (on) {
            const int di = ndead0 + i;
            S.dead_logw[di] = Xb - a1; S.dead_postX[di] = Xb + e01; S.dead_postZ[di] = Zi;
            S.dead_cuid[di] = cuid; S.dead_entry[di] = S.live_entry[slot];
            S.live_logL[slot] = PC_HUGE; S.live_cluster[slot] = -1; S.slot_src[slot] = -1;
        }
        __syncthreads();                                          // everybody has read the carry
        if (tid == m - 1) {
            double a = tM, b = tS;
            ls_comb(a, b, Zp0, 1.0);
            ls_comb(wM, wS, logZ20, 1.0); ls_comb(wpM, wpS, Zp20, 1.0);
            carry[0] = Zi; carry[1] = ls_val(wM, wS); carry[2] = Xp0 + sx; carry[3] = XX0 + sxx; carry[4] = ls_val(a, b);
            carry[5] = Sd + ls_val(zxM, zxS); carry[6] = ls_val(wpM, wpS); carry[7] = Sd + ls_val(zpxM, zpxS); carry[8] = L;
        }
        __syncthreads();
        // rows: one wave per row, 16 rows at a time
        for (int r = wv; r < m; r += PAR_NT / 64) {
            const int sl = S.sort_slot[base + r];
            const double *row = S.live + (size_t)sl * nT;
            double *dst = S.dead + (size_t)(ndead0 + base + r) * nT;
            for (int e = lane; e < nT; e += 64) dst[e] = row[e];
        }
    }
    if (tid == 0) {
        const int ncd = ctl->ncluster_dead;
        if (ncd < S.maxc_dead) { S.logZp_dead[ncd] = carry[4]; S.logZp2_dead[ncd] = carry[6]; S.cl_uid_dead[ncd] = cuid; }
        S.logLp[0] = PC_HUGE; S.imin_slot[0] = -1; S.logXp[0] = carry[2]; S.XpXq[0] = carry[3]; S.logZp[0] = carry[4];
        S.logZXp[0] = carry[5]; S.logZp2[0] = carry[6]; S.logZpXp[0] = carry[7]; S.death_thr[0] = carry[8]; S.cl_n[0] = 0;
        ctl->status = PC_ST_DONE; ctl->error = PC_ERR_NONE; ctl->ndead = ndead0 + n0; ctl->ncluster = 0; ctl->ncluster_dead = ncd + 1;
        ctl->logZ = carry[0]; ctl->logZ2 = carry[1]; ctl->cluster_deleted = 0;
    }
}
__global__ __launch_bounds__(PAR_NT) void k_final_par(PcState S) { final_par_body(S); }
__global__ __launch_bounds__(PAR_NT) void k_final_par_many(const PcManyRec *R) { final_par_body(pc_many_state(R, blockIdx.y)); }


static size_t par_lds(const PcState *S)
{
    const size_t NS = ((size_t)S->Ncap + 63) & ~(size_t)63;
    return 8 * (NS + 4 * PAR_NT + 64 + 3 * PAR_W + PAR_NT + 64 + 16 + PAR_NT) + 4 * (2 * NS + 7 * PAR_NT + 2 * 64 + 16) + 64;
}

#define PAR_STATIC_LDS 2048   /* the kernel's __shared__ tables, on top of the dynamic carve */
extern "C" int pc_par_fits(const PcState *S) { return par_lds(S) + PAR_STATIC_LDS <= 160 * 1024 && S->B <= PAR_NT; }

extern "C" int pc_launch_consume_par(const PcState *S, hipStream_t st)
{
    const size_t sh = par_lds(S);
    if (sh + PAR_STATIC_LDS > 160 * 1024 || S->B > PAR_NT) return 1;
    pc_need_dyn_lds((const void *)k_consume_par, sh);
    hipLaunchKernelGGL(k_consume_par, dim3(1), dim3(PAR_NT), sh, st, *S);
    return 0;
}

// the same launch for R runs of one shape at once (blockIdx.y = run; dR: device array of their records)
extern "C" int pc_launch_consume_par_many(const PcState *S, const PcManyRec *dR, int R, hipStream_t st)
{
    const size_t sh = par_lds(S);
    if (sh + PAR_STATIC_LDS > 160 * 1024 || S->B > PAR_NT) return 1;
    pc_need_dyn_lds((const void *)k_consume_par_many, sh);
    hipLaunchKernelGGL(k_consume_par_many, dim3(1, R), dim3(PAR_NT), sh, st, dR);
    return 0;
}

extern "C" int pc_launch_final_par(const PcState *S, hipStream_t st)
{
    hipLaunchKernelGGL(k_final_par, dim3(1), dim3(PAR_NT), 0, st, *S);
    return 0;
}
extern "C" int pc_launch_final_par_many(const PcManyRec *dR, int R, hipStream_t st)
{
    hipLaunchKernelGGL(k_final_par_many, dim3(1, R), dim3(PAR_NT), 0, st, dR);
    return 0;
}
