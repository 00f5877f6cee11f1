// pc_update.hip -- the update step (nested_sampling.F90:323-368 minus files and clustering) as FOUR short kernels for the
// common case: one cluster, nDims < 32.
//
//   clean_phantoms (run_time_info.f90:820-877): phantoms below the last death are dropped, the others are compacted;
//   calculate_covmats (:601-641): mean and population covariance of live + phantom cube coordinates;
//   calc_cholesky (utils.F90:621-649), with the scaled-identity fallback of :633-638.
//
// The general path (pc_contract.hip: k_clean_flag / k_scan_blocks / k_clean_scatter / k_reset_thresholds /
// k_cov_mean_partial / k_fold_partials / k_cov_partial / k_fold_partials / k_cov_final_chol) is nine launches of 4 - 25 us
// with the host's launch rate between them: ~140 us per update, 31 updates per run at the metric configuration = a fifth
// of the run.  Here:
//   k_upd_flag   which phantoms survive, survivors per block of 256 rows
//   k_upd_move   every block: its offset (sum of the counts before it), compaction of its survivors into the alternate
//                buffer, and -- in the same pass, while the rows are in L2 -- first and second moments of its rows' cube
//                coordinates about a SHIFT; further blocks do the same for the live points
//   k_upd_fold   the blocks' moments added in groups of sixteen
//   k_upd_final  the groups' moments added, mean / covariance / Cholesky factor, new shift, thresholds reset
// One pass instead of two (mean, then centred products) needs the shift: the moments are taken about the PREVIOUS update's
// mean (the cube centre at first), which the new mean is within a fraction of a standard deviation of, so the
// subtraction cov = M2/n - delta delta^T loses a digit at most, not the six it would lose about the origin.
#include "pc_state.h"
#include <cstdio>

#define UPD_ROWS 256
#define UPD_NT 256

// the wave's rows selected by `mask` (row of bit b = src0 + b*nT), in order: optionally copied to consecutive rows at dst;
// of those also in `mmask`, the first D elements minus the shift (and a one) go to consecutive tile rows, in order;
// sixteen rows in flight, lane = element
__device__ __forceinline__ void upd_stage_masked(const double *src0, unsigned long long mask, unsigned long long mmask, double *dst,
                                                 double *tile, int TS, const double *sh, int D, int nT, int lane)
{
    const int ne = dst ? nT : D + 1;
    int n = 0;
    while (mask) {
        int idx[16], cnt = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            idx[u] = 0;
            if (mask) { idx[u] = __ffsll((long long)mask) - 1; mask &= mask - 1; cnt = u + 1; }
        }
        for (int e = lane; e < ne; e += 64) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (u < cnt) v[u] = (e < nT) ? src0[(size_t)idx[u] * nT + e] : 0.0;
            if (dst) {
#pragma unroll
                for (int u = 0; u < 16; ++u) if (u < cnt) dst[(size_t)(n + u) * nT + e] = v[u];
            }
            if (e <= D) {                              // column D carries a one: its products are the first moments and the count
                const double s = e < D ? sh[e] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u < cnt && ((mmask >> idx[u]) & 1ull))
                        tile[(size_t)__popcll(mmask & ((1ull << idx[u]) - 1ull)) * TS + e] = e < D ? v[u] - s : 1.0;
            }
        }
        n += cnt;
    }
}

// def (here and below): the update is made AFTER the launch of the parallel contraction that passed its trigger, for the
// state at the mark the launch recorded (PcCtl::upd_*): the threshold of the clean is the logL of the death at the mark;
// the moments leave out what came after it -- phantoms of later chains, live points accepted later -- and take in the
// points that were alive then and have died since (their rows are in the dead array)
__global__ __launch_bounds__(UPD_NT) void k_upd_flag(PcState S, int nph, unsigned char *keep, int *blk_count, int def)
{
    __shared__ int cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned uid0 = S.cl_uid[0];
    const double thr = def ? S.ctl->upd_thr : S.death_thr[0];
    const int j = blockIdx.x * UPD_ROWS + tid;
    bool k = false;
    if (j < nph) { k = (S.ph_cuid[j] == uid0) && !(S.ph_logL[j] < thr); keep[j] = k ? 1 : 0; }
    const unsigned long long m = __ballot(k);
    if (lane == 0) cnt[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) blk_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// partial record of a block: [npair second moments (a <= b, row-major upper triangle) | D first moments | count], padded to E
//
// The moments are X^T X of the block's member rows X = [cube - shift | 1] (n x (D+1)) on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64; operand maps as in k_cov_partial).  Not for the flops: with one product per thread every FMA
// costs two LDS reads, and twelve waves per CU doing that are bound by the LDS pipe (10 us per block); a 16x16x4 tile
// needs two reads per 2048 flops.  Wave w takes the rows 4w .. 4w+3, 4w+16 .., the four waves' tiles are added in order.
typedef double upd_v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(UPD_NT) void k_upd_move(PcState S, int nph, int nblk, const unsigned char *keep, const int *blk_count,
                                                    double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2,
                                                    int *d_total, const double *shift, double *part, int E, int def, int nlb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int red[UPD_NT];
    __shared__ int wcnt[4], mcnt[4];
    const int tmark = def ? S.ctl->upd_tmark : 0x7fffffff, nph0u = def ? S.ctl->upd_nph0 : 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, D = S.D, nT = S.nT;
    const int ncol = D + 1, TS = (ncol + 1) | 1;   // odd row stride: the four row groups of an operand hit different banks
    double *tile = (double *)smem;                 // [UPD_ROWS + 16][TS] member rows first: cube - shift, then a column of ones
    double *sh = tile + (size_t)(UPD_ROWS + 16) * TS;     // [D]
    const int blk = blockIdx.x;
    const bool is_ph = blk < nblk, is_live = !is_ph && blk < nblk + nlb;
    for (int d = tid; d < D; d += UPD_NT) sh[d] = shift[d];
    int n = 0;                                     // member rows of this block
    if (is_ph) {
        // everything this thread will need from memory is requested before the first wait
        const int j = blk * UPD_ROWS + tid;
        int s = 0;
        for (int b = tid; b < blk; b += UPD_NT) s += blk_count[b];
        const bool k = (j < nph) && keep[j];
        double pl = 0.0; unsigned pc = 0u; unsigned long long pu = 0ull;
        if (j < nph) { pl = S.ph_logL[j]; pc = S.ph_cuid[j]; pu = S.ph_uid[j]; }
        // moments: the survivors that were phantoms at the mark (rows of regions of later chains stay, but do not count)
        const bool km = k && (j < nph0u || (j - nph0u) / S.nr < tmark);
        // ---- offset of this block's survivors = survivors of the blocks before it (integer sum: any order)
        red[tid] = s;
        const unsigned long long m = __ballot(k), mm = __ballot(km);
        if (lane == 0) { wcnt[wv] = __popcll(m); mcnt[wv] = __popcll(mm); }
        __syncthreads();
        for (int o = UPD_NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const int off0 = red[0];
        if (blk == nblk - 1 && tid == 0) { const int tot = off0 + blk_count[blk]; *d_total = tot; S.ctl->nphantom = tot; }
        // ---- compaction, row order kept
        int woff = 0, moff = 0;
        for (int x = 0; x < wv; ++x) { woff += wcnt[x]; moff += mcnt[x]; }
        const int pos = woff + __popcll(m & ((1ull << lane) - 1ull));
        if (k) { phL2[off0 + pos] = pl; phC2[off0 + pos] = pc; phU2[off0 + pos] = pu; }
        upd_stage_masked(S.phantom + (size_t)(blk * UPD_ROWS + wv * 64) * nT, m, mm, ph2 + (size_t)(off0 + woff) * nT, tile + (size_t)moff * TS, TS, sh, D, nT, lane);
        n = mcnt[0] + mcnt[1] + mcnt[2] + mcnt[3];
    } else if (is_live) {
        // ---- live points of slots [r0, r0 + UPD_ROWS)
        const int r0 = (blk - nblk) * UPD_ROWS, r = r0 + tid;
        const bool k = (r < S.Ncap) && S.live_cluster[r] == 0 && (!def || S.slot_step[r] < tmark);
        const unsigned long long m = __ballot(k);
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int x = 0; x < wv; ++x) woff += wcnt[x];
        upd_stage_masked(S.live + (size_t)(r0 + wv * 64) * nT, m, m, nullptr, tile + (size_t)woff * TS, TS, sh, D, nT, lane);
        n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    } else {
        // ---- def: points alive at the mark that died later in the launch: the dead rows of the steps t >= tmark whose
        //      dying point was a snapshot point or the newcomer of a step before the mark
        const PcCtl *ctl = S.ctl;
        const int T = ctl->upd_T, ts = ctl->upd_ts;
        const int t = tmark + (blk - nblk - nlb) * UPD_ROWS + tid;
        long long di = -1;
        if (t < ts) {
            const PcPlan *pw = S.plan + (T - 1 - t);
            const int src = pw->dead_src;
            const bool existed = src >= 0 || (T - 1 - (-src - 1)) < tmark;
            if (pw->dead_idx >= 0 && pw->logw > S.logzero && existed) di = pw->dead_idx;
        }
        const unsigned long long m = __ballot(di >= 0);
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int x = 0; x < wv; ++x) woff += wcnt[x];
        unsigned long long mb = m;
        int row = woff;
        while (mb) {                                   // rows are scattered in the dead array: one at a time, lane = coordinate
            const int b = __ffsll((long long)mb) - 1; mb &= mb - 1;
            const long long d = __shfl(di, b);
            if (lane <= D) tile[(size_t)row * TS + lane] = lane < D ? S.dead[(size_t)d * nT + lane] - sh[lane] : 1.0;
            row++;
        }
        n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
    // rows n .. next multiple of 16: zero (the matrix cores take four rows at a time, four waves)
    const int n16 = (n + 15) & ~15;
    for (int e = tid; e < (n16 - n) * TS; e += UPD_NT) tile[(size_t)n * TS + e] = 0.0;
    __syncthreads();
    // ---- X^T X, upper triangle of 16 x 16 tiles: (0,0), and with more than 16 columns (0,1), (1,1)
    const int nt = (ncol + 15) >> 4;               // 1 or 2
    upd_v4d acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = upd_v4d{0.0, 0.0, 0.0, 0.0};
    {
        const int c0 = lane & 15, c1 = 16 + (lane & 15);
        const bool on1 = nt > 1 && c1 < ncol;
        const double *p0 = tile + (size_t)(4 * wv + (lane >> 4)) * TS + c0;
        const bool on0 = c0 < ncol;
        for (int q0 = 4 * wv; q0 < n16; q0 += 16, p0 += (size_t)16 * TS) {
            const double x0 = on0 ? p0[0] : 0.0, x1 = on1 ? p0[16] : 0.0;
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0], 0, 0, 0);
            if (nt > 1) {
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc[2], 0, 0, 0);
            }
        }
    }
    __syncthreads();                               // the tile is no longer read: its memory takes the four waves' results
    double *wres = tile;                           // [4 waves][3 tiles][256]
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wres[((size_t)wv * 3 + t) * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[t][r];
    __syncthreads();
    double *out = part + (size_t)blk * E;
    const int npair = D * (D + 1) / 2;
    for (int p = tid; p < npair + D + 1; p += UPD_NT) {
        int a, b;                                  // element (a, b), a <= b, of the (D+1) x (D+1) moment matrix
        if (p < npair) { a = 0; int q = p; while (q >= D - a) { q -= D - a; a++; } b = a + q; }
        else if (p < npair + D) { a = p - npair; b = D; }
        else { a = D; b = D; }
        const int t = (a >> 4) + (b >> 4), idx = (a & 15) * 16 + (b & 15);       // tile (0,0) -> 0, (0,1) -> 1, (1,1) -> 2
        out[p] = ((wres[(size_t)(0 * 3 + t) * 256 + idx] + wres[(size_t)(1 * 3 + t) * 256 + idx]) +
                  (wres[(size_t)(2 * 3 + t) * 256 + idx] + wres[(size_t)(3 * 3 + t) * 256 + idx]));
    }
}

// partial records folded in groups of sixteen (one batch of loads per thread), in block order
#define UPD_FOLD 16
__global__ __launch_bounds__(256) void k_upd_fold(const double *part, int nb, int E, double *part2)
{
    const int g = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += 256) {
        double t[UPD_FOLD];
#pragma unroll
        for (int u = 0; u < UPD_FOLD; ++u) t[u] = (g * UPD_FOLD + u < nb) ? part[(size_t)(g * UPD_FOLD + u) * E + e] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < UPD_FOLD; ++u) s += t[u];
        part2[(size_t)g * E + e] = s;
    }
}

// fold + mean + covariance + Cholesky; one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_upd_final(PcState S, int nb, const double *part, int E, double *shift, int def)
{
    __shared__ double acc[4][256];                 // E <= 256 * 2: entries beyond 256 take a second round
    __shared__ double A[32 * 32], L[32 * 32], mu[32];
    __shared__ int bad;
    const int tid = threadIdx.x, D = S.D, npair = D * (D + 1) / 2, nE = npair + D + 1;
    __shared__ double tot[544];
    for (int e0 = 0; e0 < nE; e0 += 256) {
        const int e = e0 + (tid & 255), h = tid >> 8;         // four threads per entry: blocks h, h+4, ... in that order
        double s = 0.0;
        if (e < nE) {
            for (int k = h; k < nb; k += 4 * 16) {            // sixteen loads in flight, added in block order (no serial tail)
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = (k + 4 * u < nb) ? part[(size_t)(k + 4 * u) * E + e] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) s += t[u];
            }
        }
        acc[h][tid & 255] = s;
        __syncthreads();
        if (tid < 256 && e < nE) tot[e] = (acc[0][tid] + acc[1][tid]) + (acc[2][tid] + acc[3][tid]);
        __syncthreads();
    }
#ifdef UPD_DBG
    const long long f0 = clock64();
#endif
    if (tid == 0) bad = 0;
    const double n = tot[npair + D];
    if (tid < D) mu[tid] = tot[npair + tid] / n;               // delta = mean - shift
    __syncthreads();
    for (int p = tid; p < D * D; p += 1024) {
        const int a = p / D, b = p % D, lo = a < b ? a : b, hi = a < b ? b : a;
        const int idx = lo * D - lo * (lo - 1) / 2 + (hi - lo);
        const double c = tot[idx] / n - mu[lo] * mu[hi];       // population normalisation, run_time_info.f90:634
        A[p] = c; L[p] = 0.0;
        S.cov[p] = c;
    }
    __syncthreads();
#ifdef UPD_DBG
    const long long f1 = clock64();
#endif
    // calc_cholesky (utils.F90:621-649) by one wavefront, lane j = row j held in registers, right-looking: column i is
    // scaled by 1/sqrt(a_ii), then every later column k loses l_ji l_ki -- the same products subtracted in the same order
    // (ascending i) as the reference's dot products accumulate them; compile-time indices only (fully unrolled)
    if (tid < 64) {
        double a[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = (tid < D && k < D) ? A[tid * D + k] : 0.0;
        bool fail = false;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < D && !fail) {
                const double aii = readlane_f64(a[i], i);
                if (aii <= 0.0) fail = true;
                else {
                    const double lii = sqrt(aii);
                    const double lji = (tid == i) ? lii : a[i] / lii;
                    if (tid >= i) a[i] = lji;
#pragma unroll
                    for (int k = i + 1; k < 32; ++k)
                        if (k < D) { const double lki = readlane_f64(lji, k); if (tid > i) a[k] -= lji * lki; }
                }
            }
        }
        if (fail) { if (tid == 0) bad = 1; }
        else if (tid < D) {
#pragma unroll
            for (int k = 0; k < 32; ++k) if (k < D) L[tid * D + k] = (k <= tid) ? a[k] : 0.0;
        }
    }
    __syncthreads();
    if (bad) {                                                 // no Cholesky factor: scaled identity (utils.F90:633-638)
        double tr = 0.0;
        for (int k = 0; k < D; ++k) tr += A[k * D + k];
        for (int p = tid; p < D * D; p += 1024) L[p] = (p / D == p % D) ? sqrt(tr) : 0.0;
        __syncthreads();
    }
    for (int p = tid; p < D * D; p += 1024) S.chol[p] = L[p];
    if (tid < D) shift[tid] += mu[tid];                        // the next update's moments are taken about this mean
    // every survivor is above the last death (k_reset_thresholds) -- unless the launch went on dying after the mark:
    // then the threshold of the next clean stays the logL of its last death
    if (tid == 0) { if (!(def && S.ctl->upd_keep_thr)) S.death_thr[0] = -PC_HUGE; if (def) S.ctl->upd_pending = 0; }
#ifdef UPD_DBG
    if (tid == 0) { const long long f2 = clock64(); S.ctl->gen_cyc[2] += f1 - f0; S.ctl->gen_cyc[3] += f2 - f1; S.ctl->nn_walks += f0; }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// 32 <= nDims <= 128 (one cluster): the same single pass -- compaction of the surviving phantoms and the shifted moments
// of live + phantom coordinates while the rows go by -- with the moment matrix in 16 x 16 tiles spread over the four
// waves of a workgroup.  The general path read every phantom row three times (scatter, mean, centred products: 2.5 ms of a
// 3.4 ms update at nDims = 100, nlive 5000) and gathered the 800 coordinate bytes out of 1616-byte rows for the matrix
// cores; here a row is read once, whole, written once, and its coordinates are in LDS when the products are formed.
//   k_upd_flag, k_scan_blocks (offsets of the 256-row blocks), k_upd_move_w, k_upd_fold, k_upd_final_w, k_cov_final_chol
// Workgroups are persistent: each walks its share of the 64-row chunks with the accumulator tiles (upper triangle of
// NT x NT tiles of X^T X, X = [cube - shift | 1 | 0...], pairs q = wave, wave + 4, ...) in registers and writes ONE record.
#define UPDW_ROWS 64
template <int NT, int W>
__device__ __forceinline__ void updw_accumulate(const double *tile, int TS, int n16, int li, int lk, upd_v4d (&acc)[(NT * (NT + 1) / 2 + 3) / 4])
{
    for (int ks = 0; ks < (n16 >> 2); ++ks) {
        const double *row = tile + (size_t)(4 * ks + lk) * TS + li;
        double x[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] = row[16 * t];
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = ti; tj < NT; ++tj, ++q)
                if ((q & 3) == W) acc[q >> 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ti], x[tj], acc[q >> 2], 0, 0, 0);
    }
}
template <int NT, int W>
__device__ __forceinline__ void updw_store(double *out, int D, int li, int lk, const upd_v4d (&acc)[(NT * (NT + 1) / 2 + 3) / 4])
{
    const int npair = D * (D + 1) / 2;
    int q = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj, ++q)
            if ((q & 3) == W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = 16 * ti + lk + 4 * r, b = 16 * tj + li;
                    if (a <= b && b <= D) {
                        const int p = (b < D) ? a * D - a * (a - 1) / 2 + (b - a) : (a < D ? npair + a : npair + D);
                        out[p] = acc[q >> 2][r];
                    }
                }
            }
}

template <int NT>
__global__ __launch_bounds__(256) void k_upd_move_w(PcState S, int nph, int nblk, const unsigned char *keep, const int *blk_off,
                                                    double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2,
                                                    const double *shift, double *part, int E, int def, int nlc, int ndc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TS = 16 * NT + ((NT & 1) ? 0 : 16);                  // = 16 mod 32 doubles: the four row groups of an operand
    constexpr int NPW = (NT * (NT + 1) / 2 + 3) / 4;                   //   split over both halves of the LDS banks
    __shared__ unsigned long long m64[4], mm64[4];
    __shared__ int wcnt[4];
    const int tmark = def ? S.ctl->upd_tmark : 0x7fffffff, nph0u = def ? S.ctl->upd_nph0 : 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4, D = S.D, nT = S.nT;
    double *tile = (double *)smem;                                      // [UPDW_ROWS + 16][TS]
    double *sh = tile + (size_t)(UPDW_ROWS + 16) * TS;                  // [D]
    for (int d = tid; d < D; d += 256) sh[d] = shift[d];
    for (int e = tid; e < (UPDW_ROWS + 16) * TS; e += 256) tile[e] = 0.0;     // (the columns past the ones stay zero)
    upd_v4d acc[NPW];
#pragma unroll
    for (int t = 0; t < NPW; ++t) acc[t] = upd_v4d{0.0, 0.0, 0.0, 0.0};
    const int npc = 4 * nblk, total = npc + nlc + ndc;
    __syncthreads();
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        int n = 0;                                                      // member rows of this chunk
        if (c < npc) {
            // ---- 64 phantom rows: sub-chunk `sub` of the 256-row block `blk` (whose offset the scan left in blk_off)
            const int blk = c >> 2, sub = c & 3, j = blk * 256 + tid;
            const bool k = (j < nph) && keep[j];
            const bool km = k && (j < nph0u || (j - nph0u) / S.nr < tmark);       // a phantom at the mark (see k_upd_move)
            const unsigned long long m = __ballot(k), mm = __ballot(km);
            if (lane == 0) { m64[wv] = m; mm64[wv] = mm; }
            __syncthreads();
            int off = blk_off[blk];
            for (int x = 0; x < sub; ++x) off += __popcll(m64[x]);
            if (wv == sub && k) {
                const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                phL2[pos] = S.ph_logL[j]; phC2[pos] = S.ph_cuid[j]; phU2[pos] = S.ph_uid[j];
            }
            const unsigned long long ms = m64[sub], mms = mm64[sub], below = (1ull << (16 * wv)) - 1ull;
            upd_stage_masked(S.phantom + (size_t)(blk * 256 + sub * 64 + 16 * wv) * nT, (ms >> (16 * wv)) & 0xFFFFull, (mms >> (16 * wv)) & 0xFFFFull,
                             ph2 + (size_t)(off + __popcll(ms & below)) * nT, tile + (size_t)__popcll(mms & below) * TS, TS, sh, D, nT, lane);
            n = __popcll(mms);
        } else if (c < npc + nlc) {
            // ---- live points of slots [r0, r0 + 64)
            const int r0 = (c - npc) * UPDW_ROWS, r = r0 + tid;
            const bool k = tid < UPDW_ROWS && r < S.Ncap && S.live_cluster[r] == 0 && (!def || S.slot_step[r] < tmark);
            const unsigned long long m = __ballot(k);
            if (tid == 0) m64[0] = m;
            __syncthreads();
            const unsigned long long ms = m64[0], below = (1ull << (16 * wv)) - 1ull;
            upd_stage_masked(S.live + (size_t)(r0 + 16 * wv) * nT, (ms >> (16 * wv)) & 0xFFFFull, (ms >> (16 * wv)) & 0xFFFFull, nullptr,
                             tile + (size_t)__popcll(ms & below) * TS, TS, sh, D, nT, lane);
            n = __popcll(ms);
        } else {
            // ---- def: points alive at the mark that died later in the launch (see k_upd_move); rows scattered in the dead
            //      array: the four waves take them in turn, lane = coordinate
            const PcCtl *ctl = S.ctl;
            const int T = ctl->upd_T, ts = ctl->upd_ts;
            const int t = tmark + (c - npc - nlc) * UPDW_ROWS + tid;
            long long di = -1;
            if (tid < UPDW_ROWS && t < ts) {
                const PcPlan *pw = S.plan + (T - 1 - t);
                const int src = pw->dead_src;
                const bool existed = src >= 0 || (T - 1 - (-src - 1)) < tmark;
                if (pw->dead_idx >= 0 && pw->logw > S.logzero && existed) di = pw->dead_idx;
            }
            long long *dix = (long long *)(tile + (size_t)UPDW_ROWS * TS);       // (rows 64 .. 79 of the tile: zeroed again below)
            if (tid < UPDW_ROWS) dix[tid] = di;
            const unsigned long long m = __ballot(di >= 0);
            if (tid == 0) m64[0] = m;
            __syncthreads();
            unsigned long long mb = m64[0];
            int row = 0;
            while (mb) {
                const int b = __ffsll((long long)mb) - 1; mb &= mb - 1;
                if ((row & 3) == wv) {
                    const long long d = dix[b];
                    for (int e = lane; e <= D; e += 64) tile[(size_t)row * TS + e] = e < D ? S.dead[(size_t)d * nT + e] - sh[e] : 1.0;
                }
                row++;
            }
            n = row;
            __syncthreads();
            for (int e = tid; e < 16 * TS; e += 256) tile[(size_t)UPDW_ROWS * TS + e] = 0.0;
        }
        __syncthreads();
        // rows n .. next multiple of 16: zero up to the ones column (the matrix cores take four rows at a time)
        const int n16 = (n + 15) & ~15;
        for (int e = tid; e < (n16 - n) * (D + 1); e += 256) tile[(size_t)(n + e / (D + 1)) * TS + e % (D + 1)] = 0.0;
        __syncthreads();
        if (wv == 0) updw_accumulate<NT, 0>(tile, TS, n16, li, lk, acc);
        else if (wv == 1) updw_accumulate<NT, 1>(tile, TS, n16, li, lk, acc);
        else if (wv == 2) updw_accumulate<NT, 2>(tile, TS, n16, li, lk, acc);
        else updw_accumulate<NT, 3>(tile, TS, n16, li, lk, acc);
        __syncthreads();                                                // the tile is free for the next chunk
    }
    double *out = part + (size_t)blockIdx.x * E;
    if (wv == 0) updw_store<NT, 0>(out, D, li, lk, acc);
    else if (wv == 1) updw_store<NT, 1>(out, D, li, lk, acc);
    else if (wv == 2) updw_store<NT, 2>(out, D, li, lk, acc);
    else updw_store<NT, 3>(out, D, li, lk, acc);
}

// records added up; delta = mean - shift; n cov = M2 - n delta delta^T for k_cov_final_chol (which divides by n, stores the
// covariance and factorises in the reference's order of operations); new shift; thresholds reset
__global__ __launch_bounds__(1024) void k_upd_final_w(PcState S, int nb, const double *part, int E, double *shift, int def, double *ncov, int *count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tot = (double *)smem;                  // [nE]
    __shared__ double mu[128];
    const int tid = threadIdx.x, D = S.D, npair = D * (D + 1) / 2, nE = npair + D + 1;
    for (int e = tid; e < nE; e += 1024) {
        double s = 0.0;
        for (int k = 0; k < nb; k += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (k + u < nb) ? part[(size_t)(k + u) * E + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        tot[e] = s;
    }
    __syncthreads();
    const double n = tot[npair + D];
    if (tid < D) mu[tid] = tot[npair + tid] / n;
    __syncthreads();
    for (int p = tid; p < D * D; p += 1024) {
        const int a = p / D, b = p % D, lo = a < b ? a : b, hi = a < b ? b : a;
        ncov[p] = tot[lo * D - lo * (lo - 1) / 2 + (hi - lo)] - n * mu[lo] * mu[hi];      // population normalisation, run_time_info.f90:634
    }
    if (tid < D) shift[tid] += mu[tid];
    if (tid == 0) { count[0] = (int)n; if (!(def && S.ctl->upd_keep_thr)) S.death_thr[0] = -PC_HUGE; if (def) S.ctl->upd_pending = 0; }
}

extern "C" int pc_update_fused_entries(const PcState *S);
extern "C" void pc_launch_scan_blocks(int *blk, int nblk, int *total, int *total2, hipStream_t st);
extern "C" void pc_launch_chol_only(const PcState *S, const double *ncov, const int *count, hipStream_t st);
static int updw_grid(const PcState *S, int nph, int deferred)
{
    const int nblk = (nph + UPD_ROWS - 1) / UPD_ROWS, nlc = (S->Ncap + UPDW_ROWS - 1) / UPDW_ROWS, ndc = deferred ? (S->B + UPDW_ROWS - 1) / UPDW_ROWS : 0;
    const int total = 4 * nblk + nlc + ndc;
    return total < 512 ? total : 512;
}

extern "C" int pc_update_fused_ok(const PcState *S, int nc) { return nc == 1 && S->D <= 128; }

extern "C" int pc_update_fused_blocks(const PcState *S, int nph)
{   // partial records: one per block of k_upd_move (phantom blocks, live blocks, dead-row blocks of a deferred update),
    // then one per group of UPD_FOLD of them
    if (S->D >= 32) {                                   // persistent workgroups + their groups + room for n cov and n
        const int G = updw_grid(S, nph, 1), E = pc_update_fused_entries(S);
        return G + (G + UPD_FOLD - 1) / UPD_FOLD + (S->D * S->D + 2 + E - 1) / E;
    }
    const int nb = (nph + UPD_ROWS - 1) / UPD_ROWS + (S->Ncap + UPD_ROWS - 1) / UPD_ROWS + (S->B + UPD_ROWS - 1) / UPD_ROWS;
    return nb + (nb + UPD_FOLD - 1) / UPD_FOLD;
}
extern "C" int pc_update_fused_entries(const PcState *S) { const int e = S->D * (S->D + 1) / 2 + S->D + 1; return (e + 31) & ~31; }

// nph >= 1.  keep [nph], blk [blocks], part [pc_update_fused_blocks * pc_update_fused_entries] doubles, shift [D].
// deferred: see k_upd_flag
extern "C" void pc_launch_update_fused(const PcState *S, int nph, unsigned char *keep, int *blk, int *d_total, double *ph2, double *phL2,
                                       unsigned *phC2, unsigned long long *phU2, double *part, double *shift, int deferred, hipStream_t st)
{
    const int nblk = (nph + UPD_ROWS - 1) / UPD_ROWS, nlb = (S->Ncap + UPD_ROWS - 1) / UPD_ROWS, E = pc_update_fused_entries(S);
    const int ndb = deferred ? (S->B + UPD_ROWS - 1) / UPD_ROWS : 0;
    if (S->D >= 32) {
        const int D = S->D, NTv = (D + 1 + 15) / 16, TSv = 16 * NTv + ((NTv & 1) ? 0 : 16);
        const int nlc = (S->Ncap + UPDW_ROWS - 1) / UPDW_ROWS, ndc = deferred ? (S->B + UPDW_ROWS - 1) / UPDW_ROWS : 0;
        const int G = updw_grid(S, nph, deferred), ng = (G + UPD_FOLD - 1) / UPD_FOLD;
        double *part2 = part + (size_t)G * E, *ncov = part2 + (size_t)ng * E;
        int *count = (int *)(ncov + (size_t)D * D);
        const size_t shw = sizeof(double) * ((size_t)(UPDW_ROWS + 16) * TSv + D);
        hipLaunchKernelGGL(k_upd_flag, dim3(nblk), dim3(UPD_NT), 0, st, *S, nph, keep, blk, deferred);
        pc_launch_scan_blocks(blk, nblk, d_total, &S->ctl->nphantom, st);
#define UPDW_LAUNCH(NT) { \
            static bool done_##NT = false; \
            if (!done_##NT) { (void)hipFuncSetAttribute((const void *)k_upd_move_w<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shw); done_##NT = true; } \
            hipLaunchKernelGGL((k_upd_move_w<NT>), dim3(G), dim3(256), shw, st, *S, nph, nblk, (const unsigned char *)keep, (const int *)blk, \
                               ph2, phL2, phC2, phU2, (const double *)shift, part, E, deferred, nlc, ndc); }
        switch (NTv) { case 3: UPDW_LAUNCH(3) break; case 4: UPDW_LAUNCH(4) break; case 5: UPDW_LAUNCH(5) break; case 6: UPDW_LAUNCH(6) break;
                       case 7: UPDW_LAUNCH(7) break; case 8: UPDW_LAUNCH(8) break; default: UPDW_LAUNCH(9) break; }
#undef UPDW_LAUNCH
        hipLaunchKernelGGL(k_upd_fold, dim3(ng), dim3(256), 0, st, (const double *)part, G, E, part2);
        const size_t shf = sizeof(double) * (size_t)(D * (D + 1) / 2 + D + 1);
        static size_t donef = 0;
        if (shf > donef) { (void)hipFuncSetAttribute((const void *)k_upd_final_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shf); donef = shf; }
        hipLaunchKernelGGL(k_upd_final_w, dim3(1), dim3(1024), shf, st, *S, ng, (const double *)part2, E, shift, deferred, ncov, count);
        pc_launch_chol_only(S, ncov, count, st);
        return;
    }
    const int TS = ((S->D + 2) | 1);
    size_t sh = sizeof(double) * ((size_t)(UPD_ROWS + 16) * TS + S->D);
    if (sh < sizeof(double) * (12 * 256 + S->D)) sh = sizeof(double) * (12 * 256 + S->D);       // the waves' result tiles reuse the row tile
    static size_t done = 0;
    if (sh > done) { (void)hipFuncSetAttribute((const void *)k_upd_move, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); done = sh; }
    hipLaunchKernelGGL(k_upd_flag, dim3(nblk), dim3(UPD_NT), 0, st, *S, nph, keep, blk, deferred);
    hipLaunchKernelGGL(k_upd_move, dim3(nblk + nlb + ndb), dim3(UPD_NT), sh, st, *S, nph, nblk, (const unsigned char *)keep, (const int *)blk,
                       ph2, phL2, phC2, phU2, d_total, (const double *)shift, part, E, deferred, nlb);
    const int nb = nblk + nlb + ndb, ng = (nb + UPD_FOLD - 1) / UPD_FOLD;
    double *part2 = part + (size_t)nb * E;
    hipLaunchKernelGGL(k_upd_fold, dim3(ng), dim3(256), 0, st, (const double *)part, nb, E, part2);
    hipLaunchKernelGGL(k_upd_final, dim3(1), dim3(1024), 0, st, *S, ng, (const double *)part2, E, shift, deferred);
}
