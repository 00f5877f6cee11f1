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
//   k_upd_gather persistent workgroups: compaction of the survivors into the alternate buffer (offsets from k_scan_blocks;
//                pool mode: none) and -- in the same pass -- first and second moments of the rows' cube coordinates about a
//                SHIFT; the live points likewise
//   k_upd_fold   the blocks' moments added in groups of sixteen
//   k_upd_final  the groups' moments added, mean / covariance / Cholesky factor, new shift, thresholds reset
// One pass instead of two (mean, then centred products) needs the shift: the moments are taken about the PREVIOUS update's
// mean (the cube centre at first), which the new mean is within a fraction of a standard deviation of, so the
// subtraction cov = M2/n - delta delta^T loses a digit at most, not the six it would lose about the origin.
#include "pc_state.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

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
// (a wavefront takes one block of the counts -- 256 rows, four consecutive rows per lane: 32-byte, 16-byte and 4-byte accesses in
//  place of 8, 4 and 1 -- and a workgroup four of them; the survivors of a block are counted by ballots, no LDS)
__device__ __forceinline__ void upd_flag_body(const PcState &S, int nph, unsigned char *keep, int *blk_count, int def, int nblk)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sb = blockIdx.x * 4 + wv;
    if (sb >= nblk) return;
    const unsigned uid0 = S.cl_uid[0];
    const double thr = def ? S.ctl->upd_thr : S.death_thr[0];
    const bool pool = S.pool;
    // pool mode: a dropped phantom is dropped where it lies; the rows that count in the moments (the survivors that were
    // phantoms at the mark -- rows of chains consumed later stay, but do not count) are listed by k_upd_index
    const int tmark = (pool && def) ? S.ctl->upd_tmark : 0x7fffffff, nph0u = (pool && def) ? S.ctl->upd_nph0 : 0x7fffffff;
    const int T = (pool && def) ? S.ctl->upd_T : 0, nr = S.nr;
    const int j0 = sb * UPD_ROWS + 4 * lane;
    const bool full = j0 + 3 < nph;
    unsigned cu[4] = {PC_CUID_NONE, PC_CUID_NONE, PC_CUID_NONE, PC_CUID_NONE};
    double ll[4] = {0.0, 0.0, 0.0, 0.0};
    if (full) {
        const uint4 c = *(const uint4 *)(S.ph_cuid + j0);
        const double2 a = *(const double2 *)(S.ph_logL + j0), b2 = *(const double2 *)(S.ph_logL + j0 + 2);
        cu[0] = c.x; cu[1] = c.y; cu[2] = c.z; cu[3] = c.w; ll[0] = a.x; ll[1] = a.y; ll[2] = b2.x; ll[3] = b2.y;
    } else {
        for (int u = 0; u < 4; ++u) if (j0 + u < nph) { cu[u] = S.ph_cuid[j0 + u]; ll[u] = S.ph_logL[j0 + u]; }
    }
    unsigned kb = 0; bool wr = false; int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        bool k = false, km = false;
        if (j < nph) {
            k = (cu[u] == uid0) && !(ll[u] < thr);
            if (pool) {
                if (!k && cu[u] == uid0) { cu[u] = PC_CUID_NONE; wr = true; }
                km = k && (j < nph0u || T - 1 - (j - nph0u) / nr < tmark);
            }
            kb |= (unsigned)((k ? 1 : 0) | (km ? 2 : 0)) << (8 * u);
        }
        cnt += __popcll(__ballot(pool ? km : k));
    }
    if (full) {
        *(unsigned *)(keep + j0) = kb;
        if (wr) *(uint4 *)(S.ph_cuid + j0) = make_uint4(cu[0], cu[1], cu[2], cu[3]);
    } else {
        for (int u = 0; u < 4; ++u) if (j0 + u < nph) { keep[j0 + u] = (unsigned char)((kb >> (8 * u)) & 0xFFu); if (wr) S.ph_cuid[j0 + u] = cu[u]; }
    }
    if (lane == 0) blk_count[sb] = cnt;
}
__global__ __launch_bounds__(UPD_NT) void k_upd_flag(PcState S, int nph, unsigned char *keep, int *blk_count, int def, int nblk) { upd_flag_body(S, nph, keep, blk_count, def, nblk); }
__global__ __launch_bounds__(UPD_NT) void k_upd_flag_many(const PcManyRec *R, int def) { const PcManyView r = pc_many_view(R, blockIdx.y); if ((int)blockIdx.x * 4 >= r.ia[2]) return; upd_flag_body(r.S, r.ia[1], (unsigned char *)r.p[0], (int *)r.p[1], def, r.ia[2]); }


// exclusive scan of the block counts in place, one workgroup, 4096 counts at a time (four per thread, coalesced)
__global__ __launch_bounds__(1024) void k_upd_scan(int *cnt, int n, int *total)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + 4 * tid;
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (i0 + u < n) ? cnt[i0 + u] : 0;
        const int s = (c[0] + c[1]) + (c[2] + c[3]);
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int pre = carry;
        for (int x = 0; x < wv; ++x) pre += wsum[x];
        int run = pre + inc - s;
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (i0 + u < n) cnt[i0 + u] = run; run += c[u]; }
        if (tid == 1023) carry_s = pre + inc;
        __syncthreads();
        carry = carry_s;
    }
    if (tid == 0) *total = carry;
}

// pool mode: the rows that count, by index, in row order
__global__ __launch_bounds__(UPD_NT) void k_upd_index(int nph, const unsigned char *keep, const int *blk_off, int *idx)
{
    __shared__ int cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = blockIdx.x * UPD_ROWS + tid;
    const bool km = j < nph && (keep[j] & 2);
    const unsigned long long m = __ballot(km);
    if (lane == 0) cnt[wv] = __popcll(m);
    __syncthreads();
    int off = blk_off[blockIdx.x];
    for (int x = 0; x < wv; ++x) off += cnt[x];
    if (km) idx[off + __popcll(m & ((1ull << lane) - 1ull))] = j;
}

// the same without the scan launch in front, for index lists of up to UPD_SELF_BLOCKS blocks: a block's offset is the sum of
// the counts of the blocks before it, which 256 threads add up from L2 in about a microsecond (at 3750 blocks: 7 M loads on the
// whole chip) -- less than the one-workgroup scan kernel and the launch boundary behind it (4.7 + 3.5 us per update)
#define UPD_SELF_BLOCKS 4096
// (a workgroup of four wavefronts takes sixteen blocks of 256 rows, a wavefront four of them one after the other, four rows a lane:
//  the counts before the workgroup's first block are added up once for sixteen blocks -- with 2500 blocks a run and sixteen runs
//  in step the sums were the kernel's time.  Sixteen wavefronts a workgroup did the same at sixteen runs and took 457 us at
//  sixty-four: a workgroup that needs sixteen free slots on one CU waits while the second stream's kernels hold some)
#define UPD_IDX_NT 256
#define UPD_IDX_BLOCKS 16
__device__ __forceinline__ void upd_index_self_body(int nph, const unsigned char *keep, const int *blk_count, int nblk, int *idx, int *total)
{
    __shared__ int part[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sb0 = blockIdx.x * UPD_IDX_BLOCKS;
    int s = 0;
    for (int b = tid; b < sb0; b += UPD_IDX_NT) s += blk_count[b];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[wv] = s;
    __syncthreads();
    int off = (part[0] + part[1]) + (part[2] + part[3]);
    const int sbw = sb0 + 4 * wv;                      // this wavefront's four blocks
    for (int x = sb0; x < sbw && x < nblk; ++x) off += blk_count[x];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int q = 0; q < 4; ++q) {
        const int sb = sbw + q;
        if (sb >= nblk) return;
        const int j0 = sb * UPD_ROWS + 4 * lane;
        unsigned w = 0;
        if (j0 + 3 < nph) w = *(const unsigned *)(keep + j0);
        else for (int u = 0; u < 4; ++u) if (j0 + u < nph) w |= (unsigned)keep[j0 + u] << (8 * u);
        int pos = off, all = 0;
        bool f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f[u] = (w >> (8 * u + 1)) & 1u;
            const unsigned long long m = __ballot(f[u]);
            pos += __popcll(m & below); all += __popcll(m);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (f[u]) { idx[pos] = j0 + u; pos++; }
        off += all;
        if (sb == nblk - 1 && lane == 0) *total = off;
    }
}
__global__ __launch_bounds__(UPD_IDX_NT) void k_upd_index_self(int nph, const unsigned char *keep, const int *blk_count, int nblk, int *idx, int *total) { upd_index_self_body(nph, keep, blk_count, nblk, idx, total); }
__global__ __launch_bounds__(UPD_IDX_NT) void k_upd_index_self_many(const PcManyRec *R) { const PcManyView r = pc_many_view(R, blockIdx.y); if ((int)blockIdx.x * 16 >= r.ia[2]) return; upd_index_self_body(r.ia[1], (const unsigned char *)r.p[0], (const int *)r.p[1], r.ia[2], (int *)r.p[5], (int *)r.p[2]); }


// sixteen rows given by index, coordinates minus shift and a one, to consecutive tile rows; lane = element, the loads of
// two passes (elements lane and lane + 64) in flight together
__device__ __forceinline__ void upd_stage_idx(const double *base, const int *ridx, int cnt, double *tile, int TS, const double *sh, int D, int nT, int lane)
{
    if (cnt <= 0) return;
    int idx[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) idx[u] = u < cnt ? ridx[u] : 0;
    for (int e0 = lane; e0 <= D; e0 += 128) {
        const int e1 = e0 + 64;
        double v0[16], v1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < cnt) { v0[u] = e0 < D ? base[(size_t)idx[u] * nT + e0] : 1.0; v1[u] = e1 < D ? base[(size_t)idx[u] * nT + e1] : 1.0; }
        const double s0 = e0 < D ? sh[e0] : 0.0, s1 = e1 < D ? sh[e1] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < cnt) { tile[(size_t)u * TS + e0] = v0[u] - s0; if (e1 <= D) tile[(size_t)u * TS + e1] = v1[u] - s1; }
    }
}

typedef double upd_v4d __attribute__((ext_vector_type(4)));
// ------------------------------------------------------------------------------------------------------------------
// k_upd_gather: ONE pass over the phantom array (and the live points, and for a deferred update the points that died after
// the mark): the surviving phantoms are compacted into the alternate buffers -- or, in pool mode, left where they lie --
// and the rows that count go, coordinates minus shift and a column of ones, into an LDS tile; when the tile is full its
// X^T X is added to accumulator tiles in registers (fp64 matrix cores, v_mfma_f64_16x16x4_f64).  Workgroups are
// persistent: each walks its share of the 256-row blocks and writes ONE partial record
//   [npair second moments (a <= b, row-major upper triangle) | D first moments | count], padded to E,
// so sparse stretches of the array (pool mode: most rows are invalid between compactions) cost a ballot, not a launch
// of the whole machinery.  The matrix cores are not there for the flops: with one product per thread every FMA costs two
// LDS reads and the LDS pipe bounds the kernel; a 16x16x4 tile needs two reads per 2048 flops.
//   NT <= 2 (nDims < 32): every wave holds all tile pairs and takes the rows 4w .. 4w+3 of each group of sixteen;
//   NT >= 3: the tile pairs are dealt out to the four waves (q = wave, wave + 4, ...), every wave takes all rows.
template <int NT, int W>
__device__ __forceinline__ void updg_accumulate(const double *tile, int TS, int n16, int wv, int li, int lk, upd_v4d (&acc)[(NT <= 2) ? NT * (NT + 1) / 2 : (NT * (NT + 1) / 2 + 3) / 4])
{
    if constexpr (NT <= 2) {
        for (int q0 = 4 * wv; q0 < n16; q0 += 16) {
            const double *row = tile + (size_t)(q0 + lk) * TS + li;
            double x[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) x[t] = row[16 * t];
            int q = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ti], x[tj], acc[q], 0, 0, 0);
        }
    } else {
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
}
// element (a, b) of tile pair (ti, tj) held in (lane, register r) -> its place in the record (or -1)
__device__ __forceinline__ int updg_slot(int D, int ti, int tj, int li, int lk, int r)
{
    const int a = 16 * ti + lk + 4 * r, b = 16 * tj + li, npair = D * (D + 1) / 2;
    if (a > b || b > D) return -1;
    return (b < D) ? a * D - a * (a - 1) / 2 + (b - a) : (a < D ? npair + a : npair + D);
}

template <int NT, bool IDX>
__device__ __forceinline__ void upd_gather_body(const PcState &S, int nph, int nblk, const unsigned char *keep, const int *blk_off,
                                                double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2,
                                                const int *idx, const int *nidx_p,
                                                const double *shift, double *part, int E, int def, int nlb, int ndb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool KS = NT <= 2;
    // row stride: odd (K split: the four row groups of an operand start in different banks) / 16 mod 32 doubles (pair split)
    constexpr int TS = KS ? 16 * NT + 1 : 16 * NT + ((NT & 1) ? 0 : 16);
    constexpr int CAP = KS ? 128 : 64;                                  // tile rows; a flush when the next 64-row piece does not fit
    constexpr int NACC = KS ? NT * (NT + 1) / 2 : (NT * (NT + 1) / 2 + 3) / 4;
    __shared__ unsigned long long m64[4], mm64[4];
    __shared__ long long dix[256];
    const int tmark = def ? S.ctl->upd_tmark : 0x7fffffff, nph0u = def ? S.ctl->upd_nph0 : 0x7fffffff, updT = def ? S.ctl->upd_T : 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4, D = S.D, nT = S.nT;
    constexpr bool pool = IDX;                                          // (pool mode always comes with the index list)
    double *tile = (double *)smem;                                      // [CAP][TS]
    double *sh = tile + (size_t)CAP * TS;                               // [D]
    for (int d = tid; d < D; d += 256) sh[d] = shift[d];
    for (int e = tid; e < CAP * TS; e += 256) tile[e] = 0.0;            // (the columns past the ones stay zero)
    upd_v4d acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = upd_v4d{0.0, 0.0, 0.0, 0.0};
    int nfill = 0;
    auto flush = [&]() __attribute__((always_inline)) {
        __syncthreads();
        const int n16 = (nfill + 15) & ~15;
        for (int e = tid; e < (n16 - nfill) * (D + 1); e += 256) tile[(size_t)(nfill + e / (D + 1)) * TS + e % (D + 1)] = 0.0;
        __syncthreads();
        if (KS || wv == 0) updg_accumulate<NT, 0>(tile, TS, n16, wv, li, lk, acc);
        else if (wv == 1) updg_accumulate<NT, 1>(tile, TS, n16, wv, li, lk, acc);
        else if (wv == 2) updg_accumulate<NT, 2>(tile, TS, n16, wv, li, lk, acc);
        else updg_accumulate<NT, 3>(tile, TS, n16, wv, li, lk, acc);
        __syncthreads();
        nfill = 0;
    };
    // the 64-row piece `sub` of a 256-row block whose masks sit in m64 / mm64: rows selected by the move mask are copied to
    // consecutive rows at dst (when there is one), those of the moment mask join the tile; wave w takes the bits 16 w ..
    auto piece = [&](const double *src, int sub, double *dst) __attribute__((always_inline)) {
        const unsigned long long ms = m64[sub], mms = mm64[sub], below = (1ull << (16 * wv)) - 1ull;
        const int cnt = __popcll(mms);
        if (cnt == 0 && (dst == nullptr || ms == 0ull)) return;
        if (nfill + cnt > CAP) flush();
        upd_stage_masked(src + (size_t)(sub * 64 + 16 * wv) * nT, ((dst ? ms : mms) >> (16 * wv)) & 0xFFFFull, (mms >> (16 * wv)) & 0xFFFFull,
                         dst ? dst + (size_t)__popcll(ms & below) * nT : nullptr, tile + (size_t)(nfill + __popcll(mms & below)) * TS, TS, sh, D, nT, lane);
        nfill += cnt;
    };
    if constexpr (IDX) {
        // pool mode: the phantoms that count were listed by k_upd_index (idx, *nidx_p of them): sixty-four at a time, sixteen per
        // wave, no masks, no barriers but the flushes
        const int nidx = *nidx_p;
        const int gchunk = (int)gridDim.x - nlb - ndb;                      // the last nlb + ndb workgroups take a live / dead block each
        for (int c = blockIdx.x; (int)blockIdx.x < gchunk && c * 64 < nidx; c += gchunk) {
            const int have = min(64, nidx - c * 64);
            if (nfill + have > CAP) flush();
            upd_stage_idx(S.phantom, idx + c * 64 + 16 * wv, min(16, have - 16 * wv), tile + (size_t)(nfill + 16 * wv) * TS, TS, sh, D, nT, lane);
            nfill += have;
        }
    }
    const int total = nblk + nlb + ndb;
    for (int b = IDX ? ((int)blockIdx.x >= (int)gridDim.x - nlb - ndb ? nblk + (int)blockIdx.x - ((int)gridDim.x - nlb - ndb) : total) : blockIdx.x; b < total; b += gridDim.x) {
        __syncthreads();                                                // the masks of the block before are no longer read
        if (b < nblk) {
            // ---- 256 phantom rows
            const int j = b * 256 + tid;
            const bool k = (j < nph) && keep[j];
            // moments: the survivors that were phantoms at the mark (rows of chains consumed later stay, but do not count)
            const bool km = k && (j < nph0u || (pool ? updT - 1 - (j - nph0u) / S.nr : (j - nph0u) / S.nr) < tmark);
            const unsigned long long m = __ballot(k), mm = __ballot(km);
            if (lane == 0) { m64[wv] = m; mm64[wv] = mm; }
            __syncthreads();
            int off = 0;
            if (!pool) {
                off = blk_off[b];                                       // survivors of the blocks before (k_scan_blocks)
                for (int x = 0; x < wv; ++x) off += __popcll(m64[x]);
                if (k) {
                    const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                    phL2[pos] = S.ph_logL[j]; phC2[pos] = S.ph_cuid[j]; phU2[pos] = S.ph_uid[j];
                }
                off = blk_off[b];
            }
            const double *src = S.phantom + (size_t)b * 256 * nT;
            for (int sub = 0; sub < 4; ++sub) {
                piece(src, sub, pool ? nullptr : ph2 + (size_t)off * nT);
                off += __popcll(m64[sub]);
            }
        } else if (b < nblk + nlb) {
            // ---- live points of slots [r0, r0 + 256)
            const int r0 = (b - nblk) * 256, r = r0 + tid;
            const bool k = (r < S.Ncap) && S.live_cluster[r] == 0 && (!def || S.slot_step[r] < tmark);
            const unsigned long long m = __ballot(k);
            if (lane == 0) { m64[wv] = m; mm64[wv] = m; }
            __syncthreads();
            const double *src = S.live + (size_t)r0 * nT;
            for (int sub = 0; sub < 4; ++sub) piece(src, sub, nullptr);
        } else {
            // ---- def: points alive at the mark that died later in the launch: the dead rows of the steps t >= tmark whose
            //      dying point was a snapshot point or the newcomer of a step before the mark; scattered in the dead array:
            //      the four waves take them in turn, lane = coordinate
            const PcCtl *ctl = S.ctl;
            const int T = ctl->upd_T, ts = ctl->upd_ts;
            const int t = tmark + (b - nblk - nlb) * 256 + tid;
            long long di = -1;
            if (t < ts) {
                const PcPlan *pw = S.plan + (T - 1 - t);
                const int src = pw->dead_src;
                const bool existed = src >= 0 || (T - 1 - (-src - 1)) < tmark;
                if (pw->dead_idx >= 0 && pw->logw > S.logzero && existed) di = pw->dead_idx;
            }
            dix[tid] = di;
            const unsigned long long m = __ballot(di >= 0);
            if (lane == 0) m64[wv] = m;
            __syncthreads();
            for (int sub = 0; sub < 4; ++sub) {
                unsigned long long mb = m64[sub];
                const int cnt = __popcll(mb);
                if (cnt == 0) continue;
                if (nfill + cnt > CAP) flush();
                int row = 0;
                while (mb) {
                    const int bit = __ffsll((long long)mb) - 1; mb &= mb - 1;
                    if ((row & 3) == wv) {
                        const long long d = dix[sub * 64 + bit];
                        for (int e = lane; e <= D; e += 64) tile[(size_t)(nfill + row) * TS + e] = e < D ? S.dead[(size_t)d * nT + e] - sh[e] : 1.0;
                    }
                    row++;
                }
                nfill += cnt;
            }
        }
    }
    flush();
    double *out = part + (size_t)blockIdx.x * E;
    if constexpr (KS) {
        // the four waves' tiles added in order (the tile memory takes them)
        double *wres = tile;                           // [4 waves][NACC][256]
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) wres[((size_t)wv * NACC + t) * 256 + (lk + 4 * r) * 16 + li] = acc[t][r];
        __syncthreads();
        if (wv == 0) {
            int q = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj, ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = updg_slot(D, ti, tj, li, lk, r), idx = (lk + 4 * r) * 16 + li;
                        if (p >= 0) out[p] = ((wres[(size_t)(0 * NACC + q) * 256 + idx] + wres[(size_t)(1 * NACC + q) * 256 + idx]) +
                                              (wres[(size_t)(2 * NACC + q) * 256 + idx] + wres[(size_t)(3 * NACC + q) * 256 + idx]));
                    }
        }
    } else {
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = ti; tj < NT; ++tj, ++q)
                if ((q & 3) == wv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int p = updg_slot(D, ti, tj, li, lk, r); if (p >= 0) out[p] = acc[q >> 2][r]; }
                }
    }
}
template <int NT, bool IDX>
__global__ __launch_bounds__(256) void k_upd_gather(PcState S, int nph, int nblk, const unsigned char *keep, const int *blk_off,
                                                    double *ph2, double *phL2, unsigned *phC2, unsigned long long *phU2,
                                                    const int *idx, const int *nidx_p,
                                                    const double *shift, double *part, int E, int def, int nlb, int ndb)
{ upd_gather_body<NT, IDX>(S, nph, nblk, keep, blk_off, ph2, phL2, phC2, phU2, idx, nidx_p, shift, part, E, def, nlb, ndb); }
template <int NT>
__global__ __launch_bounds__(256) void k_upd_gather_many(const PcManyRec *R, int E, int def, int nlb, int ndb)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    upd_gather_body<NT, true>(r.S, r.ia[1], r.ia[2], (const unsigned char *)r.p[0], (const int *)r.p[1], (double *)r.p[3], (double *)r.p[4], (unsigned *)r.p[5],
                              (unsigned long long *)r.p[6], (const int *)r.p[5], (const int *)r.p[2], (const double *)r.p[8], (double *)r.p[7], E, def, nlb, ndb);
}


// partial records folded in groups of sixteen (one batch of loads per thread), in block order
#define UPD_FOLD 16
__device__ __forceinline__ void upd_fold_body(const double *part, int nb, int E, double *part2)
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
__global__ __launch_bounds__(256) void k_upd_fold(const double *part, int nb, int E, double *part2) { upd_fold_body(part, nb, E, part2); }
__global__ __launch_bounds__(256) void k_upd_fold_many(const PcManyRec *R, int nb, int E) { double *part = pc_as_global((double *)R[blockIdx.y].p[7], R); upd_fold_body(part, nb, E, part + (size_t)nb * E); }


// fold + mean + covariance + Cholesky; one workgroup of 1024 threads
__device__ __forceinline__ void upd_final_body(const PcState &S, int nb, const double *part, int E, double *shift, int def)
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
__global__ __launch_bounds__(1024) void k_upd_final(PcState S, int nb, const double *part, int E, double *shift, int def) { upd_final_body(S, nb, part, E, shift, def); }
__global__ __launch_bounds__(1024) void k_upd_final_many(const PcManyRec *R, int nb, int E, int def, int G) { const PcManyView r = pc_many_view(R, blockIdx.y); upd_final_body(r.S, nb, (const double *)r.p[7] + (size_t)G * E, E, (double *)r.p[8], def); }


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
static int upd_grid(const PcState *S, int nph, int deferred)
{
    const int nblk = (nph + UPD_ROWS - 1) / UPD_ROWS, nlb = (S->Ncap + UPD_ROWS - 1) / UPD_ROWS, ndb = deferred ? (S->B + UPD_ROWS - 1) / UPD_ROWS : 0;
    static const int cap_env = std::getenv("PC_UPD_GRID") ? std::atoi(std::getenv("PC_UPD_GRID")) : 0;
    const int cap = cap_env > 0 ? cap_env : (S->D < 32 ? 384 : 512);               // persistent workgroups (measured: 1024 / 384 / 256 / 128 -> 16.6 / 16.3 / 16.4 / 17.2 ms per run at the metric configuration: fewer records to fold against fewer rows in flight)
    if (S->pool) { const int nch = (nph + 63) / 64; return (nch < cap ? (nch > 0 ? nch : 1) : cap) + nlb + ndb; }     // index chunks + a workgroup per live / dead block
    const int total = nblk + nlb + ndb;
    return total < cap ? total : cap;
}

extern "C" int pc_update_fused_ok(const PcState *S, int nc) { return nc == 1 && S->D <= 128; }
extern "C" int pc_update_fused_blocks(const PcState *S, int nph)
{   // partial records: one per persistent workgroup of k_upd_gather, then one per group of UPD_FOLD of them, then room for
    // n cov and n (nDims >= 32: the Cholesky factor is made by k_cov_final_chol)
    const int G = upd_grid(S, nph, 1), E = pc_update_fused_entries(S);
    return G + (G + UPD_FOLD - 1) / UPD_FOLD + (S->D * S->D + 2 + E - 1) / E;
}
extern "C" int pc_update_fused_entries(const PcState *S) { const int e = S->D * (S->D + 1) / 2 + S->D + 1; return (e + 31) & ~31; }

// nph >= 1.  keep [nph], blk [blocks], part [pc_update_fused_blocks * pc_update_fused_entries] doubles, shift [D].
// deferred: see k_upd_flag
extern "C" void pc_launch_update_fused(const PcState *S, int nph, unsigned char *keep, int *blk, int *d_total, double *ph2, double *phL2,
                                       unsigned *phC2, unsigned long long *phU2, double *part, double *shift, int deferred, hipStream_t st)
{
    const int D = S->D, nblk = (nph + UPD_ROWS - 1) / UPD_ROWS, nlb = (S->Ncap + UPD_ROWS - 1) / UPD_ROWS, E = pc_update_fused_entries(S);
    const int ndb = deferred ? (S->B + UPD_ROWS - 1) / UPD_ROWS : 0;
    const int NTv = (D + 1 + 15) / 16, TSv = NTv <= 2 ? 16 * NTv + 1 : 16 * NTv + ((NTv & 1) ? 0 : 16), CAPv = NTv <= 2 ? 128 : 64;
    const int G = upd_grid(S, nph, deferred), ng = (G + UPD_FOLD - 1) / UPD_FOLD;
    double *part2 = part + (size_t)G * E;
    size_t shw = sizeof(double) * ((size_t)CAPv * TSv + D);
    if (NTv <= 2 && shw < sizeof(double) * (size_t)(4 * 3 * 256)) shw = sizeof(double) * (size_t)(4 * 3 * 256);      // the waves' result tiles reuse the row tile
    hipLaunchKernelGGL(k_upd_flag, dim3((nblk + 3) / 4), dim3(UPD_NT), 0, st, *S, nph, keep, blk, deferred, nblk);
    if (!S->pool) pc_launch_scan_blocks(blk, nblk, d_total, &S->ctl->nphantom, st);
    else if (nblk <= UPD_SELF_BLOCKS && !std::getenv("PC_UPD_SCAN_LAUNCH"))
        hipLaunchKernelGGL(k_upd_index_self, dim3((nblk + 15) / 16), dim3(UPD_IDX_NT), 0, st, nph, (const unsigned char *)keep, (const int *)blk, nblk, (int *)phC2, d_total);
    else {
        hipLaunchKernelGGL(k_upd_scan, dim3(1), dim3(1024), 0, st, blk, nblk, d_total);
        hipLaunchKernelGGL(k_upd_index, dim3(nblk), dim3(UPD_NT), 0, st, nph, (const unsigned char *)keep, (const int *)blk, (int *)phC2);
    }
    int devi = 0; (void)hipGetDevice(&devi); devi &= 63;
#define UPDG_LAUNCH(NT) { \
        pc_need_dyn_lds((const void *)k_upd_gather<NT, false>, shw); \
                          pc_need_dyn_lds((const void *)k_upd_gather<NT, true>, shw); \
        /* pool mode: the alternate id buffer is free between compactions and holds the index list, *d_total its length */ \
        if (S->pool) hipLaunchKernelGGL((k_upd_gather<NT, true>), dim3(G), dim3(256), shw, st, *S, nph, nblk, (const unsigned char *)keep, (const int *)blk, \
                           ph2, phL2, phC2, phU2, (const int *)phC2, (const int *)d_total, (const double *)shift, part, E, deferred, nlb, ndb); \
        else hipLaunchKernelGGL((k_upd_gather<NT, false>), dim3(G), dim3(256), shw, st, *S, nph, nblk, (const unsigned char *)keep, (const int *)blk, \
                           ph2, phL2, phC2, phU2, (const int *)nullptr, (const int *)nullptr, (const double *)shift, part, E, deferred, nlb, ndb); }
    switch (NTv) { case 1: UPDG_LAUNCH(1) break; case 2: UPDG_LAUNCH(2) break; case 3: UPDG_LAUNCH(3) break; case 4: UPDG_LAUNCH(4) break;
                   case 5: UPDG_LAUNCH(5) break; case 6: UPDG_LAUNCH(6) break; case 7: UPDG_LAUNCH(7) break; case 8: UPDG_LAUNCH(8) break;
                   default: UPDG_LAUNCH(9) break; }
#undef UPDG_LAUNCH
    hipLaunchKernelGGL(k_upd_fold, dim3(ng), dim3(256), 0, st, (const double *)part, G, E, part2);
    if (D < 32) { hipLaunchKernelGGL(k_upd_final, dim3(1), dim3(1024), 0, st, *S, ng, (const double *)part2, E, shift, deferred); return; }
    double *ncov = part2 + (size_t)ng * E;
    int *count = (int *)(ncov + (size_t)D * D);
    const size_t shf = sizeof(double) * (size_t)(D * (D + 1) / 2 + D + 1);
    pc_need_dyn_lds((const void *)k_upd_final_w, shf);
    hipLaunchKernelGGL(k_upd_final_w, dim3(1), dim3(1024), shf, st, *S, ng, (const double *)part2, E, shift, deferred, ncov, count);
    pc_launch_chol_only(S, ncov, count, st);
}

extern "C" int pc_update_fused_grid(const PcState *S, int nph, int deferred) { return upd_grid(S, nph, deferred); }
// the fused update for R runs of one shape at once (blockIdx.y = run); every run brings its own number of phantom rows
// (PcManyRec::ia[1], its blocks in ia[2]); all of them the same number G of gathering workgroups (pc_update_fused_grid) and
// nblk_max >= their blocks.  1: not this way (the caller launches them one by one)
extern "C" int pc_launch_update_fused_many(const PcState *S, const PcManyRec *dR, int R, int nblk_max, int G, int deferred, hipStream_t st)
{
    const int D = S->D, nlb = (S->Ncap + UPD_ROWS - 1) / UPD_ROWS, E = pc_update_fused_entries(S);
    if (!S->pool || nblk_max > UPD_SELF_BLOCKS || D >= 32 || nblk_max < 1) return 1;
    const int ndb = deferred ? (S->B + UPD_ROWS - 1) / UPD_ROWS : 0;
    const int NTv = (D + 1 + 15) / 16, TSv = NTv <= 2 ? 16 * NTv + 1 : 16 * NTv + ((NTv & 1) ? 0 : 16), CAPv = NTv <= 2 ? 128 : 64;
    const int ng = (G + UPD_FOLD - 1) / UPD_FOLD;
    size_t shw = sizeof(double) * ((size_t)CAPv * TSv + D);
    if (NTv <= 2 && shw < sizeof(double) * (size_t)(4 * 3 * 256)) shw = sizeof(double) * (size_t)(4 * 3 * 256);
    hipLaunchKernelGGL(k_upd_flag_many, dim3((nblk_max + 3) / 4, R), dim3(UPD_NT), 0, st, dR, deferred);
    hipLaunchKernelGGL(k_upd_index_self_many, dim3((nblk_max + 15) / 16, R), dim3(UPD_IDX_NT), 0, st, dR);
    int devi = 0; (void)hipGetDevice(&devi); devi &= 63;
#define UPDM_LAUNCH(NT) { \
        pc_need_dyn_lds((const void *)k_upd_gather_many<NT>, shw); \
        hipLaunchKernelGGL((k_upd_gather_many<NT>), dim3(G, R), dim3(256), shw, st, dR, E, deferred, nlb, ndb); }
    if (NTv == 1) UPDM_LAUNCH(1) else UPDM_LAUNCH(2)
#undef UPDM_LAUNCH
    hipLaunchKernelGGL(k_upd_fold_many, dim3(ng, R), dim3(256), 0, st, dR, G, E);
    hipLaunchKernelGGL(k_upd_final_many, dim3(1, R), dim3(1024), 0, st, dR, ng, E, deferred, G);
    return 0;
}
