// pc_slice_t.hip -- the slice-sampling chains with the layout turned round: lane = chain.
//
// k_slice (pc_sample.hip) gives a chain a whole wavefront, lane = coordinate: the shortest path for ONE run, whose nursery of
// B chains then occupies B wavefronts for the length of a launch although 20 of 64 lanes work and the wave waits on its own
// dependent operations most of the time.  When several runs share a device (pchip_run_repeats: repeats of one problem) the
// chip's time is what counts, not a launch's latency.  Here a wavefront carries 64 chains, one per lane, every coordinate loop
// runs inside the lane, and a nursery of 1000 chains is 16 wavefronts: sixteen runs in flight leave each other the chip.
//
// The same numbers as k_slice's fused path (template FW > 0: seed choice and whitening inside the kernel), bit for bit:
// the same keyed draws (seed: PC_DOM_SEED; deck: PC_DOM_SHUFFLE; slice s, draw k: PC_DOM_SLICE index 128 s + k), the same
// operations in the same order -- whitening row sums in ascending column, the norm on four partial sums, the chord's
// coefficients summed in the order of the wave butterfly (a balanced tree over rows of 16 coordinates), the closed form
// along the chord, derived parameters summed in ascending coordinate.  tests/test_gpu_parity.py runs both on the same
// problems and compares every row.
//
// Restates SliceSampling / slice_sample (chordal_sampling.f90:7-92, 163-273), GenerateSeed (generate.F90:19-55) for one
// cluster, generate_nhats' whitening (chordal_sampling.f90:73-82), calculate_point (calculate.f90:6-50) for the built-in
// Gaussian (gaussian.f90:25-37) under the uniform box prior (priors.f90:40-55).
//
// Scope (pc_slice_t_ok): what the fused k_slice takes (nDims <= 24, one grade, keyed draws, raw bases in HBM), the built-in
// Gaussian in closed form along the chord, ONE cluster at launch, num_repeats <= 255, derived parameters from the
// per-baby theta rows (the order k_slice uses whenever those rows fit its LDS).
#include "pc_state.h"
#include <cstdlib>

namespace {

struct Q3 { double a, b, c; };

// the lower triangle of the Cholesky factor in the order the whitening uses it: column after column, rows downwards
template <int D>
struct TriMap {
    int aa[D * (D + 1) / 2], bb[D * (D + 1) / 2];
    constexpr TriMap() : aa{}, bb{} { int e = 0; for (int b = 0; b < D; ++b) for (int a = b; a < D; ++a) { aa[e] = a; bb[e] = b; ++e; } }
};

// Sums in the order of wave_sum<NROWS> (pc_dev.h) over lanes = coordinates: a balanced tree over each row of 16 coordinates,
// rows added left to right.  Leaves past nDims are the +0.0 the idle lanes of k_slice hold (the additions of those zeros
// are made, as the butterfly makes them).  Depth first: log2(16) partial results alive, not all the leaves.
template <int DT, int LO, int N, class F>
__device__ __forceinline__ Q3 tree3(const F &leaf)
{
#pragma clang fp contract(off)
    if constexpr (LO >= DT) return Q3{0.0, 0.0, 0.0};
    else if constexpr (N == 1) return leaf(LO);
    else {
        const Q3 l = tree3<DT, LO, N / 2>(leaf);
        if constexpr (LO + N / 2 >= DT && N / 2 >= 16) return l;       // (never: rows are combined below)
        const Q3 r = tree3<DT, LO + N / 2, N / 2>(leaf);
        return Q3{l.a + r.a, l.b + r.b, l.c + r.c};
    }
}
template <int DT, class F>
__device__ __forceinline__ Q3 wave_order_sum3(const F &leaf)
{
#pragma clang fp contract(off)
    const Q3 r0 = tree3<DT, 0, 16>(leaf);
    if constexpr (DT <= 16) return r0;
    else { const Q3 r1 = tree3<DT, 16, 16>(leaf); return Q3{r0.a + r1.a, r0.b + r1.b, r0.c + r1.c}; }
}

// DT = nDims (1 .. 24); UNIT: the prior is the unit hypercube itself (lo = 0, span = 1: theta = cube, bit for bit)
// HELP: a second wavefront per 64 chains works ahead of the first -- the whitened direction and the first eight uniforms of the next
// slice, which depend on nothing the chain does, wait in LDS when the chain gets there (two buffers, one barrier per slice)
template <int DT, bool UNIT, bool HELP>
__device__ __forceinline__ void slice_t_body(const PcState &S, unsigned batch, int nchains, int nrp)
{
#pragma clang fp contract(off)       // every fused multiply-add below is written out: the roundings of k_slice, whatever this kernel's shape suggests to the compiler
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = DT, FW = DT <= 8 ? 8 : (DT <= 16 ? 16 : 24), NP = D * (D + 1) / 2;
#ifdef SLICE_T_DBG
    const long long cB = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = HELP ? (tid >> 6) : 0;
    constexpr int NTH = HELP ? 192 : 64;
    const int chain_raw = blockIdx.x * 64 + lane;
    const bool act = chain_raw < nchains;                 // lanes past the nursery follow its last chain and store nothing
    const int chain = act ? chain_raw : nchains - 1;
    const int nr = S.nr, nT = S.nT;
    const double logzero = S.logzero;
    double *sL = (double *)smem;                          // [D][D] Cholesky factor of the (one) cluster, transposed
    double *sLo = sL + (size_t)D * D;                     // [FW] prior box
    double *sSpan = sLo + FW;                             // [FW]
    unsigned char *sDeck = (unsigned char *)(sSpan + FW); // [64][nrp] each lane's deck of directions
    double *sRow = (double *)(sDeck + (size_t)64 * nrp);  // [64][nT | 1] the babies' records on their way out (nrp is a multiple of 4, 64 nrp of 8)
    double *sNh = sRow + (size_t)64 * (nT | 1);           // HELP: [64][D + 1] direction and width of the slice to come
    double *sU = sNh + (size_t)64 * (D + 1);              // HELP: [64][9] its first eight uniforms (odd stride)
    {
        constexpr TriMap<D> tm{};
        for (int e = tid; e < NP; e += NTH) sL[e] = S.chol[tm.aa[e] * D + tm.bb[e]];      // packed, in the order of use
    }
    if (tid < FW) {
        const bool on = tid < D;
        const double lo = (on && S.prior.lo) ? S.prior.lo[tid] : 0.0;
        const double hi = (on && S.prior.hi) ? S.prior.hi[tid] : 1.0;
        sLo[tid] = lo; sSpan[tid] = hi - lo;
    }
    // ---- GenerateSeed (generate.F90:19-55), one cluster: a live point drawn evenly (select_seed of pc_sample.hip)
    int slot = 0;
    const bool chain_wave = !HELP || wv == 0;             // the wavefront that walks the chains (HELP: the other one makes directions and uniforms)
    if (chain_wave) {
        const double u2s = pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 1u);
        const int ns = S.cl_n[0];
        int is = (int)ceil(u2s * ns);
        is = is < 1 ? 1 : (is > ns ? ns : is);
        slot = S.cl_list[is - 1];
        if (S.seed_override) slot = chain;
    }
    const double contour = S.logLp[0];
    if (act && chain_wave) {
        S.ch_cluster[chain] = 0; S.ch_seed_slot[chain] = slot;
        S.ch_contour[chain] = contour;                                   // nested_sampling.F90:270
        S.ch_epoch[chain] = S.ctl->admin_epoch;
        if (chain == 0) { S.ctl->i_nursery = nchains; S.ctl->batch_id = batch; }
    }
    double x0[D];
    {
        const double *seed = S.live + (size_t)slot * nT;
#pragma unroll
        for (int d = 0; d < D; ++d) x0[d] = chain_wave ? seed[d] : 0.5;
    }
    if (S.pool && act && chain_wave) for (int i = 0; i < nr; ++i) S.ph_cuid[(size_t)S.pool_base + (size_t)chain * nr + i] = PC_CUID_NONE;
    // ---- deck: the first direction stays, the others are Fisher-Yates shuffled (chordal_sampling.f90:135-142, random_utils.F90:505-532)
    unsigned char *deck = sDeck + (size_t)lane * nrp;
    const bool deck_wave = !HELP || wv == 1;
    if (deck_wave) for (int i = 0; i < nr; ++i) deck[i] = (unsigned char)i;
    for (int i = nr - 1; i >= 1 && deck_wave; --i) {
        const double u = pc_uniform(S.k0, S.k1, PC_DOM_SHUFFLE, batch, (uint32_t)chain, (uint32_t)i);
        int j = (int)ceil(u * i);
        j = j < 1 ? 1 : (j > i ? i : j);
        const unsigned char di = deck[i], dj = deck[j];
        deck[i] = dj; deck[j] = di;
    }
    __syncthreads();                                       // (one wave: the factor and the box are in LDS)
    const double mu = S.like.mu, inv_sigma = S.like.inv_sigma, qnorm = S.like.norm;
    const double *rawc = S.nhat_raw + (size_t)chain * S.nb_total * D * D;    // direction v (generation order) at + v * D
    double vv[D];
    {
        const double *p = rawc + (size_t)(deck_wave ? deck[0] : 0) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) vv[d] = deck_wave ? p[d] : 0.0;
    }
    // ---- whitening of a slice's direction: w = L n (chordal_sampling.f90:73), |w|, n^ = w / |w|, width 3 |w| (:80-82)
    //      (row sums in ascending column; the norm on four partial sums, coordinate d on sum d mod 4)
    auto whiten = [&](const double (&vv_)[D], double (&nh)[D], double &w) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");                     // (the factor is read from LDS in every slice: 2 D^2 registers if the compiler keeps it)
        {
#pragma unroll
            for (int d = 0; d < D; ++d) nh[d] = 0.0;
            // the factor streams through a window of registers, W entries at a time in the order of use, the next window on its
            // way from LDS while this one is multiplied; every row sum still takes its columns in ascending order.  The sums are
            // pinned between windows: left alone, the compiler fetches all D (D + 1) / 2 entries first
            constexpr TriMap<D> tm{};
            constexpr int W = 32, NG = (NP + W - 1) / W;
            double wa[W], wb[W];
#pragma unroll
            for (int i = 0; i < W; ++i) { wa[i] = (i < NP) ? sL[i] : 0.0; wb[i] = 0.0; }
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < W; ++i) if ((gq + 1) * W + i < NP) wb[i] = sL[(gq + 1) * W + i];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const int e = gq * W + i;
                    if (e < NP) nh[tm.aa[e]] = fma(wa[i], vv_[tm.bb[e]], nh[tm.aa[e]]);
                }
#pragma unroll
                for (int d = 0; d < D; ++d) asm volatile("" : "+v"(nh[d]));
#pragma unroll
                for (int i = 0; i < W; ++i) wa[i] = wb[i];
            }
            // (k_slice's `p += t t` over an unrolled loop from p = 0: the compiler takes 0 + t0 t0 + t4 t4 as fma(t0, t0, t4 t4))
            double pp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pp[k] = (k < D) ? fma(nh[k], nh[k], (k + 4 < D) ? nh[k + 4] * nh[k + 4] : 0.0) : 0.0;
#pragma unroll
            for (int d = 8; d < D; ++d) pp[d & 3] = fma(nh[d], nh[d], pp[d & 3]);
            const double wn = sqrt((pp[0] + pp[1]) + (pp[2] + pp[3])), iw = 1.0 / wn;
#pragma unroll
            for (int d = 0; d < D; ++d) nh[d] = nh[d] * iw;
            w = wn * 3.0;
        }
    };
    double *bl_row = S.baby_logL + (size_t)chain * nr;
    double *bl_col = S.baby_logL_T + chain;
    const int o_p0 = S.p0, o_d0 = S.d0, o_b0 = S.b0, o_l0 = S.l0, nDer = S.nDer, Bstride = S.B;
    // the records' way out (see the end of the loop): lane's first pair of a record and its stride through the 64 records
    const int RS = nT | 1, chain0 = blockIdx.x * 64, nrows = min(64, nchains - chain0);
    const size_t rstride = (size_t)nr * nT;
    const int Hq = max(nT >> 1, 1), cq0 = lane / Hq, fq0 = lane - cq0 * Hq, cstep = 64 / Hq, fstep = 64 - cstep * Hq;
    // the wave writes the 64 records of slice s out of LDS in runs of consecutive addresses
    auto copy_out = [&](int s) __attribute__((always_inline)) {
        double *out0 = S.babies + ((size_t)chain0 * nr + s) * nT;       // record of the wave's first chain; chain c: + c nr nT
        if ((nT & 1) == 0) {
            // lane l takes the pairs l, l + 64, ... of the wave's 64 records laid end to end: record c, pair f2 -> the next is
            // cstep records and fstep pairs on (one record more when the pair index wraps); both addresses move by increments
            const int H = nT >> 1;
            int f2 = fq0, c = cq0;
            const double *src = sRow + (size_t)cq0 * RS + 2 * fq0;
            double *dst = out0 + (size_t)cq0 * rstride + 2 * fq0;
            const int sstep = cstep * RS + 2 * fstep, swrap = RS - nT;
            const size_t dstep = (size_t)cstep * rstride + 2 * fstep, dwrap = rstride - nT;
            if (nrows == 64) {
                double2 cur = make_double2(src[0], src[1]);               // (the next pair is on its way from LDS while this one is stored)
                for (int i = 0; i < H; ++i) {
                    double *d0 = dst;
                    f2 += fstep; src += sstep; dst += dstep;
                    if (f2 >= H) { f2 -= H; src += swrap; dst += dwrap; }
                    const double2 nxt = (i + 1 < H) ? make_double2(src[0], src[1]) : cur;
                    {   // (through the L2, system scope: the launch does not end on the write-back of the runs' rows -- as k_slice, round 6)
                        typedef double st_v2d __attribute__((ext_vector_type(2)));
                        const st_v2d v2 = {cur.x, cur.y};
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d0), "v"(v2) : "memory");
                    }
                    cur = nxt;
                }
            } else {
                for (int i = 0; i < H; ++i) {
                    if (c < nrows) *(double2 *)dst = make_double2(src[0], src[1]);
                    f2 += fstep; c += cstep; src += sstep; dst += dstep;
                    if (f2 >= H) { f2 -= H; c++; src += swrap; dst += dwrap; }
                }
            }
        } else {
            for (int e = lane; e < nrows * nT; e += 64) { const int c = e / nT, f = e - c * nT; out0[(size_t)c * rstride + f] = sRow[(size_t)c * RS + f]; }
        }
    };
    const uint32_t k0 = S.k0, k1 = S.k1;
    if constexpr (HELP) {
        if (wv >= 1) {
            // ---- the wavefront that works ahead and behind: while the first walks slice s, this one sends the records of slice
            //      s - 1 out (their derived parameters first: gaussian.f90:36-37 from the record's theta) and makes slice s + 1's
            //      direction and uniforms into buffer (s + 1) & 1.  Two barriers per slice: X (the records' LDS is free), Y (the
            //      records of slice s are laid down, slice s + 1's direction is complete)
            // (one buffer: the direction and the uniforms of slice s + 1 are made in registers while the chain walks slice s, and laid
            //  down between the slice's two barriers -- the chain reads them before X and after Y only; 44 KB of LDS for twenty
            //  dimensions, three workgroups to a CU)
            auto compute = [&](int s, double (&nh)[D], double &w, double (&uu)[8]) __attribute__((always_inline)) {
                whiten(vv, nh, w);
                if (s + 1 < nr) {
                    const double *p = rawc + (size_t)deck[s + 1] * D;
#pragma unroll
                    for (int d = 0; d < D; ++d) vv[d] = p[d];
                }
                const uint32_t c0 = ((uint32_t)s * PC_SLICE_STRIDE) >> 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) pc_uniform2(k0, k1, PC_DOM_SLICE, batch, (uint32_t)chain, c0 + (uint32_t)c, uu[2 * c], uu[2 * c + 1]);
            };
            auto publish = [&](const double (&nh)[D], double w, const double (&uu)[8]) __attribute__((always_inline)) {
                double *pn = sNh + (size_t)lane * (D + 1);
#pragma unroll
                for (int d = 0; d < D; ++d) pn[d] = nh[d];
                pn[D] = w;
                double *pu = sU + (size_t)lane * 9;
#pragma unroll
                for (int c = 0; c < 8; ++c) pu[c] = uu[c];
            };
            auto ship = [&](int s) __attribute__((always_inline)) {
                double *mine = sRow + (size_t)lane * RS;
                if (nDer > 0) {
                    double r2 = 0.0;
#pragma unroll
                    for (int d = 0; d < D; ++d) { const double z = mine[o_p0 + d] - mu; r2 = fma(z, z, r2); }
                    const double phi0 = sqrt(r2);
                    mine[o_d0] = phi0;
                    if (nDer >= 2) mine[o_d0 + 1] = fma((double)D, log(phi0), S.like.log_vn);      // pc_log_ball
                    for (int e = 2; e < nDer; ++e) mine[o_d0 + e] = 0.0;
                }
                const double lnew = mine[o_l0];
                if (act) { bl_row[s] = lnew; bl_col[(size_t)s * Bstride] = lnew; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                copy_out(s);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            if (wv == 1) {
                double nh[D], w, uu[8];
                compute(0, nh, w, uu); publish(nh, w, uu);
                pc_lds_barrier();
                for (int s = 0; s < nr; ++s) {
                    if (s + 1 < nr) compute(s + 1, nh, w, uu);
                    pc_lds_barrier();                      // X
                    if (s + 1 < nr) publish(nh, w, uu);
                    pc_lds_barrier();                      // Y
                }
            } else {
                pc_lds_barrier();
                for (int s = 0; s < nr; ++s) {
                    if (s >= 1) ship(s - 1);
                    pc_lds_barrier();                      // X
                    pc_lds_barrier();                      // Y
                }
                ship(nr - 1);
            }
            return;
        }
        __syncthreads();                                   // (slice 0's direction and uniforms are there)
    }
    int nlike = 0;
#ifdef SLICE_T_DBG
    long long cy[6] = {0, 0, 0, 0, 0, 0}; const long long cA = clock64();
#endif
    for (int s = 0; s < nr; ++s, bl_col += Bstride) {
        // ---- whitening of this slice's direction: w = L n (chordal_sampling.f90:73), |w|, n^ = w / |w|, width 3 |w| (:80-82)
        //      (row sums in ascending column; the norm on four partial sums, coordinate d on sum d mod 4)
        double nh[D], w;
#ifdef SLICE_T_DBG
        const long long c0 = clock64();
#endif
        if constexpr (!HELP) whiten(vv, nh, w);
        else {                                              // made by the other wavefront, a slice ahead
            const double *pn = sNh + (size_t)lane * (D + 1);
#pragma unroll
            for (int d = 0; d < D; ++d) nh[d] = pn[d];
            w = pn[D];
        }
        // ---- where the chord leaves the unit hypercube, once per slice: t in [loS, hiS] is inside in every coordinate, t < loO or
        //      t > hiO is outside in one, whatever the rounding of x0 + t n^ (bounds from an approximate reciprocal, with a margin
        //      far above its error and the fused multiply-add's); in the two slivers between, the trial's own test decides
        double loS, hiS, loO, hiO;
        {
            double hi_min = PC_HUGE, lo_max = -PC_HUGE, rmax = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double n = nh[d];
                double r = __builtin_amdgcn_rcp(n);
                r = fma(fma(-n, r, 1.0), r, r);                  // (n = 0: NaN from here on, and fmin / fmax pass it over: no bound from this coordinate)
                const double a = (0.0 - x0[d]) * r, b = (1.0 - x0[d]) * r;
                hi_min = fmin(hi_min, fmax(a, b)); lo_max = fmax(lo_max, fmin(a, b)); rmax = fmax(rmax, fabs(r));
            }
            const double m_hi = fma(1e-9, fabs(hi_min), 1e-15 * rmax), m_lo = fma(1e-9, fabs(lo_max), 1e-15 * rmax);
            hiS = hi_min - m_hi; hiO = hi_min + m_hi; loS = lo_max + m_lo; loO = lo_max - m_lo;
            // (the start point is inside: lo_max <= 0 <= hi_min.  Anything else -- a NaN that got through, a start point on a wall
            //  whose coordinate does not move -- and every trial of this slice takes its own test)
            if (!((lo_max <= 0.0) & (hi_min >= 0.0) & (m_hi < PC_HUGE) & (m_lo < PC_HUGE))) { hiS = -PC_HUGE; loS = PC_HUGE; hiO = PC_HUGE; loO = -PC_HUGE; }
        }
        if (!HELP && s + 1 < nr) {                          // the next direction's raw vector travels while this slice is made
            const double *p = rawc + (size_t)deck[s + 1] * D;
#pragma unroll
            for (int d = 0; d < D; ++d) vv[d] = p[d];
        }
#ifdef SLICE_T_DBG
        const long long c1 = clock64();
#endif
        // ---- the chord's quadratic: logL(x0 + t n^) = qnorm - (qa + 2 qb t + qc t^2) / 2
        const Q3 q = wave_order_sum3<D>([&](int d) -> Q3 {
            double zA, zB;
            if constexpr (UNIT) { zA = (x0[d] - mu) * inv_sigma; zB = nh[d] * inv_sigma; }
            else { zA = (fma(sSpan[d], x0[d], sLo[d]) - mu) * inv_sigma; zB = (sSpan[d] * nh[d]) * inv_sigma; }
            return Q3{fma(zA, zA, 0.0), fma(zA, zB, 0.0), fma(zB, zB, 0.0)};      // (a lane's `0.0 + z z` of k_slice: rounded before the butterfly)
        });
        const double qa = q.a, qb = q.b, qc = q.c;
        // draw k of this slice (k = 0: the bracket; trial q: k = 1 + q)
        const uint32_t idx0 = (uint32_t)s * PC_SLICE_STRIDE;
        uint32_t have_call = 0xFFFFFFFFu; double ua = 0.0, ub = 0.0;
        auto draw = [&](uint32_t k) -> double {
            if constexpr (HELP) { if (k < 8u) return sU[(size_t)lane * 9 + k]; }
            const uint32_t call = (idx0 + k) >> 1;
            if (call != have_call) { pc_uniform2(k0, k1, PC_DOM_SLICE, batch, (uint32_t)chain, call, ua, ub); have_call = call; }
            return (k & 1u) ? ub : ua;
        };
        // calculate_point (calculate.f90:6-50) along the chord: logzero outside the unit hypercube, not counted
        auto eval = [&](double t, bool &outside) -> double {
            bool o = (t > hiO) | (t < loO);
            if (!o && !((t <= hiS) & (t >= loS))) {        // (rare: within a margin of the cube's wall)
                double mn = 0.5, mx = 0.5;
#pragma unroll
                for (int d = 0; d < D; ++d) { const double cb = fma(t, nh[d], x0[d]); mn = fmin(mn, cb); mx = fmax(mx, cb); }
                o = (mn < 0.0) | (mx > 1.0);
            }
            outside = o;
            double lg = qnorm - fma(t, fma(t, qc, 2.0 * qb), qa) / 2.0;
            if (o) lg = logzero;                            // calculate.f90:36-38
            else if (lg > logzero) nlike++;
            return lg;
        };
#ifdef SLICE_T_DBG
        const long long c2 = clock64();
#endif
        // initial bracket (chordal_sampling.f90:213-219)
        const double u0 = draw(0u);
        double tR = (1 - u0) * w, tL = -(u0 * w);
        bool oo;
        double lR = eval(tR, oo), lL = eval(tL, oo);
        // stepping out (:223-236): the two ends step independently of each other -- one loop for both, as long as either goes on
        {
            int iR = 0, iL = 0;
            bool goR = lR >= contour && lR > logzero, goL = lL >= contour && lL > logzero;
            while (goR || goL) {
                if (goR) { iR++; tR = w * iR; lR = eval(tR, oo); goR = lR >= contour && lR > logzero; }
                if (goL) { iL++; tL = -(w * iL); lL = eval(tL, oo); goL = lL >= contour && lL > logzero; }
            }
        }
#ifdef SLICE_T_DBG
        const long long c3 = clock64();
#endif
        // shrinkage (:240-271)
        double lnew = logzero, t_last = 0.0;
        bool ok = false, last_out = false;
        for (int it = 0; it <= 100 && !ok; ++it) {
            const double dl = fabs(tL), dr = fabs(tR);
            const double t = fma(draw(1u + (uint32_t)it), dr + dl, -dl);
            t_last = t;
            lnew = eval(t, last_out);
            if (lnew < contour || lnew <= logzero) { if (t > 0.0) tR = t; else tL = t; }
            else ok = true;
        }
        if (!ok) lnew = logzero;                            // "Non deterministic loglikelihood"
#ifdef SLICE_T_DBG
        const long long c4 = clock64();
#endif
        // the baby becomes the next start point (chordal_sampling.f90:85-88); its derived parameters (gaussian.f90:36-37)
        // (last_out: a slice that found no point in 101 trials and whose last trial was outside the cube keeps theta = 0 for its
        //  record, as calculate_point leaves it: practically never, and off the common path)
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            x0[d] = fma(t_last, nh[d], x0[d]);
            if constexpr (!HELP) {
                double th;
                if constexpr (UNIT) th = x0[d]; else th = fma(sSpan[d], x0[d], sLo[d]);
                const double z = th - mu;
                r2 = fma(z, z, r2);
            }
        }
        if constexpr (!HELP) if (__builtin_expect(last_out, 0)) {
            r2 = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { const double z = 0.0 - mu; r2 = fma(z, z, r2); }
        }
        // ---- the record of the baby: the lane lays it down in LDS, and a wave writes the 64 records out in runs of consecutive
        //      addresses (a lane's own stores would be 64 cache lines per instruction, 2 nDims + 4 instructions per slice).
        //      HELP: the other wavefront does that, and the derived parameters, while this one walks the next slice
        if constexpr (HELP) {
            __syncthreads();                                // X: the records of the slice before have left LDS
            double *mine = sRow + (size_t)lane * RS;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double th;
                if constexpr (UNIT) th = x0[d]; else th = fma(sSpan[d], x0[d], sLo[d]);
                mine[d] = x0[d]; mine[o_p0 + d] = __builtin_expect(last_out, 0) ? 0.0 : th;
            }
            mine[o_b0] = contour;                           // nested_sampling.F90:260
            mine[o_l0] = lnew;
        } else
        {
            double *mine = sRow + (size_t)lane * RS;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double th;
                if constexpr (UNIT) th = x0[d]; else th = fma(sSpan[d], x0[d], sLo[d]);
                mine[d] = x0[d]; mine[o_p0 + d] = __builtin_expect(last_out, 0) ? 0.0 : th;
            }
            if (nDer > 0) {
                const double phi0 = sqrt(r2);
                mine[o_d0] = phi0;
                if (nDer >= 2) mine[o_d0 + 1] = fma((double)D, log(phi0), S.like.log_vn);      // pc_log_ball
                for (int e = 2; e < nDer; ++e) mine[o_d0 + e] = 0.0;
            }
            mine[o_b0] = contour;                           // nested_sampling.F90:260
            mine[o_l0] = lnew;
            if (act) { bl_row[s] = lnew; *bl_col = lnew; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave's own LDS traffic, in order: no barrier -- with HELP the other wavefront is not here)
            copy_out(s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the records' LDS is free for the next slice once it has been read)
        }
#ifdef SLICE_T_DBG
        { const long long c5 = clock64(); cy[0] += c1 - c0; cy[1] += c2 - c1; cy[2] += c3 - c2; cy[3] += c4 - c3; cy[4] += c5 - c4; }
#endif
        if constexpr (HELP) __syncthreads();               // Y: these records are laid down; the next slice's direction and uniforms are complete
    }
#ifdef SLICE_T_DBG
    if (lane == 0) { unsigned long long *g = (unsigned long long *)S.ctl->dbg; for (int x = 0; x < 5; ++x) atomicAdd(&g[x], (unsigned long long)cy[x]); atomicAdd(&g[5], (unsigned long long)(cA - cB)); atomicAdd(&g[6], (unsigned long long)(clock64() - cA)); atomicAdd(&g[7], 1ull); }
#endif
    if (act) S.ch_nlike[chain] = nlike;
}
template <int DT, bool UNIT>
__global__ __launch_bounds__(64) void k_slice_t(PcState S, unsigned batch, int nchains, int nrp) { slice_t_body<DT, UNIT, false>(S, batch, nchains, nrp); }
template <int DT, bool UNIT, bool HELP>
__global__ __launch_bounds__(HELP ? 192 : 64) void k_slice_t_many(const PcManyRec *R, int nchains, int nrp) { slice_t_body<DT, UNIT, HELP>(pc_many_state(R, blockIdx.y), (unsigned)R[blockIdx.y].ia[0], nchains, nrp); }


static int deck_stride(int nr) { int q = (nr + 3) / 4; if ((q & 1) == 0) q++; return 4 * q; }   // bytes, an odd number of words: lanes on different banks

template <int DT>
static void launch_t(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st)
{
    constexpr int FW = DT <= 8 ? 8 : (DT <= 16 ? 16 : 24);
    const int nrp = deck_stride(S->nr), grid = (nchains + 63) / 64;
    const size_t sh = sizeof(double) * ((size_t)DT * DT + 2 * FW) + (size_t)64 * nrp + sizeof(double) * 64 * (size_t)(S->nT | 1);
    const size_t shh = sh + sizeof(double) * (size_t)64 * (DT + 1 + 9);      // + the buffer of the next slice's direction and uniforms
    const bool unit = S->prior.lo == nullptr && S->prior.hi == nullptr;
    static const bool help_off = std::getenv("PC_SLICE_T_HELP_OFF") != nullptr;
    // (worth it while the helpers find SIMDs of their own: 16 runs 66 ms against 75, 32 runs 102.5 against 106, 64 runs 201 against 184)
    // (the helping wavefronts while a CU has at most two workgroups of the launch: with three -- the LDS would hold them -- the chains'
    //  own wavefronts share SIMDs with the helpers of their neighbours, and forty-eight runs were no faster than without: 132 ms against 130)
    static const long long help_env = std::getenv("PC_SLICE_T_HELP_MAX") ? std::atoll(std::getenv("PC_SLICE_T_HELP_MAX")) : -1;
    const long long help_max = help_env >= 0 ? help_env : 256LL * std::max<long long>(1, std::min<long long>(2, (long long)(156 * 1024) / (long long)shh));
    if (dR && !help_off && shh <= 64 * 1024 && (long long)grid * R <= help_max) {      // runs in step: a second wavefront per 64 chains works a slice ahead
        if (shh > 48 * 1024) { if (unit) pc_need_dyn_lds((const void *)k_slice_t_many<DT, true, true>, shh); else pc_need_dyn_lds((const void *)k_slice_t_many<DT, false, true>, shh); }
        if (unit) hipLaunchKernelGGL((k_slice_t_many<DT, true, true>), dim3(grid, R), dim3(192), shh, st, dR, nchains, nrp);
        else hipLaunchKernelGGL((k_slice_t_many<DT, false, true>), dim3(grid, R), dim3(192), shh, st, dR, nchains, nrp);
        return;
    }
    if (sh > 48 * 1024) {                                         // (long decks and wide records: pc_slice_t_ok keeps it under 64 KB)
        if (dR) { if (unit) pc_need_dyn_lds((const void *)k_slice_t_many<DT, true, false>, sh); else pc_need_dyn_lds((const void *)k_slice_t_many<DT, false, false>, sh); }
        else { if (unit) pc_need_dyn_lds((const void *)k_slice_t<DT, true>, sh); else pc_need_dyn_lds((const void *)k_slice_t<DT, false>, sh); }
    }
    if (dR) {
        if (unit) hipLaunchKernelGGL((k_slice_t_many<DT, true, false>), dim3(grid, R), dim3(64), sh, st, dR, nchains, nrp);
        else hipLaunchKernelGGL((k_slice_t_many<DT, false, false>), dim3(grid, R), dim3(64), sh, st, dR, nchains, nrp);
    } else if (unit) hipLaunchKernelGGL((k_slice_t<DT, true>), dim3(grid), dim3(64), sh, st, *S, batch, nchains, nrp);
    else hipLaunchKernelGGL((k_slice_t<DT, false>), dim3(grid), dim3(64), sh, st, *S, batch, nchains, nrp);
}

}   // namespace

namespace {

// ------------------------------------------------------------------------------------------
// The orthonormal bases of k_nhats<.., 1> (pc_sample.hip: one grade, keyed draws, nDims <= 24) in two kernels that leave the chip
// to the runs next door: the deviates of all bases by a thread per stream call (k_deviates_t: no LDS, as wide as the nursery:
// random_utils.F90:251-263), then normalisation and Gram-Schmidt (random_utils.F90:276-298, 391-399) with thread = vector and
// 64 / nDims bases to a wavefront (k_bases_packed) -- k_nhats' arithmetic operation for operation: dot products on four partial
// sums (and the way the compiler fuses `0 + a0 a0 + a4 a4` there), the projection as one fused multiply-add per coordinate.
// (Tried and dropped: a basis per lane in LDS -- 3.2 KB a basis, a CU holds fifty: the runs queue for LDS.)
// ------------------------------------------------------------------------------------------
// a.a as k_nhats' PC_DOT4(x, a, a) comes out of the compiler: four partial sums over coordinates d = k, k + 4, ...; each starts as
// 0 + a_k a_k + a_{k+4} a_{k+4}, of which ONE product is rounded and the other fused into the addition -- which one is the
// compiler's choice per site (read from the ISA of libpolychord_hip.so: the pivot's q.q fuses a_k everywhere; the norm of a raw
// vector fuses a_4, not a_0, in the kernels compiled for nDims <= 16).  tests/test_gpu_parity.py holds the two kernels together.
template <int D, bool FIRST_RIGHT>
__device__ __forceinline__ double dot4_same(const double (&a)[D])
{
#pragma clang fp contract(off)
    double pp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= D) pp[k] = 0.0;
        else if (k + 4 >= D) pp[k] = fma(a[k], a[k], 0.0);
        else if (FIRST_RIGHT && k == 0) pp[k] = fma(a[k + 4 < D ? k + 4 : 0], a[k + 4 < D ? k + 4 : 0], a[k] * a[k]);
        else pp[k] = fma(a[k], a[k], a[k + 4 < D ? k + 4 : 0] * a[k + 4 < D ? k + 4 : 0]);
    }
#pragma unroll
    for (int d = 8; d < D; ++d) pp[d & 3] = fma(a[d], a[d], pp[d & 3]);
    return (pp[0] + pp[1]) + (pp[2] + pp[3]);
}
template <int D>
__device__ __forceinline__ double dot4(const double (&a)[D], const double (&b)[D])
{
#pragma clang fp contract(off)
    double pp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < D; ++d) pp[d & 3] = fma(a[d], b[d], pp[d & 3]);
    return (pp[0] + pp[1]) + (pp[2] + pp[3]);
}

// The deviates of a thread's own vector, made where they are used (round 6): element d of vector i of basis b of a chain is position
// e0(b) + i nDims + d of the chain's stream, two positions to a Philox call (random_utils.F90:251-263) -- what k_deviates_t writes to
// nhat_raw and k_bases_packed reads back (105 + 99 MB of a launch of sixteen runs) stays in registers.  AS241 as there: the central
// branch in line, the arguments that fall in a tail (15 %) collected in LDS -- a lane's own behind those of the lanes before it, so
// that each finds its results again by counting -- and finished by the whole wavefront together: the same function values.
#define BASES_TQ 768
template <int DMAX>
__device__ __forceinline__ void bases_own_deviates(const PcState &S, unsigned batch, int g, int i, bool active, double (&v)[DMAX], double *tq)
{
    constexpr int NK = DMAX / 2 + 1;
    const int D = S.D, lane = threadIdx.x & 63;
    const int chain = g / S.nb_total, basis = g - chain * S.nb_total;
    const uint32_t ef = (uint32_t)pc_sel(S.g_e0, 0) + (uint32_t)basis * (uint32_t)(D * D) + (uint32_t)(i * D);
    const int p = active ? (int)(ef & 1u) : 0;
    const uint32_t c0 = ef >> 1;
    const int nk = (D + 1 + (__any(p) ? 1 : 0)) / 2;          // calls a lane (wave-uniform)
    double s[2 * NK];
    unsigned tmask = 0u;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        s[2 * k] = 0.0; s[2 * k + 1] = 0.0;
        if (k < nk) {
            double ua = 0.5, ub = 0.5;
            if (active) pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, c0 + (uint32_t)k, ua, ub);
            const bool na = active && 2 * k - p >= 0 && 2 * k - p < D, nb = active && 2 * k + 1 - p < D;      // the positions my vector has
            bool ta, tb;
            const double xa = pc_inv_normal_central(ua, ta), xb = pc_inv_normal_central(ub, tb);
            ta = ta && na; tb = tb && nb;
            s[2 * k] = ta ? ua : xa; s[2 * k + 1] = tb ? ub : xb;
            tmask |= (ta ? 1u : 0u) << (2 * k) | (tb ? 1u : 0u) << (2 * k + 1);
        }
    }
    {
        const int tcount = __popc(tmask);
        int inc = tcount;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        const int total = __shfl(inc, 63), lbase = inc - tcount;
        for (int w0 = 0; w0 < total; w0 += BASES_TQ) {        // (one round: a wavefront has ~200 tail arguments)
            int c = lbase - w0;
#pragma unroll
            for (int j = 0; j < 2 * NK; ++j) if ((tmask >> j) & 1u) { if (c >= 0 && c < BASES_TQ) tq[c] = s[j]; c++; }
            const int nq = min(BASES_TQ, total - w0);
            for (int k = lane; k < nq; k += 64) tq[k] = pc_inv_normal_tail(tq[k]);
            c = lbase - w0;
#pragma unroll
            for (int j = 0; j < 2 * NK; ++j) if ((tmask >> j) & 1u) { if (c >= 0 && c < BASES_TQ) s[j] = tq[c]; c++; }
        }
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) v[d] = (active && d < D) ? (p ? s[d + 1] : s[d]) : 0.0;
}

// step 2: thread = vector, as in k_nhats, but 64 / nDims bases to a wavefront (a basis keeps nDims of a wave's lanes busy) and
// the deviates already made: every thread keeps its vector in registers, the pivot goes round through a few hundred bytes of
// LDS, a barrier per Gram-Schmidt step.  The same operations per vector as k_nhats<DMAX, 64, 1>.
template <int DMAX, bool OWN = false>
__device__ __forceinline__ void bases_packed_body(const PcState &S, int nbases, unsigned batch = 0u)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double tq[OWN ? BASES_TQ : 1];
    const int D = S.D, per = 64 / D;
    const int tid = threadIdx.x, sub = tid / D, i = tid - sub * D;
    const int g = blockIdx.x * per + sub;
    const bool active = sub < per && g < nbases;
    double *Q = (double *)smem + (size_t)(sub < per ? sub : 0) * 2 * DMAX;      // [2][DMAX] the pivot of this basis, double buffered
    double v[DMAX];
    double *raw = S.nhat_raw + ((size_t)(active ? g : 0) * D + i) * D;          // [chain][basis][vector][D], linear in the basis number
    if constexpr (OWN) bases_own_deviates<DMAX>(S, batch, active ? g : 0, i, active, v, tq);
    else {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = (active && d < D) ? raw[d] : 0.0;
    }
    {   // random_direction (random_utils.F90:276-298)
        const double inrm = 1.0 / sqrt(dot4_same<DMAX, (DMAX <= 16)>(v));
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = v[d] * inrm;
    }
    if (i == 0 && active) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) Q[d] = v[d];
    }
    __syncthreads();
    for (int j = 0; j < D; ++j) {                                               // Gram-Schmidt (random_utils.F90:391-399), k_nhats' steps
        const double *q = Q + (size_t)(j & 1) * DMAX;
        double qv[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) qv[d] = active ? q[d] : 0.0;
        const double qq = dot4_same<DMAX, false>(qv), dv = dot4<DMAX>(qv, v);
        if (i == j) {
            const double inrm = 1.0 / sqrt(qq);
#pragma unroll
            for (int d = 0; d < DMAX; ++d) v[d] = v[d] * inrm;
        } else if (active && i > j) {
            const double cproj = dv / qq;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) v[d] = fma(-cproj, qv[d], v[d]);
            if (i == j + 1) {
                double *qn = Q + (size_t)((j + 1) & 1) * DMAX;
#pragma unroll
                for (int d = 0; d < DMAX; ++d) qn[d] = v[d];
            }
        }
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) if (d < D) raw[d] = v[d];
    }
}
template <int DMAX>
__global__ __launch_bounds__(64) void k_bases_packed(PcState S, int nbases) { bases_packed_body<DMAX>(S, nbases); }
template <int DMAX>
__global__ __launch_bounds__(64) void k_bases_packed_many(const PcManyRec *R, int nbases) { bases_packed_body<DMAX>(pc_many_state(R, blockIdx.y), nbases); }
// the two steps in one launch: deviates in registers (bases_own_deviates)
template <int DMAX>
__global__ __launch_bounds__(64) void k_bases_own(PcState S, unsigned batch, int nbases) { bases_packed_body<DMAX, true>(S, nbases, batch); }
template <int DMAX>
__global__ __launch_bounds__(64) void k_bases_own_many(const PcManyRec *R, int nbases) { bases_packed_body<DMAX, true>(pc_many_state(R, blockIdx.y), nbases, (unsigned)R[blockIdx.y].ia[0]); }

// step 1, the deviates: a thread per call of the stream (two positions), no LDS, as wide as the nursery
// Four stream calls (eight deviates) a thread.  AS241's central branch is two polynomials and a division; the tails (15 % of the
// arguments) cost a log and a square root on top, and a wavefront that calls the whole function pays for both on every call because
// some lane is always in a tail.  So: the central branch in line, the tail arguments of a wavefront collected in LDS (ballot + prefix
// count: no atomics) and finished together -- two passes of the tail branch per wavefront instead of eight, the same arithmetic
// (pc_dev.h: the function in two halves): 118 -> ~70 us for sixteen runs of the metric configuration.
#define DEVT_CALLS 4
__device__ __forceinline__ void deviates_t_body(const PcState &S, unsigned batch, int nbases, int NC)
{
    __shared__ double qU[4][2 * DEVT_CALLS * 64];
    __shared__ int qA[4][2 * DEVT_CALLS * 64];
    const int D = S.D, DD = D * D, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ncalls = (long long)nbases * NC;
    int qn = 0;                                              // (wave-uniform) tail arguments queued so far
    auto put = [&](bool want, double u, int dst) {
        bool t = false;
        double x = 0.0;
        if (want) x = pc_inv_normal_central(u, t);
        if (want && !t) S.nhat_raw[dst] = x;
        const unsigned long long m = __ballot(want && t);
        if (want && t) { const int pos = qn + __popcll(m & ((1ull << lane) - 1ull)); qU[wv][pos] = u; qA[wv][pos] = dst; }
        qn += __popcll(m);
    };
#pragma unroll
    for (int k = 0; k < DEVT_CALLS; ++k) {
        const long long t = ((long long)blockIdx.x * DEVT_CALLS + k) * 256 + threadIdx.x;
        const bool in = t < ncalls;
        const int g = in ? (int)(t / NC) : 0, m = in ? (int)(t - (long long)g * NC) : 0;
        const int chain = g / S.nb_total, basis = g - chain * S.nb_total;
        const uint32_t e0 = (uint32_t)pc_sel(S.g_e0, 0) + (uint32_t)basis * DD, e1 = e0 + (uint32_t)DD;
        const uint32_t call = (e0 >> 1) + (uint32_t)m;
        double ua = 0.5, ub = 0.5;
        if (in) pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, call, ua, ub);
        const uint32_t ia = 2 * call, ib = ia + 1;
        const int base = g * DD;                             // (nbases * nDims^2 < 2^31: 64 runs x 1024 chains are launched run by run in the grid's y)
        put(in && ia >= e0 && ia < e1, ua, base + (int)(ia - e0));
        put(in && ib >= e0 && ib < e1, ub, base + (int)(ib - e0));
    }
    __syncthreads();
    for (int k = lane; k < qn; k += 64) S.nhat_raw[qA[wv][k]] = pc_inv_normal_tail(qU[wv][k]);
}
__global__ __launch_bounds__(256) void k_deviates_t(PcState S, unsigned batch, int nbases, int NC) { deviates_t_body(S, batch, nbases, NC); }
__global__ __launch_bounds__(256) void k_deviates_t_many(const PcManyRec *R, int nbases, int NC) { deviates_t_body(pc_many_state(R, blockIdx.y), (unsigned)R[blockIdx.y].ia[0], nbases, NC); }


template <int DT>
static int launch_bases_t(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st)
{
    constexpr int DD = DT * DT, NC = (DD + 1) / 2, DM = DT <= 8 ? 8 : (DT <= 16 ? 16 : 24);
    const int nbases = nchains * S->nb_total;
    const long long ncalls = (long long)nbases * NC;
    const unsigned gdev = (unsigned)((ncalls + 256 * DEVT_CALLS - 1) / (256 * DEVT_CALLS));
    const int perw = 64 / DT, blocksw = (nbases + perw - 1) / perw;
    const size_t shw = sizeof(double) * (size_t)perw * 2 * DM;
    // (settings.ablate bit 12 / PC_BASES_OWN: NOT the default -- measured with sixteen and sixty-four runs in step, one box, bench.py's process:
    //  4.51 / 5.66 G evaluations a second against 4.54 / 5.80 by the two kernels.  A third of the round's HBM bytes less (the deviates' 105 MB
    //  written and 99 MB read back per launch of sixteen runs), 173 us against 91 + 122 -- but the deviates are integer and fp64 issue, not
    //  bytes, and a kernel whose workgroups live through a whole Gram-Schmidt keeps the main stream's k_apply_pool_many out of the compute
    //  units where the short-lived workgroups of k_deviates_t let it in: 47 -> 150 us, and the rounds with an update wait for that chain)
    static const bool own_env = std::getenv("PC_BASES_OWN") != nullptr;
    if (own_env || (S->ablate & 4096)) {
        if (dR) hipLaunchKernelGGL((k_bases_own_many<DM>), dim3(blocksw, R), dim3(64), shw, st, dR, nbases);
        else hipLaunchKernelGGL((k_bases_own<DM>), dim3(blocksw), dim3(64), shw, st, *S, batch, nbases);
        return 0;
    }
    if (dR) {
        hipLaunchKernelGGL(k_deviates_t_many, dim3(gdev, R), dim3(256), 0, st, dR, nbases, NC);
        hipLaunchKernelGGL((k_bases_packed_many<DM>), dim3(blocksw, R), dim3(64), shw, st, dR, nbases);
    } else {
        hipLaunchKernelGGL(k_deviates_t, dim3(gdev), dim3(256), 0, st, *S, batch, nbases, NC);
        hipLaunchKernelGGL((k_bases_packed<DM>), dim3(blocksw), dim3(64), shw, st, *S, nbases);
    }
    return 0;
}

}   // namespace

static bool bases_t_takes(const PcState *S)
{
    static const bool off = std::getenv("PC_BASES_T_OFF") != nullptr;
    return !(off || S->ngrade > 1 || S->seq_mode || S->nhat_raw == nullptr || S->D < 2 || S->D > 24);
}
static int bases_t_dispatch(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st)
{
    if (!bases_t_takes(S)) return 1;
    switch (S->D) {
#define PC_T(n) case n: return launch_bases_t<n>(S, dR, R, batch, nchains, st);
        PC_T(2) PC_T(3) PC_T(4) PC_T(5) PC_T(6) PC_T(7) PC_T(8) PC_T(9) PC_T(10) PC_T(11) PC_T(12)
        PC_T(13) PC_T(14) PC_T(15) PC_T(16) PC_T(17) PC_T(18) PC_T(19) PC_T(20) PC_T(21) PC_T(22) PC_T(23) PC_T(24)
#undef PC_T
    }
    return 1;
}
extern "C" int pc_bases_t_ok(const PcState *S) { return bases_t_takes(S) ? 1 : 0; }
extern "C" int pc_launch_bases_t(const PcState *S, unsigned batch, int nchains, hipStream_t st) { return bases_t_dispatch(S, nullptr, 0, batch, nchains, st); }
extern "C" int pc_launch_bases_t_many(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st) { return bases_t_dispatch(S, dR, R, batch, nchains, st); }

namespace {
}   // namespace

// what k_slice_t takes; phi_lds_fits = the condition under which k_slice keeps the babies' theta rows in LDS (pc_launch_slice_fused)
extern "C" int pc_slice_t_ok(const PcState *S, int ncluster)
{
    static const bool off = std::getenv("PC_SLICE_T_OFF") != nullptr;
    if (off || ncluster != 1) return 0;
    if (S->D > 24 || S->ngrade > 1 || S->seq_mode || S->nhat_raw == nullptr || S->nr > 255) return 0;
    if (S->like.kind != PC_LIKE_GAUSSIAN || (S->ablate & 1)) return 0;
    const size_t sh0 = sizeof(double) * ((size_t)S->D + S->nr) + 16, tb = sizeof(double) * (size_t)S->nr * (S->D + 1);
    if (S->nDer > 0 && sh0 + tb > 48 * 1024) return 0;      // (k_slice would take the derived parameters' other summation order)
    const int FW = S->D <= 8 ? 8 : (S->D <= 16 ? 16 : 24);
    const size_t sh = sizeof(double) * ((size_t)S->D * S->D + 2 * FW) + (size_t)64 * deck_stride(S->nr) + sizeof(double) * 64 * (size_t)(S->nT | 1);
    if (sh > 64 * 1024) return 0;                           // (the wave's LDS: factor, box, 64 decks, 64 records)
    return 1;
}

static int slice_t_dispatch(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st)
{
    switch (S->D) {
#define PC_T(n) case n: launch_t<n>(S, dR, R, batch, nchains, st); return 0;
        PC_T(1) PC_T(2) PC_T(3) PC_T(4) PC_T(5) PC_T(6) PC_T(7) PC_T(8) PC_T(9) PC_T(10) PC_T(11) PC_T(12)
        PC_T(13) PC_T(14) PC_T(15) PC_T(16) PC_T(17) PC_T(18) PC_T(19) PC_T(20) PC_T(21) PC_T(22) PC_T(23) PC_T(24)
#undef PC_T
    }
    return 1;
}
extern "C" int pc_launch_slice_t(const PcState *S, unsigned batch, int nchains, hipStream_t st) { return slice_t_dispatch(S, nullptr, 0, batch, nchains, st); }
extern "C" int pc_launch_slice_t_many(const PcState *S, const PcManyRec *dR, int R, unsigned batch, int nchains, hipStream_t st) { return slice_t_dispatch(S, dR, R, batch, nchains, st); }
