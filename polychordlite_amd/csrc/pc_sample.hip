// pc_sample.hip -- the parallel half of the engine: prior sampling of the initial live set,
// random whitened directions (K0) and the batched slice-sampling chains (K1).
//
// One nursery batch = B independent chains, all seeded from one snapshot of the live set:
// exactly the reference's synchronous farm with nprocs-1 = B
// (src/polychord/nested_sampling.F90:262-286), but the "workers" are wavefronts.
//
//   K0 k_nhats   one workgroup per (chain, basis); thread i owns basis vector i in registers.
//                Restates generate_nhats (chordal_sampling.f90:94-145), random_orthonormal_basis
//                (random_utils.F90:381-403, row-oriented but arithmetically identical Gram-Schmidt),
//                GenerateSeed (generate.F90:19-55) and the whitening nhats = L.nhats
//                (chordal_sampling.f90:73-82).
//   K1 k_slice   one wavefront per chain; lane d owns cube coordinate d (+64k).  Restates
//                SliceSampling / slice_sample (chordal_sampling.f90:7-92, 163-273) and
//                calculate_point (calculate.f90:6-50); likelihood sums are DPP butterflies.
#include "pc_state.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------
// likelihood on a wave: lane owns DPL coordinates (dim = lane + 64*k)
// ------------------------------------------------------------------------------------------
template <int DPL>
struct LaneDims {
    double lo[DPL], span[DPL];   // uniform prior box (priors.f90:40-55)
    double mean[DPL];            // corr gaussian mean / twin gaussian means are derived
    bool on[DPL];
};

template <int DPL, int NROWS>
__device__ __forceinline__ double wsum(double v) { return (DPL > 1) ? wave_sum<4>(v) : wave_sum<NROWS>(v); }

// returns logL of theta (uniform over the wave).  ybuf: per-wave LDS scratch of >= D doubles.
// (KIND >= 0: the likelihood is known when the kernel is compiled -- k_slice's LEAN variants -- and the other branches are not there)
template <int DPL, int NROWS, int KIND = -1>
__device__ __forceinline__ double like_eval(const PcState &S, const double (&th)[DPL], const LaneDims<DPL> &ld,
                                            int lane, double *ybuf)
{
    const int D = S.D;
    const PcLike &L = S.like;
    const int kind = KIND >= 0 ? KIND : L.kind;
    if (kind == PC_LIKE_GAUSSIAN) {            // gaussian.f90:25-34
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) { const double z = (th[k] - L.mu) * L.inv_sigma; s += z * z; }
        s = wsum<DPL, NROWS>(s);
        return L.norm - s / 2.0;
    } else if (kind == PC_LIKE_RASTRIGIN) {    // rastrigin.f90:33
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k)
            if (ld.on[k]) s += 8.515435146961291 /* log(4991.21750) */ + th[k] * th[k] - 10.0 * cos(PC_TWO_PI * th[k]);
        s = wsum<DPL, NROWS>(s);
        return -s;
    } else if (kind == PC_LIKE_TWIN_GAUSSIAN) {  // twin_gaussian.f90:29-46
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k)
            if (ld.on[k]) {
                const int dim = lane + 64 * k;
                const double m1 = dim < 2 ? -0.5 : 0.0, m2 = dim < 2 ? 0.5 : 0.0;
                const double z1 = (th[k] - m1) * L.inv_sigma, z2 = (th[k] - m2) * L.inv_sigma;
                s1 += z1 * z1; s2 += z2 * z2;
            }
        s1 = wsum<DPL, NROWS>(s1); s2 = wsum<DPL, NROWS>(s2);
        return pc_logaddexp(L.norm - s1 / 2.0, L.norm - s2 / 2.0) - 0.6931471805599453;
    } else {                                      // random_gaussian.f90:17-30, utils.F90:1028-1048
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) ybuf[lane + 64 * k] = th[k] - ld.mean[k];
        __syncthreads();                          // one wave per workgroup: cheap
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k)
            if (ld.on[k]) {
                const int a = lane + 64 * k;
                double t = 0.0;
                // invcov is uploaded TRANSPOSED so that lanes read consecutive addresses
                for (int b = 0; b < D; ++b) t += L.invcov[(size_t)b * D + a] * ybuf[b];
                q += (th[k] - ld.mean[k]) * t;
            }
        q = wsum<DPL, NROWS>(q);
        __syncthreads();
        return -((double)D * PC_LOG_TWO_PI + L.logdetcov) / 2.0 - q / 2.0;
    }
}

// derived parameters of an accepted point (lane-uniform results)
template <int DPL, int NROWS, int KIND = -1>
__device__ __forceinline__ void like_phi(const PcState &S, const double (&th)[DPL], const LaneDims<DPL> &ld,
                                         int lane, double &phi0, double &phi1)
{
    phi0 = 0.0; phi1 = 0.0;
    if (S.nDer == 0) return;
    const int kind = KIND >= 0 ? KIND : S.like.kind;
    if (kind == PC_LIKE_GAUSSIAN) {        // gaussian.f90:36-37
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) r2 += (th[k] - S.like.mu) * (th[k] - S.like.mu);
        r2 = wsum<DPL, NROWS>(r2);
        phi0 = sqrt(r2);
        if (S.nDer >= 2) phi1 = pc_log_ball(phi0, S.D, S.like.log_vn);
    } else if (kind == PC_LIKE_TWIN_GAUSSIAN) {   // twin_gaussian.f90:48-52
        const double t0 = readlane_f64(th[0], 0);
        phi0 = (t0 > 0.5) ? 1.0 : -1.0;
    }
}

// ------------------------------------------------------------------------------------------
// initial live points: GenerateLivePoints, linear mode (generate.F90:150-183)
// one wave per attempt; attempts are the oracle's PC_DOM_LIVEGEN streams.
// ------------------------------------------------------------------------------------------
template <int DPL>
__global__ __launch_bounds__(64) void k_generate_live(PcState S, int attempt0, double *rows /* [n][nT] */,
                                                     double *rows_logL)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *ybuf = (double *)smem;
    const int lane = threadIdx.x, a = blockIdx.x, attempt = attempt0 + a;
    const int D = S.D, nT = S.nT;
    LaneDims<DPL> ld;
    double cube[DPL], th[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        const int dim = lane + 64 * k;
        ld.on[k] = dim < D;
        const double lo = (ld.on[k] && S.prior.lo) ? S.prior.lo[dim] : 0.0;
        const double hi = (ld.on[k] && S.prior.hi) ? S.prior.hi[dim] : 1.0;
        ld.lo[k] = lo; ld.span[k] = hi - lo;
        ld.mean[k] = (ld.on[k] && S.like.mean) ? S.like.mean[dim] : 0.0;
        cube[k] = !ld.on[k] ? 0.5 : (S.seq_mode ? pc_seq_uniform(S, (unsigned long long)attempt * D + dim)
                                                 : pc_uniform(S.k0, S.k1, PC_DOM_LIVEGEN, 0u, (uint32_t)attempt, (uint32_t)dim));
        th[k] = ld.lo[k] + ld.span[k] * cube[k];
    }
    const double logL = like_eval<DPL, 4>(S, th, ld, lane, ybuf);
    double phi0, phi1;
    like_phi<DPL, 4>(S, th, ld, lane, phi0, phi1);
    double *row = rows + (size_t)a * nT;
#pragma unroll
    for (int k = 0; k < DPL; ++k)
        if (ld.on[k]) { row[lane + 64 * k] = cube[k]; row[S.p0 + lane + 64 * k] = th[k]; }
    if (lane == 0) {
        if (S.nDer >= 1) row[S.d0] = phi0;
        if (S.nDer >= 2) row[S.d0 + 1] = phi1;
        for (int e = 2; e < S.nDer; ++e) row[S.d0 + e] = 0.0;
        row[S.b0] = S.logzero;                   // generate.F90:163
        row[S.l0] = logL;
        rows_logL[a] = logL;
    }
}

// ------------------------------------------------------------------------------------------
// K0: seed choice + random orthonormal bases + whitening
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void select_seed(const PcState &S, unsigned batch, int chain, int &sel, int &slot)
{   // GenerateSeed, generate.F90:42-53; random_integer_P random_utils.F90:548-576
    const int nc = S.ctl->ncluster;
    if (nc == 1) {       // one cluster: the volume-weighted draw (generate.F90:36-41) can only return it
        sel = 0;
        const double u2s = S.seq_mode ? pc_seq_uniform(S, S.ctl->seq + 1) : pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 1u);
        const int ns = S.cl_n[0];
        int is = (int)ceil(u2s * ns);
        is = is < 1 ? 1 : (is > ns ? ns : is);
        slot = S.cl_list[is - 1];
        if (S.seed_override) slot = chain;
        return;
    }
    double m = S.logXp[0];
    for (int c = 1; c < nc; ++c) m = fmax(m, S.logXp[c]);
    double sum = 0.0;
    for (int c = 0; c < nc; ++c) sum += exp(S.logXp[c] - m);
    const double lse = m + log(sum);
    double norm = 0.0;
    for (int c = 0; c < nc; ++c) norm += exp(S.logXp[c] - lse);
    const double u = S.seq_mode ? pc_seq_uniform(S, S.ctl->seq) : pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 0u);
    double cdf = 0.0;
    sel = nc - 1;
    for (int c = 0; c < nc; ++c) { cdf += exp(S.logXp[c] - lse) / norm; if (u < cdf) { sel = c; break; } }
    const double u2 = S.seq_mode ? pc_seq_uniform(S, S.ctl->seq + 1) : pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 1u);
    const int n = S.cl_n[sel];
    int idx = (int)ceil(u2 * n);
    idx = idx < 1 ? 1 : (idx > n ? n : idx);
    slot = S.cl_list[(size_t)sel * S.Ncap + idx - 1];
    if (S.seed_override) slot = chain;
}

// The deck of a chain (chordal_sampling.f90:135-142, random_utils.F90:505-532: the first direction stays, the others are Fisher-Yates shuffled),
// keyed draws, num_repeats <= 64: the value for position p lives in lane p.  A pure function of (keys, nursery, chain): k_slice makes it, or finds
// it behind the bases where k_nhats<.., 1> left it on the side stream (round 6) under a tag of those four numbers -- a matching tag IS the right deck,
// whatever older run wrote it.
__device__ __forceinline__ int pc_deck_keyed(const PcState &S, unsigned batch, int chain, int lane, int nr)
{
    int dk = lane, jv = 0;
    if (lane >= 1 && lane < nr) {
        const double u = pc_uniform(S.k0, S.k1, PC_DOM_SHUFFLE, batch, (uint32_t)chain, (uint32_t)lane);
        int j = (int)ceil(u * lane);
        jv = j < 1 ? 1 : (j > lane ? lane : j);
    }
    for (int i = nr - 1; i >= 1; --i) {
        const int j = __builtin_amdgcn_readlane(jv, i);
        const int di = __builtin_amdgcn_readlane(dk, i), dj = __builtin_amdgcn_readlane(dk, j);
        dk = (lane == i) ? dj : ((lane == j) ? di : dk);
    }
    return dk;
}
// (everything the deck is a function of; a record that carries this tag carries that deck, whichever run, shape or buffer layout wrote it)
__device__ __forceinline__ unsigned long long pc_deck_tag(const PcState &S, unsigned batch, int chain, int nr)
{
    return ((((unsigned long long)S.k0 << 32) | (unsigned long long)S.k1) ^ ((unsigned long long)batch * 0x9E3779B97F4A7C15ull)
            ^ ((unsigned long long)(unsigned)chain * 0xC2B2AE3D27D4EB4Full) ^ ((unsigned long long)(unsigned)nr << 56)) | 1ull;
}
// behind the bases of a nursery (nDims <= 24: the engine allocates the tail): one record of 66 ints a chain, [tag lo, tag hi, deck[64]]
__device__ __forceinline__ int *pc_deck_record(const PcState &S, int chain) { return (int *)(S.nhat_raw + (size_t)S.B * S.nb_total * S.D * S.D) + (size_t)chain * 66; }

// select_seed by a whole wavefront (several clusters): the same numbers -- every exponential and quotient has the operands of the loops above and
// every sum their order -- with the clusters' volumes loaded and exponentiated a lane each and only the additions in sequence (v_readlane): one
// thread's three loops of a dependent global load + exponential per cluster were ~30 k cycles in front of every chain at two dozen clusters.
__device__ __forceinline__ void select_seed_wave(const PcState &S, unsigned batch, int chain, int lane, int &sel, int &slot)
{
    const int nc = S.ctl->ncluster;
    if (nc == 1 || S.seq_mode) {
        int a = 0, b = 0;
        if (lane == 0) select_seed(S, batch, chain, a, b);
        sel = __builtin_amdgcn_readfirstlane(a); slot = __builtin_amdgcn_readfirstlane(b);
        return;
    }
    double m = -PC_HUGE;
    for (int c0 = 0; c0 < nc; c0 += 64) { const int c = c0 + lane; m = fmax(m, c < nc ? S.logXp[c] : -PC_HUGE); }
    m = wave_max(m);
    double sum = 0.0;
    for (int c0 = 0; c0 < nc; c0 += 64) {
        const int c = c0 + lane;
        const double e = c < nc ? exp(S.logXp[c] - m) : 0.0;
        const int n = nc - c0 < 64 ? nc - c0 : 64;
        for (int q = 0; q < n; ++q) sum += readlane_f64(e, q);
    }
    const double lse = m + log(sum);
    double norm = 0.0;
    for (int c0 = 0; c0 < nc; c0 += 64) {
        const int c = c0 + lane;
        const double t = c < nc ? exp(S.logXp[c] - lse) : 0.0;
        const int n = nc - c0 < 64 ? nc - c0 : 64;
        for (int q = 0; q < n; ++q) norm += readlane_f64(t, q);
    }
    const double u = pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 0u);
    double cdf = 0.0;
    sel = nc - 1;
    bool found = false;
    for (int c0 = 0; c0 < nc && !found; c0 += 64) {
        const int c = c0 + lane;
        const double r = c < nc ? exp(S.logXp[c] - lse) / norm : 0.0;
        const int n = nc - c0 < 64 ? nc - c0 : 64;
        for (int q = 0; q < n; ++q) { cdf += readlane_f64(r, q); if (u < cdf) { sel = c0 + q; found = true; break; } }
    }
    const double u2 = pc_uniform(S.k0, S.k1, PC_DOM_SEED, batch, (uint32_t)chain, 1u);
    const int n = S.cl_n[sel];
    int idx = (int)ceil(u2 * n);
    idx = idx < 1 ? 1 : (idx > n ? n : idx);
    slot = S.cl_list[(size_t)sel * S.Ncap + idx - 1];
    if (S.seed_override) slot = chain;
}

// PART 0: the whole kernel.  PART 1 / 2: the two halves of a split launch -- the orthonormal bases depend on nothing
// but the keys and the batch number (1: they go to S.nhat_raw, on a side stream while the previous nursery is being
// consumed), seed selection and whitening need the live set and the covariance of the moment (2).
template <int DMAX, int NT, int PART = 0>
__global__ __launch_bounds__(NT) void k_nhats(PcState S, unsigned batch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = S.D, nr = S.nr;
    double *G = (double *)smem;          // [D*D] deviates, vector-major
    double *Q = G + (size_t)(D + 8) * (D + 8);   // [2][DMAX] broadcast of the current vector (double buffered)
    int *sh = (int *)(Q + 2 * DMAX);     // [2] chosen cluster, seed slot
#ifdef NHATS_DBG
    long long ncyc[8]; ncyc[0] = clock64();
#endif
    const int tid = threadIdx.x, chain = blockIdx.y;
    // grade of this block and its basis within the grade (chordal_sampling.f90:119-130): the basis spans the
    // parameters off..D-1, its vectors carry zeros in front; one grade: off = 0, Dg = D
    int grade, basis;
    pc_grade_of_basis(S, blockIdx.x, grade, basis);
    const int off = pc_sel(S.g_off, grade), Dg = D - off, nrg = pc_sel(S.g_nr, grade), col0 = pc_sel(S.g_col0, grade);
    if (PART != 1 && tid == 0) {
        int sel, slot;
        select_seed(S, batch, chain, sel, slot);
        sh[0] = sel; sh[1] = slot;
        if (blockIdx.x == 0) {
            S.ch_cluster[chain] = sel; S.ch_seed_slot[chain] = slot;
            S.ch_contour[chain] = S.logLp[sel];          // nested_sampling.F90:270
            S.ch_epoch[chain] = S.ctl->admin_epoch;
            if (chain == 0) { S.ctl->i_nursery = gridDim.y; S.ctl->batch_id = batch; }
        }
    }
#ifdef NHATS_DBG
    ncyc[1] = clock64();
#endif
    if constexpr (PART == 1 && NT == 64) {
        // drawn ahead with the bases: the chain's deck (k_slice's 39 dependent swaps, off its wavefront)
        if (blockIdx.x == 0 && S.D <= 24 && S.ngrade <= 1 && !S.seq_mode && nr <= 64) {
            const int dk = pc_deck_keyed(S, batch, chain, tid, nr);
            int *rec = pc_deck_record(S, chain);
            rec[2 + tid] = dk;
            const unsigned long long tg = pc_deck_tag(S, batch, chain, nr);
            if (tid == 0) { rec[0] = (int)(unsigned)tg; rec[1] = (int)(unsigned)(tg >> 32); }
        }
    }
    const int i = tid;
    const bool active = i < Dg;
    double v[DMAX];
    double *raw = S.nhat_raw + (((size_t)chain * gridDim.x + blockIdx.x) * D + (i < D ? i : 0)) * D;   // [chain][basis][vector][D]
    if constexpr (PART == 2) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = (active && d < D) ? raw[d] : 0.0;
        __syncthreads();
    } else {
    // gaussian deviates: running index of stream (batch, chain) in PC_DOM_NHAT, grade after grade, basis after basis,
    // vector after vector (seq_mode: after the two seed draws: generate_nhats inside SliceSampling)
    const uint32_t eoff = S.seq_mode ? (uint32_t)S.ctl->seq + 2u : 0u;
    const uint32_t e0 = eoff + (uint32_t)pc_sel(S.g_e0, grade) + (uint32_t)basis * Dg * Dg, e1 = e0 + (uint32_t)Dg * Dg;
    if (off > 0) {
        for (int e = tid; e < Dg * D; e += NT) G[e] = 0.0;
        __syncthreads();
    }
    for (uint32_t call = (e0 >> 1) + tid; call <= ((e1 - 1) >> 1); call += NT) {
        double ua, ub;
        if (S.seq_mode) pc_uniform2(S.k0, S.k1, PC_DOM_SEQ, 0u, 0u, call, ua, ub);
        else pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, call, ua, ub);
        const uint32_t ia = 2 * call, ib = 2 * call + 1;
        // element x of the basis = coordinate off + x % Dg of vector x / Dg
        if (ia >= e0 && ia < e1) { const uint32_t x = ia - e0; G[off == 0 ? x : (x / Dg) * D + off + x % Dg] = pc_inv_normal_cdf(ua); }
        if (ib >= e0 && ib < e1) { const uint32_t x = ib - e0; G[off == 0 ? x : (x / Dg) * D + off + x % Dg] = pc_inv_normal_cdf(ub); }
    }
    __syncthreads();
#ifdef NHATS_DBG
    ncyc[2] = clock64();
#endif
#pragma unroll
    for (int d = 0; d < DMAX; ++d) v[d] = (active && d < D) ? G[(size_t)i * D + d] : 0.0;
    // dot products run on four partial sums (the summation order every other kernel of these bases reproduces)
#define PC_DOT4(RES, A, B) { double p0_ = 0.0, p1_ = 0.0, p2_ = 0.0, p3_ = 0.0; \
        _Pragma("unroll") for (int d = 0; d < DMAX; d += 4) { \
            p0_ += (A)[d] * (B)[d]; p1_ += (A)[d + 1] * (B)[d + 1]; p2_ += (A)[d + 2] * (B)[d + 2]; p3_ += (A)[d + 3] * (B)[d + 3]; } \
        RES = (p0_ + p1_) + (p2_ + p3_); }
    // (register vectors are zero padded up to DMAX, LDS rows up to D + 8: no per-element bounds tests, which
    //  cost a scalar compare-and-branch each)
    // random_direction (random_utils.F90:276-298): normalise the raw deviates
    {
        double n2;
        PC_DOT4(n2, v, v)
        const double inrm = 1.0 / sqrt(n2);
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = v[d] * inrm;
    }
#ifdef NHATS_DBG
    ncyc[3] = clock64();
#endif
    // Gram-Schmidt, row oriented (the projections of random_utils.F90:391-399 in the same order): at step j
    // the vector v_j -- already orthogonal to its predecessors, not yet normalised -- is broadcast
    // through LDS; every later vector removes its component along it, (v.q / q.q) q, while thread j
    // normalises.  One barrier per step; q.q is recomputed by everybody instead of being broadcast.
    double *Qb = Q;                                   // [2][QS] double buffer
    const int QS = DMAX;
    if (i == 0) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) Qb[d] = v[d];
    }
    __syncthreads();
#define PC_GS_STEP(QV) { \
        double qq, dv; \
        PC_DOT4(qq, QV, QV) \
        PC_DOT4(dv, QV, v) \
        if (i == j) { \
            const double inrm = 1.0 / sqrt(qq); \
            _Pragma("unroll") for (int d = 0; d < DMAX; ++d) v[d] = v[d] * inrm; \
        } else if (active && i > j) { \
            const double cproj = dv / qq; \
            _Pragma("unroll") for (int d = 0; d < DMAX; ++d) v[d] = v[d] - cproj * (QV)[d]; \
            if (i == j + 1) { \
                double *qn = Qb + (size_t)((j + 1) & 1) * QS; \
                _Pragma("unroll") for (int d = 0; d < DMAX; ++d) qn[d] = v[d]; \
            } \
        } }
    for (int j = 0; j < Dg; ++j) {
        const double *q = Qb + (size_t)(j & 1) * QS;
        if constexpr (DMAX <= 32) {
            double qv[DMAX];                          // registers: one LDS pass per step
#pragma unroll
            for (int d = 0; d < DMAX; ++d) qv[d] = q[d];
            PC_GS_STEP(qv)
        } else {
            PC_GS_STEP(q)                             // large nDims: stream q from LDS, v alone fills the registers
        }
        __syncthreads();
    }
#undef PC_GS_STEP
    if constexpr (PART == 1) {
        if (active) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < D) raw[d] = v[d];
        }
        return;
    }
    }   // PART != 2
#ifdef NHATS_DBG
    ncyc[4] = clock64();
#endif
    // whitening  w = L.n  (chordal_sampling.f90:73)
    const int col = col0 + basis * Dg + i;
    const bool wanted = basis * Dg + i < nrg;         // the last basis of a grade is truncated
    if constexpr (DMAX <= 32) {
        // the deviate buffer is free: it receives the Cholesky factor, every thread multiplies its own
        // vector from registers; the D row sums are independent chains (row a adds b = 0..a in order)
        const double *Lg = S.chol + (size_t)sh[0] * D * D;
        for (int e = tid; e < D * D; e += NT) G[e] = Lg[e];
        __syncthreads();
        if (active && wanted) {
            double t[DMAX];
#pragma unroll
            for (int a = 0; a < DMAX; ++a) t[a] = 0.0;
#pragma unroll
            for (int b = 0; b < DMAX; ++b)
#pragma unroll
                for (int a = b; a < DMAX; ++a) t[a] += G[(size_t)a * D + b] * v[b];
#pragma unroll
            for (int a = 0; a < DMAX; ++a) if (a >= D) t[a] = 0.0;     // rows past D read padding
            double n2;
            PC_DOT4(n2, t, t)
            const double w = sqrt(n2), iw = 1.0 / w;           // chordal_sampling.f90:80-82
            double *out = S.nhat + ((size_t)chain * nr + col) * D;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < D) out[d] = t[d] * iw;
            S.nhat_w[(size_t)chain * nr + col] = w * 3.0;
        }
    } else {
        // large nDims: the finished vectors go back to LDS (odd row stride: no bank conflicts when every thread
        // walks its own row) and the Cholesky factor streams through a tile of rows in the padding of the buffer,
        // loaded cooperatively -- the product used to read every L(a,b) from global memory inside a dependent loop
        const int DS = D + 1, TR = 12;
        double *Lt = G + (size_t)D * DS;             // [TR][D]
        __syncthreads();
        if (active) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < D) G[(size_t)i * DS + d] = v[d];
        }
        const double *Lc = S.chol + (size_t)sh[0] * D * D;
        double *mine = G + (size_t)i * DS;
        for (int a_hi = D - 1; a_hi >= 0; a_hi -= TR) {     // in place: row a only needs n[0..a], rows go downwards
            const int a_lo = max(0, a_hi - TR + 1), nrow = a_hi - a_lo + 1;
            __syncthreads();
            for (int e = tid; e < nrow * D; e += NT) Lt[e] = Lc[(size_t)a_lo * D + e];
            __syncthreads();
            if (active && wanted) {
                for (int a = a_hi; a >= a_lo; --a) {
                    const double *Lr = Lt + (size_t)(a - a_lo) * D;
                    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
                    int bb = 0;
                    for (; bb + 3 <= a; bb += 4) {
                        t0 += Lr[bb] * mine[bb]; t1 += Lr[bb + 1] * mine[bb + 1]; t2 += Lr[bb + 2] * mine[bb + 2]; t3 += Lr[bb + 3] * mine[bb + 3];
                    }
                    for (; bb <= a; ++bb) t0 += Lr[bb] * mine[bb];
                    mine[a] = (t0 + t1) + (t2 + t3);
                }
            }
        }
        if (active && wanted) {
            double n0 = 0.0, n1 = 0.0;
            int d = 0;
            for (; d + 1 < D; d += 2) { n0 += mine[d] * mine[d]; n1 += mine[d + 1] * mine[d + 1]; }
            if (d < D) n0 += mine[d] * mine[d];
            const double w = sqrt(n0 + n1);                   // chordal_sampling.f90:80-82
            Q[i] = 1.0 / w;
            S.nhat_w[(size_t)chain * nr + col] = w * 3.0;
        }
        __syncthreads();
        // rows leave coalesced
        for (int r = 0; r < Dg && basis * Dg + r < nrg; ++r) {
            double *out = S.nhat + ((size_t)chain * nr + col0 + basis * Dg + r) * D;
            const double iw = Q[r];
            for (int d = tid; d < D; d += NT) out[d] = G[(size_t)r * DS + d] * iw;
        }
    }
#ifdef NHATS_DBG
    ncyc[5] = clock64();
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) for (int x = 0; x < 5; ++x) S.ctl->dbg[x] += ncyc[x + 1] - ncyc[x];
#endif
#undef PC_DOT4
}

// ------------------------------------------------------------------------------------------
// K0 for 128 < nDims <= 256: a basis no longer fits in a CU's LDS (256 x 256 fp64 = 512 KB).  One workgroup of 256 threads per
// (chain, basis), thread i owns vector i; the basis lives in a global scratch block, coordinate-major (V[d][i]: the
// threads of a wave touch consecutive addresses, the pivot column is a broadcast read served by the L1/L2).  Same
// arithmetic as k_nhats: row-oriented Gram-Schmidt against the not yet normalised pivot, dot products on four partial
// sums, whitening in place from the last row upwards.
// ------------------------------------------------------------------------------------------
#define PC_BIG_NT 256
#define PC_BIG_P 16
__global__ __launch_bounds__(PC_BIG_NT) void k_nhats_big(PcState S, unsigned batch)
{
    __shared__ int sh[2];
    __shared__ double iwv[PC_BIG_NT];
    __shared__ double Pq[PC_BIG_P * PC_BIG_NT];        // pivot panel, [pivot][coordinate]
    __shared__ double pqq[PC_BIG_P];
    const int D = S.D, nr = S.nr, tid = threadIdx.x, chain = blockIdx.y, i = tid;
    int grade, basis;
    pc_grade_of_basis(S, blockIdx.x, grade, basis);
    const int off = pc_sel(S.g_off, grade), Dg = D - off, nrg = pc_sel(S.g_nr, grade), col0 = pc_sel(S.g_col0, grade);
    if (tid == 0) {
        int sel, slot;
        select_seed(S, batch, chain, sel, slot);
        sh[0] = sel; sh[1] = slot;
        if (blockIdx.x == 0) {
            S.ch_cluster[chain] = sel; S.ch_seed_slot[chain] = slot;
            S.ch_contour[chain] = S.logLp[sel];          // nested_sampling.F90:270
            S.ch_epoch[chain] = S.ctl->admin_epoch;
            if (chain == 0) { S.ctl->i_nursery = gridDim.y; S.ctl->batch_id = batch; }
        }
    }
    double *V = S.nhat_raw + ((size_t)chain * gridDim.x + blockIdx.x) * D * PC_BIG_NT;   // [D][256]
    // vectors past the truncated end of a grade's last basis are never used and nothing depends on them
    const int nvec = min(Dg, nrg - basis * Dg);
    const bool active = i < nvec;
    const uint32_t eoff = S.seq_mode ? (uint32_t)S.ctl->seq + 2u : 0u;
    const uint32_t e0 = eoff + (uint32_t)pc_sel(S.g_e0, grade) + (uint32_t)basis * Dg * Dg, e1 = e0 + (uint32_t)Dg * Dg;
    for (int e = tid; e < off * PC_BIG_NT; e += PC_BIG_NT) V[e] = 0.0;
    for (uint32_t call = (e0 >> 1) + tid; call <= ((e1 - 1) >> 1); call += PC_BIG_NT) {
        double ua, ub;
        if (S.seq_mode) pc_uniform2(S.k0, S.k1, PC_DOM_SEQ, 0u, 0u, call, ua, ub);
        else pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, call, ua, ub);
        const uint32_t ia = 2 * call, ib = 2 * call + 1;
        if (ia >= e0 && ia < e1) { const uint32_t x = ia - e0; V[(size_t)(off + x % Dg) * PC_BIG_NT + x / Dg] = pc_inv_normal_cdf(ua); }
        if (ib >= e0 && ib < e1) { const uint32_t x = ib - e0; V[(size_t)(off + x % Dg) * PC_BIG_NT + x / Dg] = pc_inv_normal_cdf(ub); }
    }
    __syncthreads();
    double *mine = V + i;
    auto dot_own = [&](const double *q) {              // q . mine (q == mine: the squared norm), four partial sums
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        int d = off;
        for (; d + 3 < D; d += 4) {
            p0 += q[(size_t)d * PC_BIG_NT] * mine[(size_t)d * PC_BIG_NT]; p1 += q[(size_t)(d + 1) * PC_BIG_NT] * mine[(size_t)(d + 1) * PC_BIG_NT];
            p2 += q[(size_t)(d + 2) * PC_BIG_NT] * mine[(size_t)(d + 2) * PC_BIG_NT]; p3 += q[(size_t)(d + 3) * PC_BIG_NT] * mine[(size_t)(d + 3) * PC_BIG_NT];
        }
        for (; d < D; ++d) p0 += q[(size_t)d * PC_BIG_NT] * mine[(size_t)d * PC_BIG_NT];
        return (p0 + p1) + (p2 + p3);
    };
    auto norm2 = [&](const double *q) {                // q . q in the same four-way order
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        int d = off;
        for (; d + 3 < D; d += 4) {
            const double a = q[(size_t)d * PC_BIG_NT], b = q[(size_t)(d + 1) * PC_BIG_NT], c = q[(size_t)(d + 2) * PC_BIG_NT], e = q[(size_t)(d + 3) * PC_BIG_NT];
            p0 += a * a; p1 += b * b; p2 += c * c; p3 += e * e;
        }
        for (; d < D; ++d) { const double a = q[(size_t)d * PC_BIG_NT]; p0 += a * a; }
        return (p0 + p1) + (p2 + p3);
    };
    if (active) {                                      // random_direction (random_utils.F90:276-298)
        const double inrm = 1.0 / sqrt(dot_own(mine));
        for (int d = off; d < D; ++d) mine[(size_t)d * PC_BIG_NT] *= inrm;
    }
    __syncthreads();
    // Gram-Schmidt (random_utils.F90:391-399) in panels of sixteen pivots.  A panel is staged in LDS and orthogonalised
    // there (modified Gram-Schmidt, one wave per later vector of the panel, lanes over the coordinates); every vector
    // behind the panel then takes its sixteen projections in one pass over its coordinates and removes them in a
    // second: three global accesses per coordinate and panel instead of five per coordinate and pivot.  (Within a
    // panel the pivots are mutually orthogonal, so projecting on all of them at once differs from one after the other
    // by round-off only.)  Pivots stay unnormalised: (v.q / q.q) q; every owner normalises after the loop.
    {
        const int wv = tid >> 6, lane = tid & 63;
        for (int j0 = 0; j0 < nvec; j0 += PC_BIG_P) {
            const int np = min(PC_BIG_P, nvec - j0);
            for (int e = tid; e < np * D; e += PC_BIG_NT) {           // vector index fastest: 128-B runs of the scratch
                const int pp = e % np, d = e / np;
                Pq[pp * PC_BIG_NT + d] = V[(size_t)d * PC_BIG_NT + j0 + pp];
            }
            for (int e = np * PC_BIG_NT + tid; e < PC_BIG_P * PC_BIG_NT; e += PC_BIG_NT) Pq[e] = 0.0;
            __syncthreads();
            for (int pp = 0; pp + 1 < np; ++pp) {
                const double *qp = Pq + pp * PC_BIG_NT;
                for (int q = pp + 1 + wv; q < np; q += 4) {
                    double *vq = Pq + q * PC_BIG_NT;
                    double dq = 0.0, dd = 0.0;
                    for (int d = off + lane; d < D; d += 64) { const double a = qp[d], b2 = vq[d]; dq += a * b2; dd += a * a; }
                    dq = wave_sum<4>(dq); dd = wave_sum<4>(dd);
                    const double c = dq / dd;
                    for (int d = off + lane; d < D; d += 64) vq[d] -= c * qp[d];
                }
                __syncthreads();
            }
            for (int pp = wv; pp < np; pp += 4) {
                const double *qp = Pq + pp * PC_BIG_NT;
                double dd = 0.0;
                for (int d = off + lane; d < D; d += 64) dd += qp[d] * qp[d];
                dd = wave_sum<4>(dd);
                if (lane == 0) pqq[pp] = dd;
            }
            for (int e = tid; e < np * D; e += PC_BIG_NT) {
                const int pp = e % np, d = e / np;
                V[(size_t)d * PC_BIG_NT + j0 + pp] = Pq[pp * PC_BIG_NT + d];
            }
            __syncthreads();
            if (active && i >= j0 + np) {
                double c[PC_BIG_P];
#pragma unroll
                for (int pp = 0; pp < PC_BIG_P; ++pp) c[pp] = 0.0;
                for (int d = off; d < D; ++d) {
                    const double x = mine[(size_t)d * PC_BIG_NT];
#pragma unroll
                    for (int pp = 0; pp < PC_BIG_P; ++pp) c[pp] += x * Pq[pp * PC_BIG_NT + d];
                }
#pragma unroll
                for (int pp = 0; pp < PC_BIG_P; ++pp) c[pp] = pp < np ? c[pp] / pqq[pp] : 0.0;
                for (int d = off; d < D; ++d) {
                    double x = mine[(size_t)d * PC_BIG_NT];
#pragma unroll
                    for (int pp = 0; pp < PC_BIG_P; ++pp) x -= c[pp] * Pq[pp * PC_BIG_NT + d];
                    mine[(size_t)d * PC_BIG_NT] = x;
                }
            }
            __syncthreads();
        }
    }
    if (active) {
        const double inrm = 1.0 / sqrt(dot_own(mine));
        for (int d = off; d < D; ++d) mine[(size_t)d * PC_BIG_NT] *= inrm;
    }
    // whitening  w = L.n  (chordal_sampling.f90:73), in place, sixteen rows of L at a time from the last row upwards
    // (row a only needs n[0..a]): the rows wait in LDS, a thread reads each of its coordinates once per tile and
    // feeds sixteen independent sums in ascending column order (one row at a time re-read the vector nDims/2 times:
    // 20 GB per launch at nDims = 200)
    {
        const double *Lc = S.chol + (size_t)sh[0] * D * D;
        for (int a_hi = D - 1; a_hi >= 0; a_hi -= PC_BIG_P) {
            const int a_lo = max(0, a_hi - PC_BIG_P + 1), nrow = a_hi - a_lo + 1;
            __syncthreads();                              // the previous tile (or the last pivot panel) is no longer read
            for (int e = tid; e < PC_BIG_P * PC_BIG_NT; e += PC_BIG_NT) {
                const int r = e / PC_BIG_NT, b2 = e % PC_BIG_NT;
                Pq[e] = (r < nrow && b2 <= a_hi) ? Lc[(size_t)(a_lo + r) * D + b2] : 0.0;
            }
            __syncthreads();
            if (active) {
                double acc[PC_BIG_P];
#pragma unroll
                for (int r = 0; r < PC_BIG_P; ++r) acc[r] = 0.0;
                for (int b2 = 0; b2 <= a_hi; ++b2) {
                    const double x = mine[(size_t)b2 * PC_BIG_NT];
#pragma unroll
                    for (int r = 0; r < PC_BIG_P; ++r) acc[r] += Pq[r * PC_BIG_NT + b2] * x;
                }
#pragma unroll
                for (int r = 0; r < PC_BIG_P; ++r) if (r < nrow) mine[(size_t)(a_lo + r) * PC_BIG_NT] = acc[r];
            }
        }
    }
    if (active) {
        double n0 = 0.0, n1 = 0.0;
        int d = 0;
        for (; d + 1 < D; d += 2) { const double a = mine[(size_t)d * PC_BIG_NT], b = mine[(size_t)(d + 1) * PC_BIG_NT]; n0 += a * a; n1 += b * b; }
        if (d < D) { const double a = mine[(size_t)d * PC_BIG_NT]; n0 += a * a; }
        const double w = sqrt(n0 + n1);                   // chordal_sampling.f90:80-82
        iwv[i] = 1.0 / w;
        S.nhat_w[(size_t)chain * nr + col0 + basis * Dg + i] = w * 3.0;
    }
    __syncthreads();
    for (int r = 0; r < nvec; ++r) {
        double *out = S.nhat + ((size_t)chain * nr + col0 + basis * Dg + r) * D;
        const double iw = iwv[r];
        for (int d = tid; d < D; d += PC_BIG_NT) out[d] = V[(size_t)d * PC_BIG_NT + r] * iw;
    }
}

// ------------------------------------------------------------------------------------------
// K0 for 64 < nDims <= 128: FOUR threads per basis vector (32 coordinates each, all in registers), 512 threads per
// basis.  Dot products are 8 deep instead of 32, the four partial sums meet through DPP quad permutes, the pivot
// travels through 2 KB of LDS, and the whitening streams the Cholesky factor through 32-row LDS tiles whose rows
// line up with the four coordinate blocks.  (One thread per vector needed 128 fp64 registers and spilled; vectors
// kept in LDS made every Gram-Schmidt step an LDS round trip per coordinate: 1.5 ms per launch at nDims = 100.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double quad_sum(double v)
{
    v += dpp_f64<PC_DPP_XOR1>(v);
    v += dpp_f64<PC_DPP_XOR2>(v);
    return v;
}
// HV = coordinates per thread: 8 (nDims <= 32), 16 (<= 64), 32 (<= 128); 4*HV vectors of 4*HV padded coordinates
// PART (HV = 32, one grade): 0 = the whole kernel; 1 = deviates + Gram-Schmidt only, the orthonormal basis goes to nhat_raw
// in the operand layout of k_whiten, which does the rest.  Part 1 touches nothing the contraction changes: it is drawn on
// the side stream while earlier nurseries are sampled and consumed.  (The production first half is k_basis -- Gram-Schmidt
// in panels; this one, pivot by pivot, stays as the reference-order variant behind PC_BASIS_PANEL_OFF.)
template <int HV, int PART = 0>
__global__ __launch_bounds__(16 * HV) void k_nhats_q(PcState S, unsigned batch)
{
    // (fused where ONE expression says a * b + c and nowhere else: the body is compiled as the whole kernel, as its two halves and with
    //  the run in the grid, and what the optimiser fuses across statements differed between those -- a run in step must be the run alone)
#pragma clang fp contract(on)
#include "pc_nhats_q_body.inc"
}
template <int HV, int PART = 0>
__global__ __launch_bounds__(16 * HV) void k_nhats_q_many(const PcManyRec *__restrict__ R)
{
    // (pointers left generic here: with them made global -- pc_many_state -- the compiler fused other multiply-adds than in the one-run
    //  kernel, the same statements, and a run in step was no longer bit for bit the run alone)
    const PcState S = R[blockIdx.z].S;
    const unsigned batch = (unsigned)R[blockIdx.z].ia[0];
    {
#pragma clang fp contract(on)
#include "pc_nhats_q_body.inc"
    }
}

// ------------------------------------------------------------------------------------------
// First half of K0 for 64 < nDims <= 128, one grade (split launch): the orthonormal bases (random_utils.F90:381-437),
// Gram-Schmidt in PANELS of sixteen vectors with the trailing update on the fp64 matrix cores.
//
// Eight waves per basis, wave g owns vectors 16 g .. 16 g + 15 in the pair layout of k_whiten (which reads them back as
// its B operand).  Panel p: wave p orthogonalises its sixteen vectors among themselves -- the reference's loop, pivot by
// pivot, the pivot travelling through LDS inside ONE wave (no block barrier, the other waves are parked) -- and leaves
// them, normalised, in LDS; then every later wave projects its sixteen vectors on the whole panel at once:
//     C = Q V^T   (16 x 16, contraction over the coordinates),    V <- V - C^T Q,
// two products whose operands are the registers the vectors live in (V is the B operand of the first and the accumulator
// of the second; C comes out of the first in the layout the second wants it in) and rows of the panel read from LDS.
// That is the reference's arithmetic inside a panel and block classical Gram-Schmidt across panels: the coefficients
// of a panel's pivots are taken from the vector as it was BEFORE the panel, not after each pivot -- the same numbers up to
// rounding of the order of the basis' condition number times epsilon (what k_nhats_big does beyond 128 dimensions).
// One barrier per panel instead of one per pivot, 56 matrix instructions per wave and panel instead of sixteen rounds of
// 32 LDS loads + 96 FMAs: the one-pivot-at-a-time kernel spent 110 us per basis here.
// sum over the four lanes 16 apart (the four coordinate classes of a vector in the pair layout), every lane ends with the
// total: gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap) instead of two trips through the LDS crossbar
__device__ __forceinline__ double lk_sum4(double x)
{
    {
        const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
        x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
        x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    return x;
}
// one pivot of the in-panel Gram-Schmidt (random_utils.F90:391-399): the pivot is vector J of the wave, i.e. lane J of every
// row of sixteen lanes -- its coordinates reach the other lanes of the row by DPP row broadcast, no LDS round trip
template <int J, int NM>
__device__ __forceinline__ void gs_pivot(double (&v)[NM], int li)
{
    double q[NM];
#pragma unroll
    for (int n = 0; n < NM; ++n)
        q[n] = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v[n]), 0x150 + J, 0xF, 0xF, false),
                                __builtin_amdgcn_update_dpp(0, __double2loint(v[n]), 0x150 + J, 0xF, 0xF, false));
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
    for (int n = 0; n < NM; n += 4) {
        a0 += q[n] * q[n]; a1 += q[n + 1] * q[n + 1]; a2 += q[n + 2] * q[n + 2]; a3 += q[n + 3] * q[n + 3];
        c0 += q[n] * v[n]; c1 += q[n + 1] * v[n + 1]; c2 += q[n + 2] * v[n + 2]; c3 += q[n + 3] * v[n + 3];
    }
    const double qq = lk_sum4((a0 + a1) + (a2 + a3)), dv = lk_sum4((c0 + c1) + (c2 + c3));
    // 1 / qq by Newton from the hardware estimate (two steps: full precision)
    double rq = __builtin_amdgcn_rcp(qq);
    rq = fma(rq, fma(-qq, rq, 1.0), rq);
    rq = fma(rq, fma(-qq, rq, 1.0), rq);
    const double cproj = (li > J) ? dv * rq : 0.0;                      // (the pivot itself and the vectors before it stay)
#pragma unroll
    for (int n = 0; n < NM; ++n) v[n] -= cproj * q[n];
}

template <int NT>
__global__ __launch_bounds__(512) void k_basis(PcState S, unsigned batch)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int NM = 4 * NT, NS = 130, RAWB = 32 * 512;
    __shared__ __attribute__((aligned(16))) double Qp[2][16 * NS];
    __builtin_amdgcn_s_setprio(3);                 // (the side stream's kernel is the one the run waits for: section 5d)
    const int D = S.D;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int chain = blockIdx.y, basis = blockIdx.x;
    const int ivec = 16 * g + li, np = (D + 15) >> 4;
    const bool vact = ivec < D;
    // gaussian deviates: element (basis D + i) D + d of stream (batch, chain) in PC_DOM_NHAT, two per Philox call
#ifdef BASIS_DBG
    const long long t_start = clock64();
#endif
    double v[NM];
#pragma unroll
    for (int n = 0; n < NM; ++n) v[n] = 0.0;
    {
        // AS241 in two halves (pc_dev.h): the central branch inline; the arguments that fall in a tail (15 %) wait in an LDS
        // queue and are finished together, so a wave runs the log / sqrt branch about ten times instead of 4 NT times
        constexpr int TQ = 16;                                           // queue slots a lane (beyond: the whole function inline)
        __shared__ double tailq[TQ * 512];
        unsigned tmask = 0u; int tcount = 0;
        auto deviate = [&](double u, int n) __attribute__((always_inline)) {
            bool t;
            double x = pc_inv_normal_central(u, t);
            if (t) { if (tcount < TQ) { tailq[tcount * 512 + tid] = u; tmask |= 1u << n; tcount++; } else x = pc_inv_normal_cdf(u); }
            return x;
        };
        if (vact) {
            const long long e0 = (long long)pc_sel(S.g_e0, 0) + ((long long)basis * D + ivec) * D;
#pragma unroll
            for (int n = 0; n < NM; n += 2) {
                const int d = 8 * (n >> 1) + 2 * lk;
                if (d < D) {
                    const long long e = e0 + d;
                    double ua, ub;
                    pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, (uint32_t)(e >> 1), ua, ub);
                    if ((e & 1ll) == 0) { v[n] = deviate(ua, n); if (d + 1 < D) v[n + 1] = deviate(ub, n + 1); }
                    else {
                        v[n] = deviate(ub, n);
                        if (d + 1 < D) { pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, (uint32_t)(e >> 1) + 1u, ua, ub); v[n + 1] = deviate(ua, n + 1); }
                    }
                }
            }
        }
        for (int k = 0; __any(k < tcount); ++k)
            if (k < tcount) tailq[k * 512 + tid] = pc_inv_normal_tail(tailq[k * 512 + tid]);
        int c = 0;
#pragma unroll
        for (int n = 0; n < NM; ++n) if ((tmask >> n) & 1u) { v[n] = tailq[c * 512 + tid]; c++; }
    }
    {   // random_direction (random_utils.F90:276-298)
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int n = 0; n < NM; n += 2) { p0 += v[n] * v[n]; p1 += v[n + 1] * v[n + 1]; }
        const double n2 = lk_sum4(p0 + p1), inrm = vact ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
        for (int n = 0; n < NM; ++n) v[n] *= inrm;
    }
    const int arow_l = 8 * (li >> 3) + 2 * (li & 3) + ((li >> 2) & 1);   // row of an output tile that sits in my A-operand slot
    double *rawb = S.nhat_raw + ((size_t)chain * S.nb_total + basis) * (size_t)RAWB + (size_t)g * 64 + lane;
#ifdef BASIS_DBG
    const bool dbg = blockIdx.x == 0 && blockIdx.y == 0 && lane == 0;
    const long long t_rng = clock64();
    if (dbg && g == 0) atomicAdd((unsigned long long *)&S.ctl->gen_cyc[0], (unsigned long long)(t_rng - t_start));
#endif
    for (int p = 0; p < np; ++p) {
        double *Q = Qp[p & 1];
#ifdef BASIS_DBG
        const long long t_a = clock64();
#endif
        if (g == p) {
            // ---- my panel: Gram-Schmidt pivot by pivot (random_utils.F90:391-399), the pivot unnormalised as there
            const int cnt = min(16, D - 16 * p);
#define PC_GS(J) if (J < cnt) gs_pivot<J, NM>(v, li);
            PC_GS(0) PC_GS(1) PC_GS(2) PC_GS(3) PC_GS(4) PC_GS(5) PC_GS(6) PC_GS(7)
            PC_GS(8) PC_GS(9) PC_GS(10) PC_GS(11) PC_GS(12) PC_GS(13) PC_GS(14) PC_GS(15)
#undef PC_GS
            {   // the panel's vectors are final: normalise (the reference does it when a vector becomes the pivot: same vector)
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int n = 0; n < NM; n += 2) { p0 += v[n] * v[n]; p1 += v[n + 1] * v[n + 1]; }
                const double n2 = lk_sum4(p0 + p1), inrm = (li < cnt) ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
                for (int n = 0; n < NM; ++n) v[n] *= inrm;
            }
            // the finished panel, normalised, for the waves behind; and out to HBM
#pragma unroll
            for (int n = 0; n < NM; n += 2) *(v2d *)&Q[li * NS + 4 * n + 2 * lk] = v2d{v[n], v[n + 1]};
#pragma unroll
            for (int n = 0; n < NM; ++n) rawb[(size_t)n * 512] = v[n];
#ifdef BASIS_DBG
            if (dbg) atomicAdd((unsigned long long *)&S.ctl->gen_cyc[1], (unsigned long long)(clock64() - t_a));
#endif
        }
        __syncthreads();
#ifdef BASIS_DBG
        const long long t_b = clock64();
#endif
        if (g > p && 16 * g < D) {
            // ---- a later wave: C = Q V^T, V <- V - C^T Q
            v4d c4 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n < NM; n += 2) {
                const v2d a2 = *(const v2d *)&Q[li * NS + 4 * n + 2 * lk];
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, v[n], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, v[n + 1], c4, 0, 0, 0);
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                v4d a4 = v4d{v[4 * ti], v[4 * ti + 1], v[4 * ti + 2], v[4 * ti + 3]};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Q[(4 * ks + lk) * NS + 16 * ti + arow_l], -c4[ks], a4, 0, 0, 0);
                v[4 * ti] = a4[0]; v[4 * ti + 1] = a4[1]; v[4 * ti + 2] = a4[2]; v[4 * ti + 3] = a4[3];
            }
#ifdef BASIS_DBG
            if (dbg && g == p + 1) atomicAdd((unsigned long long *)&S.ctl->gen_cyc[2], (unsigned long long)(clock64() - t_b));
#endif
        }
    }
#ifdef BASIS_DBG
    if (dbg && g == 0) { atomicAdd((unsigned long long *)&S.ctl->gen_cyc[3], (unsigned long long)(clock64() - t_start)); atomicAdd((unsigned long long *)&S.ctl->nn_walks, 1ull); }
#endif
}

// ------------------------------------------------------------------------------------------
// Second half of K0 for 64 < nDims <= 128, one grade (split launch, see k_nhats_q<32, 1>): seeds, whitening W = L.N and, for
// the correlated Gaussian, the products M.(span o n^) -- everything on the fp64 matrix cores, sixteen vectors per wave.
//
// Register layout ("pair layout"): lane (li = lane & 15, lk = lane >> 4) of the wave that owns vectors 16 g .. 16 g + 15
// holds, of vector 16 g + li, the coordinates  dim(n, lk) = 8 (n >> 1) + 2 lk + (n & 1),  n = 0 .. 4 NT - 1  -- pairs of
// neighbours, so that a row leaves in 16-byte pieces.  That IS the B operand of v_mfma_f64_16x16x4_f64 for contraction
// step n (B[k = lk][j = li]) once the A operand uses the same numbering of the contracted index, and it is also the D
// layout of an output tile whose sixteen rows are numbered  row(ti, i) = 16 ti + 8 (i >> 3) + 2 (i & 3) + ((i >> 2) & 1):
// register r of tile ti of lane lk is row i = lk + 4 r, i.e. dim(4 ti + r, lk).  So the basis read from HBM is the B operand
// of L.N, its normalised result is the B operand of M.s, and both results leave from the registers they were summed in:
// no basis in LDS, no transposes.  LDS holds two tiles of sixteen matrix rows (the one in use, the one arriving).
// A block is four waves = 64 vectors, two blocks per basis; each streams the tiles of L and M itself.
template <int NT>
__global__ __launch_bounds__(256) void k_whiten(PcState S, unsigned batch)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int NM = 4 * NT, NR = 16 * NT, NS = 130, RAWB = 32 * 512;
    __shared__ __attribute__((aligned(16))) double tiles[2][16 * NS];
    __shared__ double spn[128], y0s[128];
    __shared__ int sh[2];
    const int D = S.D, nr = S.nr;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int chain = blockIdx.y, basis = blockIdx.x >> 1, half = blockIdx.x & 1, g = 4 * half + wv;
    const bool wact = 16 * g < D;                                       // my sixteen vectors exist (at least one of them)
    const bool ms = S.nhat_Ms != nullptr;
    if (tid == 0) {
        int sel, slot;
        select_seed(S, batch, chain, sel, slot);
        sh[0] = sel; sh[1] = slot;
        if (blockIdx.x == 0) {
            S.ch_cluster[chain] = sel; S.ch_seed_slot[chain] = slot;
            S.ch_contour[chain] = S.logLp[sel];          // nested_sampling.F90:270
            S.ch_epoch[chain] = S.ctl->admin_epoch;
            if (chain == 0) { S.ctl->i_nursery = gridDim.y; S.ctl->batch_id = batch; }
        }
    }
    // the basis, straight into the operand registers
    double b[NM];
    {
        const double *rawb = S.nhat_raw + ((size_t)chain * S.nb_total + basis) * (size_t)RAWB + (size_t)g * 64 + lane;
#pragma unroll
        for (int n = 0; n < NM; ++n) b[n] = wact ? rawb[(size_t)n * 512] : 0.0;
    }
    __syncthreads();
    const double *Lc = S.chol + (size_t)sh[0] * D * D;
    if (ms && tid < 128) {
        const bool on = tid < D;
        const double lo = (on && S.prior.lo) ? S.prior.lo[tid] : 0.0, hi = (on && S.prior.hi) ? S.prior.hi[tid] : 1.0;
        spn[tid] = on ? hi - lo : 0.0;
        const double c0 = on ? S.live[(size_t)sh[1] * S.nT + tid] : 0.0;
        y0s[tid] = on ? (lo + (hi - lo) * c0) - (S.like.mean ? S.like.mean[tid] : 0.0) : 0.0;
    }
    // tiles travel global -> registers -> LDS one tile ahead of the matrix cores: one barrier per tile
    double pre[8];
    auto load_L = [&](int ti) __attribute__((always_inline)) {
        const int kmax = min(NR, 16 * (ti + 1));                        // L(a, b) = 0 for b > a
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int x = tid + q * 256, r = x / kmax, bcol = x - r * kmax, arow = 16 * ti + r;
            pre[q] = (x < 16 * kmax && arow < D && bcol < D) ? Lc[(size_t)arow * D + bcol] : 0.0;
        }
    };
    auto store_L = [&](int ti) __attribute__((always_inline)) {
        const int kmax = min(NR, 16 * (ti + 1));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int x = tid + q * 256, r = x / kmax, bcol = x - r * kmax;
            if (x < 16 * kmax) tiles[ti & 1][r * NS + bcol] = pre[q];
        }
    };
    const double *Mt = S.like.invcov;                                   // Mt[b * D + a] = M(a, b)
    auto load_M = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int x = tid + q * 256, r = x & 15, bcol = x >> 4, arow = 16 * ti + r;
            pre[q] = (x < 16 * NR && arow < D && bcol < D) ? Mt[(size_t)bcol * D + arow] : 0.0;
        }
    };
    auto store_M = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int x = tid + q * 256, r = x & 15, bcol = x >> 4;
            if (x < 16 * NR) tiles[ti & 1][r * NS + bcol] = pre[q];
        }
    };
    // my row of a tile as A operand (row(ti, li) - 16 ti), and where contraction steps n, n + 1 (n even) sit in it
    const int arow_l = 8 * (li >> 3) + 2 * (li & 3) + ((li >> 2) & 1);
    const int aoff = arow_l * NS + 2 * lk;
    v4d acc[NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[ti] = v4d{0.0, 0.0, 0.0, 0.0};
    load_L(0);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        store_L(ti);
        if (ti + 1 < NT) load_L(ti + 1);
        __syncthreads();
        if (wact) {
            const double *pa = &tiles[ti & 1][aoff];
            v4d a4 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n < 4 * (ti + 1); n += 2) {
                const v2d a2 = *(const v2d *)(pa + 4 * n);              // dims 8 (n >> 1) + 2 lk, + 1
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b[n], a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b[n + 1], a4, 0, 0, 0);
            }
            acc[ti] = a4;
        }
    }
    if (ms) load_M(0);
    // |w| of my vector (chordal_sampling.f90:80-82): my registers, then the other three lane groups
    double n2 = 0.0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) n2 += acc[ti][r] * acc[ti][r];
    n2 += __shfl_xor(n2, 16); n2 += __shfl_xor(n2, 32);
    const double wn = sqrt(n2), iw = 1.0 / wn;
    const int ivec = 16 * g + li;
    const bool wanted = wact && ivec < D && basis * D + ivec < nr;
    const size_t orow = ((size_t)chain * nr + (size_t)basis * D + ivec) * D;
    const bool al16 = (D & 1) == 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) b[4 * ti + r] = acc[ti][r] * iw;   // n^ in pair layout
    auto put_rows = [&](double *base) __attribute__((always_inline)) {
        if (!wanted) return;
        double *out = base + orow;
#pragma unroll
        for (int n = 0; n < NM; n += 2) {
            const int d = 8 * (n >> 1) + 2 * lk;
            if (d + 1 < D && al16) *(v2d *)(out + d) = v2d{b[n], b[n + 1]};
            else { if (d < D) out[d] = b[n]; if (d + 1 < D) out[d + 1] = b[n + 1]; }
        }
    };
    put_rows(S.nhat);
    if (wanted && lk == 0) S.nhat_w[(size_t)chain * nr + (size_t)basis * D + ivec] = wn * 3.0;
    if (!ms) return;
    // ---- correlated Gaussian (random_gaussian.f90:17-30): along a chord the exponent is quadratic (see ChainCtx) and all a
    //      slice needs of the matrix is M.s, s = span o n^; every direction of the chain is known here.  One wave of the
    //      chain's first basis also forms M.(theta_seed - mean), the product the chain needs for its start point.
#pragma unroll
    for (int n = 0; n < NM; ++n) b[n] *= spn[8 * (n >> 1) + 2 * lk + (n & 1)];
    const bool do_y0 = basis == 0 && half == 1 && wv == 3;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[ti] = v4d{0.0, 0.0, 0.0, 0.0};
    __syncthreads();                                                    // the last tile of L has been consumed
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        store_M(ti);
        if (ti + 1 < NT) load_M(ti + 1);
        __syncthreads();
        const double *pa = &tiles[ti & 1][aoff];
        if (wact) {
            v4d a4 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n < NM; n += 2) {
                const v2d a2 = *(const v2d *)(pa + 4 * n);
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b[n], a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b[n + 1], a4, 0, 0, 0);
            }
            acc[ti] = a4;
        }
        if (do_y0) {
            double t = 0.0;
#pragma unroll
            for (int n = 0; n < NM; n += 2) {
                const v2d a2 = *(const v2d *)(pa + 4 * n);
                const int d = 8 * (n >> 1) + 2 * lk;
                t += a2.x * y0s[d] + a2.y * y0s[d + 1];
            }
            t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
            const int a = 16 * ti + arow_l;
            if (lk == 0 && a < D) S.ch_My[(size_t)chain * D + a] = t;
        }
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) b[4 * ti + r] = acc[ti][r];
    put_rows(S.nhat_Ms);
}

// ------------------------------------------------------------------------------------------
// K1: one slice-sampling chain per wavefront
// ------------------------------------------------------------------------------------------
template <int DPL, int NROWS>
struct ChainCtx {
    const PcState &S;
    const LaneDims<DPL> &ld;
    int lane;
    double *ybuf;
    int nlike;
    // Quadratic-form likelihoods (gaussian.f90, random_gaussian.f90) under a uniform prior: along a chord
    // theta(t) = theta0 + t s the exponent is  q(t) = qa + 2 qb t + qc t^2  with
    //   qa = y.M.y,  qb = s.M.y,  qc = s.M.s   (y = theta0 - mu; M = 1/sigma^2 or the inverse covariance),
    // all three reduced once per slice.  A trial is then a handful of scalar operations instead of a
    // wave reduction (or a D x D matrix-vector product) per likelihood call.  Every trial still counts as
    // one evaluation (calculate.f90:44).
    bool quad;
    double qa, qb, qc, qnorm;
};

// calculate_point (calculate.f90:6-50) at x0 + t*nh; leaves cube/theta of the trial in registers
template <int DPL, int NROWS, int KIND = -1>
__device__ __forceinline__ double eval_at(ChainCtx<DPL, NROWS> &C, const double (&x0)[DPL], const double (&nh)[DPL],
                                          double t, double (&cube)[DPL], double (&th)[DPL])
{
    bool outside = false;
    if (C.quad) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            cube[k] = x0[k] + t * nh[k];
            if (C.ld.on[k]) outside |= (cube[k] < 0.0) | (cube[k] > 1.0);
            th[k] = C.ld.lo[k] + C.ld.span[k] * cube[k];
        }
        double lg = C.qnorm - (C.qa + t * (2.0 * C.qb + t * C.qc)) / 2.0;
        if (__ballot(outside) != 0ull) {
#pragma unroll
            for (int k = 0; k < DPL; ++k) th[k] = 0.0;
            lg = C.S.logzero;                   // calculate.f90:36-38
        } else if (lg > C.S.logzero) C.nlike++;
        return lg;
    }
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        cube[k] = x0[k] + t * nh[k];
        if (C.ld.on[k]) outside |= (cube[k] < 0.0) | (cube[k] > 1.0);
    }
    if (__ballot(outside) != 0ull) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) th[k] = 0.0;
        return C.S.logzero;
    }
#pragma unroll
    for (int k = 0; k < DPL; ++k) th[k] = C.ld.lo[k] + C.ld.span[k] * cube[k];
    const double logL = like_eval<DPL, NROWS, KIND>(C.S, th, C.ld, C.lane, C.ybuf);
    if (logL > C.S.logzero) C.nlike++;
    return logL;
}

// Two independent trial points at once (the two ends of the initial bracket): the per-point work is a
// dependent chain (FMA -> compare -> reduction), so the second evaluation rides in the shadow of the first.
// Straight-line code: both likelihoods are computed unconditionally and masked afterwards.
template <int DPL, int NROWS, int KIND = -1>
__device__ __forceinline__ void eval_pair(ChainCtx<DPL, NROWS> &C, const double (&x0)[DPL], const double (&nh)[DPL],
                                          double tA, double tB, double &lA, double &lB)
{
    const PcLike &L = C.S.like;
    const int kind = KIND >= 0 ? KIND : L.kind;
    if (C.quad) {
        bool oA = false, oB = false;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double cA = x0[k] + tA * nh[k], cB = x0[k] + tB * nh[k];
            if (C.ld.on[k]) { oA |= (cA < 0.0) | (cA > 1.0); oB |= (cB < 0.0) | (cB > 1.0); }
        }
        lA = C.qnorm - (C.qa + tA * (2.0 * C.qb + tA * C.qc)) / 2.0;
        lB = C.qnorm - (C.qa + tB * (2.0 * C.qb + tB * C.qc)) / 2.0;
        if (__ballot(oA) != 0ull) lA = C.S.logzero; else if (lA > C.S.logzero) C.nlike++;     // calculate.f90:36-38
        if (__ballot(oB) != 0ull) lB = C.S.logzero; else if (lB > C.S.logzero) C.nlike++;
        return;
    }
    if (kind != PC_LIKE_RASTRIGIN && kind != PC_LIKE_TWIN_GAUSSIAN) {
        // any other functor (the quadratic built-ins when settings.ablate bit 0 takes their closed form away): the two points through
        // like_eval, one call behind the other in the source, their reductions side by side in the schedule
        // (until round 6 these kinds fell through to the twin-Gaussian sums below: an initial bracket judged by another likelihood never
        //  stepped out, and the "general functor" figures were those of a run with 3.3 evaluations a slice instead of 4.5)
        bool outA = false, outB = false;
        double thA[DPL], thB[DPL];
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double cA = x0[k] + tA * nh[k], cB = x0[k] + tB * nh[k];
            if (C.ld.on[k]) { outA |= (cA < 0.0) | (cA > 1.0); outB |= (cB < 0.0) | (cB > 1.0); }
            thA[k] = C.ld.lo[k] + C.ld.span[k] * cA; thB[k] = C.ld.lo[k] + C.ld.span[k] * cB;
        }
        const bool oa = __ballot(outA) != 0ull, ob = __ballot(outB) != 0ull;
        lA = like_eval<DPL, NROWS, KIND>(C.S, thA, C.ld, C.lane, C.ybuf);
        lB = like_eval<DPL, NROWS, KIND>(C.S, thB, C.ld, C.lane, C.ybuf);
        if (oa) lA = C.S.logzero; else if (lA > C.S.logzero) C.nlike++;     // calculate.f90:36-38
        if (ob) lB = C.S.logzero; else if (lB > C.S.logzero) C.nlike++;
        return;
    }
    bool outA = false, outB = false;
    double sA = 0.0, sB = 0.0, s2A = 0.0, s2B = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        const double cA = x0[k] + tA * nh[k], cB = x0[k] + tB * nh[k];
        if (C.ld.on[k]) { outA |= (cA < 0.0) | (cA > 1.0); outB |= (cB < 0.0) | (cB > 1.0); }
        const double thA = C.ld.lo[k] + C.ld.span[k] * cA, thB = C.ld.lo[k] + C.ld.span[k] * cB;
        if (C.ld.on[k]) {
            if (kind == PC_LIKE_RASTRIGIN) {
                sA += 8.515435146961291 + thA * thA - 10.0 * cos(PC_TWO_PI * thA);
                sB += 8.515435146961291 + thB * thB - 10.0 * cos(PC_TWO_PI * thB);
            } else {
                const int dim = C.lane + 64 * k;
                const double m1 = dim < 2 ? -0.5 : 0.0, m2 = dim < 2 ? 0.5 : 0.0;
                const double a1 = (thA - m1) * L.inv_sigma, a2 = (thA - m2) * L.inv_sigma;
                const double b1 = (thB - m1) * L.inv_sigma, b2 = (thB - m2) * L.inv_sigma;
                sA += a1 * a1; s2A += a2 * a2; sB += b1 * b1; s2B += b2 * b2;
            }
        }
    }
    const bool oa = __ballot(outA) != 0ull, ob = __ballot(outB) != 0ull;
    sA = wsum<DPL, NROWS>(sA); sB = wsum<DPL, NROWS>(sB);
    if (kind == PC_LIKE_RASTRIGIN) { lA = -sA; lB = -sB; }
    else {
        s2A = wsum<DPL, NROWS>(s2A); s2B = wsum<DPL, NROWS>(s2B);
        lA = pc_logaddexp(L.norm - sA / 2.0, L.norm - s2A / 2.0) - 0.6931471805599453;
        lB = pc_logaddexp(L.norm - sB / 2.0, L.norm - s2B / 2.0) - 0.6931471805599453;
    }
    if (oa) lA = C.S.logzero; else if (lA > C.S.logzero) C.nlike++;     // calculate.f90:36-38
    if (ob) lB = C.S.logzero; else if (lB > C.S.logzero) C.nlike++;
}

// SPECIAL = false is the production kernel.  SPECIAL = true adds the two rare modes, both decided at run time:
// more than one parameter grade (the evaluations of a slice are booked to the grade of its direction) and the
// sequential-stream test mode (every draw taken from ONE running stream in the reference's program order).
// WPB = chains (wavefronts) per workgroup.  One, except for the correlated Gaussian with its inverse covariance in LDS:
// an 80 KB matrix per chain left one wave per CU; WPB chains share one copy (every barrier below is executed the same
// number of times by every chain: per slice, never per likelihood evaluation).
// FW > 0 (production path for nDims <= 24, one grade): the kernel also does what the second half of k_nhats did -- seed
// choice (GenerateSeed) and whitening of the orthonormal directions with the seed cluster's Cholesky factor -- so that
// the directions never travel through HBM: a prologue whitens all of the chain's directions at once, lane = direction
// (the loops of k_nhats, bit for bit: row sums in ascending b, the norm on four partial sums), into LDS, from where the
// slices pick them up in deck order.  FW = unroll width >= nDims.
// PC_SLICE_WAVES (build-time experiment): cap the registers so that this many waves share a SIMD
#ifdef PC_SLICE_WAVES
#define PC_SLICE_ATTR __attribute__((amdgpu_waves_per_eu(PC_SLICE_WAVES, PC_SLICE_WAVES)))
#else
#define PC_SLICE_ATTR
#endif
template <int DPL, int NROWS, bool SPECIAL, int WPB = 1, int FW = 0, int LEAN = 0>
__global__ PC_SLICE_ATTR __launch_bounds__(64 * WPB * ((FW > 0 && (LEAN == 1 || LEAN == 3 || LEAN == 5) && WPB == 4) ? 2 : 1)) void k_slice(PcState S, unsigned batch, int phi_lds, int mat_lds)
{
#include "pc_slice_body.inc"
}

// several runs of a device in step (Cohort, pc_engine.hip): the same statements on each run's own state, blockIdx.y = run.  (Included, not
// called: handing the state to a function by reference moved fused multiply-adds in the one-run kernel -- -ffp-contract=fast works on
// whatever the optimiser has made of the code -- and the lane-per-chain kernel of pc_slice_t.hip is matched to that kernel's ISA.)
template <int DPL, int NROWS, bool SPECIAL, int WPB = 1, int FW = 0, int LEAN = 0>
__global__ PC_SLICE_ATTR __launch_bounds__(64 * WPB) void k_slice_many(const PcManyRec *__restrict__ R, int phi_lds, int mat_lds)
{
    // (pointers left generic here: with them made global -- pc_many_state -- the compiler fused other multiply-adds than in the one-run
    //  kernel, the same statements, and a run in step was no longer bit for bit the run alone)
    const PcState S = R[blockIdx.y].S;
    const unsigned batch = (unsigned)R[blockIdx.y].ia[0];
#include "pc_slice_body.inc"
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
extern "C" int pc_launch_generate_live(const PcState *S, int attempt0, int n, double *rows, double *rows_logL,
                                       hipStream_t st)
{
    const size_t sh = sizeof(double) * S->D;
    if (S->D <= 64) hipLaunchKernelGGL((k_generate_live<1>), dim3(n), dim3(64), sh, st, *S, attempt0, rows, rows_logL);
    else if (S->D <= 128) hipLaunchKernelGGL((k_generate_live<2>), dim3(n), dim3(64), sh, st, *S, attempt0, rows, rows_logL);
    else if (S->D <= 256) hipLaunchKernelGGL((k_generate_live<4>), dim3(n), dim3(64), sh, st, *S, attempt0, rows, rows_logL);
    else return 1;
    return 0;
}

// the split launch (see k_nhats): part 1 = bases, part 2 = seeds + whitening; returns 1 where only the whole kernel exists
extern "C" int pc_nhats_splittable(const PcState *S)
{
    const char *e = std::getenv("PC_NHATS_QUAD_MIN");
    if (S->D > 64 && S->D <= 128) return S->ngrade <= 1 && !S->seq_mode && S->nhat_raw != nullptr;    // k_nhats_q<32, 1 / 2>
    if (S->D > 24 && S->D <= 64) return S->ngrade <= 1 && !S->seq_mode && S->nhat_raw != nullptr && (!e || std::atoi(e) <= 25);      // k_nhats_q<8 / 16, 1 / 2> (round 5)
    return S->D <= 24 && S->D < (e ? std::atoi(e) : 25) && !S->seq_mode && S->nhat_raw != nullptr;
}
// (two tile buffers up to nDims 112; beyond that one, with a barrier more per tile: 160 KB of LDS)
static size_t pc_nhats_q32_lds(int D) { const int NR = ((D + 15) / 16) * 16; return sizeof(double) * (2 * 4 * 34 + (size_t)(NR + (NR <= 112 ? 32 : 16)) * 129); }
extern "C" int pc_launch_bases_t(const PcState *S, unsigned batch, int nchains, hipStream_t st);
extern "C" int pc_launch_nhats_part(const PcState *S, unsigned batch, int nchains, int part, hipStream_t st, int packed)
{
    if (!pc_nhats_splittable(S)) return 1;
    const int D = S->D;
    dim3 grid(S->nb_total, nchains);
    if (D > 64) {
        if (part == 1) {
            static const bool panel_off = std::getenv("PC_BASIS_PANEL_OFF") != nullptr;
            const int nt1 = (D + 15) / 16;
            if (panel_off) hipLaunchKernelGGL((k_nhats_q<32, 1>), grid, dim3(512), sizeof(double) * 2 * 4 * 34, st, *S, batch);
            else if (nt1 <= 5) hipLaunchKernelGGL((k_basis<5>), grid, dim3(512), 0, st, *S, batch);
            else if (nt1 == 6) hipLaunchKernelGGL((k_basis<6>), grid, dim3(512), 0, st, *S, batch);
            else if (nt1 == 7) hipLaunchKernelGGL((k_basis<7>), grid, dim3(512), 0, st, *S, batch);
            else hipLaunchKernelGGL((k_basis<8>), grid, dim3(512), 0, st, *S, batch);
        }
        else {
            dim3 g2(S->nb_total * 2, nchains);
            const int nt = (D + 15) / 16;
            if (nt <= 5) hipLaunchKernelGGL((k_whiten<5>), g2, dim3(256), 0, st, *S, batch);
            else if (nt == 6) hipLaunchKernelGGL((k_whiten<6>), g2, dim3(256), 0, st, *S, batch);
            else if (nt == 7) hipLaunchKernelGGL((k_whiten<7>), g2, dim3(256), 0, st, *S, batch);
            else hipLaunchKernelGGL((k_whiten<8>), g2, dim3(256), 0, st, *S, batch);
        }
        return 0;
    }
    if (D > 24) {
        // nDims 25 ... 64: the four-threads-per-vector kernel in two halves -- deviates + Gram-Schmidt (nothing the contraction changes:
        // drawn ahead on the side stream), then seeds + whitening in front of k_slice
        auto lds_q = [](int HV) { return sizeof(double) * (size_t)(2 + HV) * 4 * (HV + 2); };
        if (D <= 32) { if (part == 1) hipLaunchKernelGGL((k_nhats_q<8, 1>), grid, dim3(128), lds_q(8), st, *S, batch); else hipLaunchKernelGGL((k_nhats_q<8, 2>), grid, dim3(128), lds_q(8), st, *S, batch); }
        else { if (part == 1) hipLaunchKernelGGL((k_nhats_q<16, 1>), grid, dim3(256), lds_q(16), st, *S, batch); else hipLaunchKernelGGL((k_nhats_q<16, 2>), grid, dim3(256), lds_q(16), st, *S, batch); }
        return 0;
    }
    const size_t sh = sizeof(double) * ((size_t)(D + 8) * (D + 8) + 2 * 128) + 16;
    // several runs on the device (packed != 0): the same bases, lane = basis (pc_slice_t.hip)
    if (part == 1 && packed && pc_launch_bases_t(S, batch, nchains, st) == 0) return 0;
    if (part == 1) {
        if (D <= 8) hipLaunchKernelGGL((k_nhats<8, 64, 1>), grid, dim3(64), sh, st, *S, batch);
        else if (D <= 16) hipLaunchKernelGGL((k_nhats<16, 64, 1>), grid, dim3(64), sh, st, *S, batch);
        else hipLaunchKernelGGL((k_nhats<24, 64, 1>), grid, dim3(64), sh, st, *S, batch);
    } else {
        if (D <= 8) hipLaunchKernelGGL((k_nhats<8, 64, 2>), grid, dim3(64), sh, st, *S, batch);
        else if (D <= 16) hipLaunchKernelGGL((k_nhats<16, 64, 2>), grid, dim3(64), sh, st, *S, batch);
        else hipLaunchKernelGGL((k_nhats<24, 64, 2>), grid, dim3(64), sh, st, *S, batch);
    }
    return 0;
}

extern "C" int pc_launch_nhats(const PcState *S, unsigned batch, int nchains, hipStream_t st)
{
    const int D = S->D, nb = S->nb_total;
    dim3 grid(nb, nchains);
    static int quad_min = -1;                       // smallest nDims that takes the four-threads-per-vector kernel
    if (quad_min < 0) { const char *e = std::getenv("PC_NHATS_QUAD_MIN"); quad_min = e ? std::atoi(e) : 25; }   // measured: 20-D 52 vs 39 us (old kernel better), 28-D 43 vs 47, 40-D 95 vs 133, 64-D 129 vs 240
    if (D >= quad_min) {
        // dynamic LDS: pivot buffer + Cholesky tile (HV rows of 4 (HV + 2) doubles), or basis + sixteen rows of L (HV = 32)
        auto lds_q = [](int HV) { return sizeof(double) * (size_t)(2 + HV) * 4 * (HV + 2); };
        if (D <= 32) hipLaunchKernelGGL((k_nhats_q<8>), grid, dim3(128), lds_q(8), st, *S, batch);
        else if (D <= 64) hipLaunchKernelGGL((k_nhats_q<16>), grid, dim3(256), lds_q(16), st, *S, batch);
        else if (D <= 128) {
            const size_t shq = pc_nhats_q32_lds(D);
            pc_need_dyn_lds((const void *)k_nhats_q<32>, shq);
            hipLaunchKernelGGL((k_nhats_q<32>), grid, dim3(512), shq, st, *S, batch);
        }
        else if (D <= 256 && S->nhat_raw) hipLaunchKernelGGL(k_nhats_big, grid, dim3(PC_BIG_NT), 0, st, *S, batch);
        else return 1;
        return 0;
    }
    // deviates / Cholesky factor with padded rows, then the double-buffered pivot (2 x DMAX <= 2 x 128)
    const size_t sh = sizeof(double) * ((size_t)(D + 8) * (D + 8) + 2 * 128) + 16;
    if (D <= 8) hipLaunchKernelGGL((k_nhats<8, 64>), grid, dim3(64), sh, st, *S, batch);
    else if (D <= 16) hipLaunchKernelGGL((k_nhats<16, 64>), grid, dim3(64), sh, st, *S, batch);
    else if (D <= 24) hipLaunchKernelGGL((k_nhats<24, 64>), grid, dim3(64), sh, st, *S, batch);
    else if (D <= 32) hipLaunchKernelGGL((k_nhats<32, 64>), grid, dim3(64), sh, st, *S, batch);
    else if (D <= 64) hipLaunchKernelGGL((k_nhats<64, 64>), grid, dim3(64), sh, st, *S, batch);
    else return 1;
    return 0;
}

extern "C" int pc_slice_fusable(const PcState *S)
{   // the slice kernel can do seeds + whitening itself: raw bases in HBM (split launch), one grade, nDims <= 24
    static const bool off = std::getenv("PC_SLICE_FUSED_OFF") != nullptr;
    return !off && S->D <= 24 && pc_nhats_splittable(S) && S->ngrade <= 1 && S->like.kind != PC_LIKE_CORR_GAUSSIAN && S->nr <= 1024;
}

// the functor variants of k_slice (LEAN = 3 Rastrigin, 4 twin Gaussian, 5 the Gaussian when settings.ablate bit 0 asks for it as a functor): one grade, keyed draws
static int slice_lean_functor(const PcState *S)
{
    static const bool off = std::getenv("PC_SLICE_LEAN_OFF") != nullptr;
    if (off || S->ngrade > 1 || S->seq_mode) return 0;
    return S->like.kind == PC_LIKE_RASTRIGIN ? 3 : (S->like.kind == PC_LIKE_TWIN_GAUSSIAN ? 4 : ((S->like.kind == PC_LIKE_GAUSSIAN && (S->ablate & 1)) ? 5 : 0));
}

extern "C" int pc_launch_slice_fused(const PcState *S, unsigned batch, int nchains, hipStream_t st)
{
    if (!pc_slice_fusable(S)) return 1;
    const size_t sh0 = sizeof(double) * ((size_t)S->D + S->nr) + 16;
    const size_t tb = sizeof(double) * (size_t)S->nr * (S->D + 1);
    const int phi_lds = (S->nDer > 0 && sh0 + tb <= 48 * 1024) ? 1 : 0;
    const int D = S->D, FWv = D <= 8 ? 8 : (D <= 16 ? 16 : 24);
    const size_t sh = sh0 + (phi_lds ? tb : 0) + sizeof(double) * ((size_t)FWv * D + (size_t)S->nr * (D + 2));   // + L, directions, widths
    if (sh > 150 * 1024) return 1;
    static const bool lean_off = std::getenv("PC_SLICE_LEAN_OFF") != nullptr;
    const bool lean = !lean_off && S->like.kind == PC_LIKE_GAUSSIAN && !(S->ablate & 1) && phi_lds && S->nr <= 64 && !S->seq_mode && S->ngrade <= 1;
    const int leanf = slice_lean_functor(S);
    // (a helper wavefront per chain for the lean variants whose deck lives in registers: pc_slice_body.inc; settings.ablate bit 13 / PC_SLICE_HELPER_OFF: without)
    // four chains a workgroup with their four helper wavefronts (pc_slice_body.inc): the lean variants whose deck lives in registers, nurseries of a
    // multiple of four chains; settings.ablate bit 13 / PC_SLICE_HELPER_OFF: one wavefront a workgroup as before (the same numbers)
    static const bool helper_off = std::getenv("PC_SLICE_HELPER_OFF") != nullptr;
    const size_t pw4 = ((size_t)D + S->nr + ((phi_lds || lean) ? (size_t)S->nr * (D + 1) : 0) + (size_t)FWv * D + (size_t)S->nr * (D + 2) + (size_t)((S->nr + 3) / 4) * 128 + (size_t)S->nr + 1) & ~(size_t)1;
    const size_t sh4 = 4 * sizeof(double) * pw4 + 16;
    const bool help = !helper_off && !(S->ablate & 8192) && S->nr <= 64 && (nchains & 3) == 0 && sh4 <= 150 * 1024 && !S->spec_guard;
#define PC_SLICE_FUSED_L(NROWS, FW, LN) { \
        if ((LN == 3 || LN == 5) && help) { \
        if (sh4 > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<1, NROWS, false, 4, FW, LN>, sh4); \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 4, FW, LN>), dim3(nchains / 4), dim3(512), sh4, st, *S, batch, phi_lds, 0); } else { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<1, NROWS, false, 1, FW, LN>, sh); \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 1, FW, LN>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, 0); } }
#define PC_SLICE_FUSED(NROWS, FW) { \
        if (leanf == 3) PC_SLICE_FUSED_L(NROWS, FW, 3) else if (leanf == 4) PC_SLICE_FUSED_L(NROWS, FW, 4) else if (leanf == 5) PC_SLICE_FUSED_L(NROWS, FW, 5) else \
        if (lean) { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<1, NROWS, false, 1, FW, 1>, sh); \
        if (help) { if (sh4 > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<1, NROWS, false, 4, FW, 1>, sh4); \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 4, FW, 1>), dim3(nchains / 4), dim3(512), sh4, st, *S, batch, phi_lds, 0); } else \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 1, FW, 1>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, 0); } else { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<1, NROWS, false, 1, FW>, sh); \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 1, FW>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, 0); } }
    if (D <= 8) PC_SLICE_FUSED(1, 8)
    else if (D <= 16) PC_SLICE_FUSED(1, 16)
    else PC_SLICE_FUSED(2, 24)
#undef PC_SLICE_FUSED
#undef PC_SLICE_FUSED_L
    return 0;
}

// Several runs of a device in step: the sampling kernel of any device likelihood launched once for all of them (grid.y = run; every
// run the same shape: nDims, num_repeats, chains, likelihood kind).  fused: k_slice with the seed choice and the whitening inside
// (pc_launch_slice_fused); else the plain kernel behind k_nhats*.  1: a shape only the one-run launchers take.
extern "C" int pc_launch_slice_many(const PcState *S, const PcManyRec *dR, int R, int nchains, int fused, hipStream_t st)
{
    const int D = S->D;
    const size_t sh0 = sizeof(double) * ((size_t)D + S->nr) + 16;
    const size_t tb = sizeof(double) * (size_t)S->nr * (D + 1);
    const int phi_lds = (S->nDer > 0 && sh0 + tb <= 48 * 1024) ? 1 : 0;
    if (fused) {
        if (!pc_slice_fusable(S)) return 1;
        const int FWv = D <= 8 ? 8 : (D <= 16 ? 16 : 24);
        const size_t sh = sh0 + (phi_lds ? tb : 0) + sizeof(double) * ((size_t)FWv * D + (size_t)S->nr * (D + 2));
        if (sh > 150 * 1024) return 1;
        const int leanf = slice_lean_functor(S) == 5 ? 0 : slice_lean_functor(S);
#define PC_SLICE_FUSED_ML(NROWS, FW, LN) { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice_many<1, NROWS, false, 1, FW, LN>, sh); \
        hipLaunchKernelGGL((k_slice_many<1, NROWS, false, 1, FW, LN>), dim3(nchains, R), dim3(64), sh, st, dR, phi_lds, 0); }
#define PC_SLICE_FUSED_M(NROWS, FW) { if (leanf == 3) PC_SLICE_FUSED_ML(NROWS, FW, 3) else if (leanf == 4) PC_SLICE_FUSED_ML(NROWS, FW, 4) else PC_SLICE_FUSED_ML(NROWS, FW, 0) }
        if (D <= 8) PC_SLICE_FUSED_M(1, 8)
        else if (D <= 16) PC_SLICE_FUSED_M(1, 16)
        else PC_SLICE_FUSED_M(2, 24)
#undef PC_SLICE_FUSED_M
#undef PC_SLICE_FUSED_ML
        return 0;
    }
    if (S->like.kind == PC_LIKE_CORR_GAUSSIAN || S->ngrade > 1 || S->seq_mode || D > 64) return 1;
    const size_t sh = sh0 + (phi_lds ? tb : 0);
    const int leanf = slice_lean_functor(S) == 5 ? 0 : slice_lean_functor(S);
#define PC_SLICE_ML(DPL, NROWS, LN) { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice_many<DPL, NROWS, false, 1, 0, LN>, sh); \
        hipLaunchKernelGGL((k_slice_many<DPL, NROWS, false, 1, 0, LN>), dim3(nchains, R), dim3(64), sh, st, dR, phi_lds, 0); }
#define PC_SLICE_M(DPL, NROWS) { if (leanf == 3) PC_SLICE_ML(DPL, NROWS, 3) else if (leanf == 4) PC_SLICE_ML(DPL, NROWS, 4) else PC_SLICE_ML(DPL, NROWS, 0) }
    if (D <= 16) PC_SLICE_M(1, 1)
    else if (D <= 32) PC_SLICE_M(1, 2)
    else PC_SLICE_M(1, 4)
#undef PC_SLICE_M
#undef PC_SLICE_ML
    return 0;
}

// ... and the directions (bases, seeds, whitening in one kernel: pc_launch_nhats) for the shapes that do not split: 24 < nDims <= 64
extern "C" int pc_launch_nhats_many(const PcState *S, const PcManyRec *dR, int R, int nchains, hipStream_t st)
{
    const int D = S->D, nb = S->nb_total;
    if (D < 25 || D > 64 || S->seq_mode || std::getenv("PC_NHATS_QUAD_MIN")) return 1;
    dim3 grid(nb, nchains, R);
    auto lds_q = [](int HV) { return sizeof(double) * (size_t)(2 + HV) * 4 * (HV + 2); };
    if (D <= 32) hipLaunchKernelGGL((k_nhats_q_many<8>), grid, dim3(128), lds_q(8), st, dR);
    else hipLaunchKernelGGL((k_nhats_q_many<16>), grid, dim3(256), lds_q(16), st, dR);
    return 0;
}

extern "C" int pc_launch_slice(const PcState *S, unsigned batch, int nchains, hipStream_t st)
{
    // theta of every baby stays in LDS (derived parameters at the end of the chain) when it fits; so does the
    // inverse covariance of the correlated Gaussian
    const size_t sh0 = sizeof(double) * ((size_t)S->D + S->nr) + 16;     // ybuf + two int decks
    const size_t tb = sizeof(double) * (size_t)S->nr * (S->D + 1);
    const int phi_lds = (S->nDer > 0 && sh0 + tb <= 48 * 1024) ? 1 : 0;
    size_t sh = sh0 + (phi_lds ? tb : 0);
    const size_t mb = sizeof(double) * (size_t)S->D * S->D;
    const int mat_lds = (S->like.kind == PC_LIKE_CORR_GAUSSIAN && S->nhat_Ms == nullptr && sh + mb <= 150 * 1024) ? 1 : 0;
    const int D = S->D;
    // four chains per workgroup around one LDS copy of the inverse covariance (65 <= nDims <= 128)
    static const bool wpb_off = std::getenv("PC_SLICE_WPB_OFF") != nullptr;
    if (mat_lds && D > 64 && D <= 128 && nchains % 4 == 0 && S->ngrade <= 1 && !S->seq_mode && !wpb_off && 4 * sh + mb <= 150 * 1024) {
        const size_t sh4 = 4 * sh + mb;
        if (sh4 > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<2, 4, false, 4>, sh4);
        hipLaunchKernelGGL((k_slice<2, 4, false, 4>), dim3(nchains / 4), dim3(256), sh4, st, *S, batch, phi_lds, mat_lds);
        return 0;
    }
    if (mat_lds) sh += mb;
    static const bool lean2_off = std::getenv("PC_SLICE_LEAN_OFF") != nullptr;
    if (!lean2_off && S->like.kind == PC_LIKE_CORR_GAUSSIAN && S->nhat_Ms != nullptr && !(S->ablate & 1) && S->nDer == 0 && S->nr > 64 && D > 64 && D <= 128 &&
        S->ngrade <= 1 && !S->seq_mode && !mat_lds) {
        // BASELINE configs[4]'s shape: the kernel without its other variants (LEAN = 2)
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<2, 4, false, 1, 0, 2>, sh);
        hipLaunchKernelGGL((k_slice<2, 4, false, 1, 0, 2>), dim3(nchains), dim3(64), sh, st, *S, batch, 0, 0);
        return 0;
    }
#define PC_SLICE_LAUNCH1(DPL, NROWS, GR) { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<DPL, NROWS, GR>, sh); \
        hipLaunchKernelGGL((k_slice<DPL, NROWS, GR>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, mat_lds); }
    const int leanf = (D <= 64 && !mat_lds) ? slice_lean_functor(S) : 0;
#define PC_SLICE_LAUNCHL(DPL, NROWS, LN) { \
        if (sh > 48 * 1024) pc_need_dyn_lds((const void *)k_slice<DPL, NROWS, false, 1, 0, LN>, sh); \
        hipLaunchKernelGGL((k_slice<DPL, NROWS, false, 1, 0, LN>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, mat_lds); }
#define PC_SLICE_LAUNCH(DPL, NROWS) { if (DPL == 1 && leanf == 3) PC_SLICE_LAUNCHL(1, NROWS, 3) else if (DPL == 1 && leanf == 4) PC_SLICE_LAUNCHL(1, NROWS, 4) else if (DPL == 1 && leanf == 5) PC_SLICE_LAUNCHL(1, NROWS, 5) else \
        if (S->ngrade > 1 || S->seq_mode) PC_SLICE_LAUNCH1(DPL, NROWS, true) else PC_SLICE_LAUNCH1(DPL, NROWS, false) }
    if (D <= 16) PC_SLICE_LAUNCH(1, 1)
    else if (D <= 32) PC_SLICE_LAUNCH(1, 2)
    else if (D <= 64) PC_SLICE_LAUNCH(1, 4)
    else if (D <= 128) PC_SLICE_LAUNCH(2, 4)
    else if (D <= 256) PC_SLICE_LAUNCH(4, 4)
    else return 1;
#undef PC_SLICE_LAUNCH
#undef PC_SLICE_LAUNCHL
#undef PC_SLICE_LAUNCH1
    return 0;
}
