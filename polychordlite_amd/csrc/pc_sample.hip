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
template <int DPL, int NROWS>
__device__ __forceinline__ double like_eval(const PcState &S, const double (&th)[DPL], const LaneDims<DPL> &ld,
                                            int lane, double *ybuf)
{
    const int D = S.D;
    const PcLike &L = S.like;
    if (L.kind == PC_LIKE_GAUSSIAN) {            // gaussian.f90:25-34
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) { const double z = (th[k] - L.mu) * L.inv_sigma; s += z * z; }
        s = wsum<DPL, NROWS>(s);
        return L.norm - s / 2.0;
    } else if (L.kind == PC_LIKE_RASTRIGIN) {    // rastrigin.f90:33
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k)
            if (ld.on[k]) s += 8.515435146961291 /* log(4991.21750) */ + th[k] * th[k] - 10.0 * cos(PC_TWO_PI * th[k]);
        s = wsum<DPL, NROWS>(s);
        return -s;
    } else if (L.kind == PC_LIKE_TWIN_GAUSSIAN) {  // twin_gaussian.f90:29-46
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
template <int DPL, int NROWS>
__device__ __forceinline__ void like_phi(const PcState &S, const double (&th)[DPL], const LaneDims<DPL> &ld,
                                         int lane, double &phi0, double &phi1)
{
    phi0 = 0.0; phi1 = 0.0;
    if (S.nDer == 0) return;
    if (S.like.kind == PC_LIKE_GAUSSIAN) {        // gaussian.f90:36-37
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) r2 += (th[k] - S.like.mu) * (th[k] - S.like.mu);
        r2 = wsum<DPL, NROWS>(r2);
        phi0 = sqrt(r2);
        if (S.nDer >= 2) phi1 = pc_log_ball(phi0, S.D, S.like.log_vn);
    } else if (S.like.kind == PC_LIKE_TWIN_GAUSSIAN) {   // twin_gaussian.f90:48-52
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
    // dot products run on four partial sums: a dependent fp64 add costs ~32 cycles, a 20-term serial dot 640
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
    constexpr int DP = 4 * HV, NTQ = 16 * HV;
    // LDS rows are stored as four coordinate blocks of HV + 2 doubles: the four threads of a vector read their blocks
    // at the same time, and blocks exactly HV doubles (a multiple of 256 B) apart would all start in the same bank
    constexpr int HP = HV + 2, DPP = 4 * HP;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    double (*Qb)[DPP] = (double (*)[DPP])smem_q;                      // [2][DPP] pivot, double buffered
    double (*Lt)[DPP] = (double (*)[DPP])(smem_q + sizeof(double) * 2 * DPP);   // [HV][DPP] HV rows of the Cholesky factor (HV < 32)
    __shared__ int sh[2];
    const int D = S.D, nr = S.nr;
    const int tid = threadIdx.x, chain = blockIdx.y;
    const int i = tid >> 2, h = tid & 3, d0 = HV * h, p0 = HP * h;    // my vector, my coordinate block (p0: in LDS rows)
    int grade, basis;                                                 // chordal_sampling.f90:119-130, see k_nhats
    pc_grade_of_basis(S, blockIdx.x, grade, basis);
    const int off = pc_sel(S.g_off, grade), Dg = D - off, nrg = pc_sel(S.g_nr, grade), col0 = pc_sel(S.g_col0, grade);
    const bool active = i < Dg;
#ifdef NHATSQ_DBG
    long long qc[6]; qc[0] = clock64();
#endif
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
    double *rawb = (PART != 0) ? S.nhat_raw + ((size_t)chain * S.nb_total + blockIdx.x) * (size_t)(HV * NTQ) : nullptr;
    // gaussian deviates of my 32 coordinates: element (basis*D + i)*D + d of stream (batch, chain) in PC_DOM_NHAT,
    // two per Philox call
    double v[HV];
#pragma unroll
    for (int e = 0; e < HV; ++e) v[e] = 0.0;
    double lpre[(HV * DP + NTQ - 1) / NTQ];
    {
    if (active) {
        // stream element of my register 0 (coordinate d0); registers [r_lo, r_hi) hold coordinates that exist and move
        const long long e0 = (long long)(S.seq_mode ? (uint32_t)S.ctl->seq + 2u : 0u) + pc_sel(S.g_e0, grade)
                             + ((long long)basis * Dg + i) * Dg + (d0 - off);
        const int r_lo = max(0, off - d0), r_hi = min(HV, D - d0);
        if (r_hi > r_lo) {
            // Philox call number cbase + cc holds stream elements 2 (cbase + cc) and + 1, i.e. my registers 2 cc - par and
            // 2 cc - par + 1 with par = parity of e0: every call lands in one of two fixed register pairs
            const long long cbase = e0 >> 1;
            const bool par = (e0 & 1ll) != 0;
#pragma unroll
            for (int cc = 0; cc < HV / 2 + 1; ++cc) {
                const int ea = 2 * cc - (par ? 1 : 0), eb = ea + 1;
                if (eb >= r_lo && ea < r_hi) {
                    const uint32_t call = (uint32_t)(cbase + cc);
                    double ua, ub;
                    if (S.seq_mode) pc_uniform2(S.k0, S.k1, PC_DOM_SEQ, 0u, 0u, call, ua, ub);
                    else pc_uniform2(S.k0, S.k1, PC_DOM_NHAT, batch, (uint32_t)chain, call, ua, ub);
                    const double ga = pc_inv_normal_cdf(ua), gb = pc_inv_normal_cdf(ub);
                    if (2 * cc - 1 >= 0 && 2 * cc - 1 < HV) v[(2 * cc - 1 >= 0 && 2 * cc - 1 < HV) ? 2 * cc - 1 : 0] = par ? ga : v[(2 * cc - 1 >= 0 && 2 * cc - 1 < HV) ? 2 * cc - 1 : 0];
                    if (2 * cc < HV) v[2 * cc < HV ? 2 * cc : 0] = par ? gb : ga;
                    if (2 * cc + 1 < HV) v[2 * cc + 1 < HV ? 2 * cc + 1 : 0] = par ? v[2 * cc + 1 < HV ? 2 * cc + 1 : 0] : gb;
                }
            }
#pragma unroll
            for (int e = 0; e < HV; ++e) v[e] = (e >= r_lo && e < r_hi) ? v[e] : 0.0;   // coordinates that do not exist / do not move
        }
    }
#ifdef NHATSQ_DBG
    qc[1] = clock64();
#endif
#define PC_DOT32(RES, A, B) { double p0_ = 0.0, p1_ = 0.0, p2_ = 0.0, p3_ = 0.0; \
        _Pragma("unroll") for (int e = 0; e < HV; e += 4) { \
            p0_ += (A)[e] * (B)[e]; p1_ += (A)[e + 1] * (B)[e + 1]; p2_ += (A)[e + 2] * (B)[e + 2]; p3_ += (A)[e + 3] * (B)[e + 3]; } \
        RES = quad_sum((p0_ + p1_) + (p2_ + p3_)); }
    {   // random_direction (random_utils.F90:276-298)
        double n2;
        PC_DOT32(n2, v, v)
        const double inrm = active ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
        for (int e = 0; e < HV; ++e) v[e] *= inrm;
    }
    if (i == 0) {
#pragma unroll
        for (int e = 0; e < HV; ++e) Qb[0][p0 + e] = v[e];
    }
    __syncthreads();
    // first tile of the Cholesky factor: requested now, consumed after the loop
    if constexpr (HV < 32) {
        const double *Lc0 = S.chol + (size_t)sh[0] * D * D;
#pragma unroll
        for (int x = 0; x < (HV * DP + NTQ - 1) / NTQ; ++x) {
            const int y = tid + x * NTQ, r = y / DP, b = y % DP;
            lpre[x] = (y < HV * DP && r < D && b < D) ? Lc0[(size_t)r * D + b] : 0.0;
        }
    }
#ifdef NHATSQ_DBG
    qc[2] = clock64();
#endif
    // Gram-Schmidt (random_utils.F90:391-399): same projections as before, pivot unnormalised
    for (int j = 0; j < Dg; ++j) {
        // a wave whose sixteen vectors are all finished (i < j) only keeps the barrier company
        if ((((tid >> 6) + 1) << 4) <= j) { __syncthreads(); continue; }
        double q[HV];
#pragma unroll
        for (int e = 0; e < HV; ++e) q[e] = Qb[j & 1][p0 + e];
        double qq, dv;
        PC_DOT32(qq, q, q)
        PC_DOT32(dv, q, v)
        if (i == j) {
            const double inrm = 1.0 / sqrt(qq);
#pragma unroll
            for (int e = 0; e < HV; ++e) v[e] *= inrm;
        } else if (active && i > j) {
            const double cproj = dv / qq;
#pragma unroll
            for (int e = 0; e < HV; ++e) v[e] -= cproj * q[e];
            if (i == j + 1) {
#pragma unroll
                for (int e = 0; e < HV; ++e) Qb[(j + 1) & 1][p0 + e] = v[e];
            }
        }
        __syncthreads();
    }
#ifdef NHATSQ_DBG
    qc[3] = clock64();
#endif
    }
    if constexpr (PART == 1) {
        // the layout k_whiten reads as its B operand: [n][group of sixteen vectors][lk][vector in group], coordinate
        // 32 h + e = 8 (n >> 1) + 2 lk + (n & 1)
#pragma unroll
        for (int e = 0; e < HV; ++e) {
            const int n = 8 * h + 2 * (e >> 3) + (e & 1), lkk = (e & 7) >> 1;
            rawb[((size_t)n * 8 + (i >> 4)) * 64 + lkk * 16 + (i & 15)] = v[e];
        }
        return;
    }
    // whitening  w = L.n  (chordal_sampling.f90:73): tile k holds rows 32k..32k+31 of L, i.e. exactly the output
    // coordinates of block h = k
    const int col = col0 + basis * Dg + i;
    const bool wanted = basis * Dg + i < nrg;
    const double *Lc = S.chol + (size_t)sh[0] * D * D;
    if constexpr (HV == 32) {
        // nDims 65..128: W = L.N on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  The basis goes to LDS (row = vector,
        // odd row stride), L passes through LDS sixteen rows at a time (lower triangle only), wave tj owns the sixteen
        // output columns (vectors) 16 tj ..: it keeps their tiles in registers, normalises, and writes whole rows.
        // Operand maps as in k_cov_partial: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15], D row = (lane>>4)+4 reg.
        typedef double v4d __attribute__((ext_vector_type(4)));
        constexpr int NS = 129;
        const int nt = (D + 15) >> 4, NR = nt * 16;
        double *Nl = (double *)smem_q + 2 * DPP;                       // [NR][NS]
        double *L16 = Nl + (size_t)NR * NS;                            // [2][16][NS]: the tile in use and the next one
        const bool dbuf = NR <= 112;                                   // (beyond: one buffer, and a barrier before it is refilled)
        // tiles of L (and of M below) travel global -> registers -> LDS one tile ahead of the matrix cores: one barrier per tile
        double pre[4];
        auto load_L = [&](int ti) __attribute__((always_inline)) {
            const int kmax = min(NR, 16 * (ti + 1));                   // L(a, b) = 0 for b > a
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = tid + q * NTQ, r = x / kmax, bcol = x - r * kmax, arow = 16 * ti + r;
                pre[q] = (x < 16 * kmax && arow < D && bcol < D) ? Lc[(size_t)arow * D + bcol] : 0.0;
            }
        };
        auto store_L = [&](int ti) __attribute__((always_inline)) {
            const int kmax = min(NR, 16 * (ti + 1));
            double *buf = L16 + (size_t)(dbuf ? (ti & 1) : 0) * 16 * NS;
            if (!dbuf) __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = tid + q * NTQ, r = x / kmax, bcol = x - r * kmax;
                if (x < 16 * kmax) buf[(size_t)r * NS + bcol] = pre[q];
            }
        };
        load_L(0);
        if (i < NR) {
#pragma unroll
            for (int e = 0; e < HV; ++e) Nl[(size_t)i * NS + d0 + e] = v[e];    // zero beyond nDims and beyond the basis
        }
        const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
        v4d acc[8];
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) acc[ti] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            if (ti < nt) {
                store_L(ti);
                if (ti + 1 < nt) load_L(ti + 1);
                __syncthreads();                                       // tile ti (and, the first time, the basis) complete
                if (wv < nt) {
                    const int kmax = min(NR, 16 * (ti + 1));
                    const double *pa = L16 + (size_t)(dbuf ? (ti & 1) : 0) * 16 * NS + (size_t)li * NS + lk;
                    const double *pb = Nl + (size_t)(16 * wv + li) * NS + lk;
                    v4d a4 = v4d{0.0, 0.0, 0.0, 0.0};
                    for (int k0 = 0; k0 < kmax; k0 += 4) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], a4, 0, 0, 0);
                    acc[ti] = a4;
                }
            }
        }
        if (wv < nt) {
            // |w| of my column (chordal_sampling.f90:80-82): my four row groups, then the other three lane groups
            double n2 = 0.0;
#pragma unroll
            for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) n2 += acc[ti][r] * acc[ti][r];
            n2 += __shfl_xor(n2, 16); n2 += __shfl_xor(n2, 32);
            const double wn = sqrt(n2), iw = 1.0 / wn;
            const int ivec = 16 * wv + li;                             // vector of this basis = column
            double *mine = Nl + (size_t)ivec * NS;                     // the basis rows of this wave are no longer read
#pragma unroll
            for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) if (ti < nt) mine[16 * ti + lk + 4 * r] = acc[ti][r] * iw;
            if (lk == 0 && ivec < Dg && basis * Dg + ivec < nrg) S.nhat_w[(size_t)chain * nr + col0 + basis * Dg + ivec] = wn * 3.0;
            // rows leave coalesced
            for (int cc = 0; cc < 16; ++cc) {
                const int iv = 16 * wv + cc;
                if (iv < Dg && basis * Dg + iv < nrg) {
                    double *out = S.nhat + ((size_t)chain * nr + col0 + basis * Dg + iv) * D;
                    const double *src = Nl + (size_t)iv * NS;
                    for (int a2 = lane; a2 < D; a2 += 64) out[a2] = src[a2];
                }
            }
        }
        if (S.nhat_Ms != nullptr) {
            // correlated Gaussian (random_gaussian.f90:17-30): along a chord the exponent is quadratic (see ChainCtx) and all a
            // slice needs of the matrix is M.s, s = span o n^.  Every direction of the chain is known HERE: the products are
            // one more [D x D] x [D x 16] pass per wave on the matrix cores (rows of M streamed through the tile buffer of L)
            // instead of a matrix-vector product per slice inside the chain.  Wave 0 of the chain's first basis also forms
            // M.(theta_seed - mean), the one product the chain needs for its start point.
            double *spn = (double *)smem_q, *y0s = spn + 128;           // the pivot buffers are free now (2 x 136 doubles)
            const double *Mt = S.like.invcov;                           // Mt[b * D + a] = M(a, b)
            auto load_M = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int x = tid + q * NTQ, r = x & 15, bcol = x >> 4, arow = 16 * ti + r;
                    pre[q] = (x < 16 * NR && arow < D && bcol < D) ? Mt[(size_t)bcol * D + arow] : 0.0;
                }
            };
            auto store_M = [&](int ti) __attribute__((always_inline)) {
                double *buf = L16 + (size_t)(dbuf ? (ti & 1) : 0) * 16 * NS;
                if (!dbuf) __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int x = tid + q * NTQ, r = x & 15, bcol = x >> 4;
                    if (x < 16 * NR) buf[(size_t)r * NS + bcol] = pre[q];
                }
            };
            __syncthreads();
            if (tid < NR) {
                const bool on = tid < D;
                const double lo = (on && S.prior.lo) ? S.prior.lo[tid] : 0.0, hi = (on && S.prior.hi) ? S.prior.hi[tid] : 1.0;
                spn[tid] = on ? hi - lo : 0.0;
                const double c0 = on ? S.live[(size_t)sh[1] * S.nT + tid] : 0.0;
                y0s[tid] = on ? (lo + (hi - lo) * c0) - (S.like.mean ? S.like.mean[tid] : 0.0) : 0.0;
            }
            load_M(0);
            __syncthreads();
            if (wv < nt) {                                              // my sixteen rows: n^ -> s
                for (int x = lane; x < 16 * NR; x += 64) { const int r = x / NR, a2 = x % NR; Nl[(size_t)(16 * wv + r) * NS + a2] *= spn[a2]; }
            }
            v4d ac2[8];
#pragma unroll
            for (int ti = 0; ti < 8; ++ti) ac2[ti] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ti = 0; ti < 8; ++ti) {
                if (ti < nt) {
                    store_M(ti);
                    if (ti + 1 < nt) load_M(ti + 1);
                    __syncthreads();
                    const double *tile = L16 + (size_t)(dbuf ? (ti & 1) : 0) * 16 * NS;
                    if (wv < nt) {
                        const double *pa = tile + (size_t)li * NS + lk;
                        const double *pb = Nl + (size_t)(16 * wv + li) * NS + lk;
                        v4d a4 = v4d{0.0, 0.0, 0.0, 0.0};
                        for (int k0 = 0; k0 < NR; k0 += 4) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], a4, 0, 0, 0);
                        ac2[ti] = a4;
                    }
                    if (wv == 7 && blockIdx.x == 0) {                  // M.y0, rows of this tile: lane = (row, quarter of the columns)
                        double t = 0.0;
                        for (int b = lk; b < NR; b += 4) t += tile[(size_t)li * NS + b] * y0s[b];
                        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
                        if (lk == 0 && 16 * ti + li < D) S.ch_My[(size_t)chain * D + 16 * ti + li] = t;
                    }
                }
            }
            if (wv < nt) {
                double *mine = Nl + (size_t)(16 * wv + li) * NS;
#pragma unroll
                for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (ti < nt) mine[16 * ti + lk + 4 * r] = ac2[ti][r];
                for (int cc = 0; cc < 16; ++cc) {
                    const int iv = 16 * wv + cc;
                    if (iv < Dg && basis * Dg + iv < nrg) {
                        double *out = S.nhat_Ms + ((size_t)chain * nr + col0 + basis * Dg + iv) * D;
                        const double *src = Nl + (size_t)iv * NS;
                        for (int a2 = lane; a2 < D; a2 += 64) out[a2] = src[a2];
                    }
                }
            }
        }
        return;
    }
    double w[HV];
#pragma unroll
    for (int e = 0; e < HV; ++e) w[e] = 0.0;
    for (int k = 0; k * HV < D; ++k) {
        __syncthreads();
        if (k == 0) {
            // the first tile was requested before the Gram-Schmidt loop (registers lpre): its latency is hidden
#pragma unroll
            for (int x = 0; x < (HV * DP + NTQ - 1) / NTQ; ++x) { const int y = tid + x * NTQ, b = y % DP; if (y < HV * DP) Lt[y / DP][(b / HV) * HP + b % HV] = lpre[x]; }
        } else {
            for (int x = tid; x < HV * DP; x += NTQ) {
                const int r = x / DP, b = x % DP, a = HV * k + r;
                Lt[r][(b / HV) * HP + b % HV] = (a < D && b < D) ? Lc[(size_t)a * D + b] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < HV; ++r) {
            double t;
            PC_DOT32(t, (&Lt[r][p0]), v)
            if (h == k) w[r] = t;
        }
    }
    if (active && wanted) {
        double n2;
        PC_DOT32(n2, w, w)
        const double wn = sqrt(n2), iw = 1.0 / wn;              // chordal_sampling.f90:80-82
        double *out = S.nhat + ((size_t)chain * nr + col) * D + d0;
#pragma unroll
        for (int e = 0; e < HV; ++e) if (d0 + e < D) out[e] = w[e] * iw;
        if (h == 0) S.nhat_w[(size_t)chain * nr + col] = wn * 3.0;
    }
#undef PC_DOT32
#ifdef NHATSQ_DBG
    qc[4] = clock64();
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) for (int x = 0; x < 4; ++x) S.ctl->gen_cyc[x] += qc[x + 1] - qc[x];
#endif
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
template <int DPL, int NROWS>
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
    const double logL = like_eval<DPL, NROWS>(C.S, th, C.ld, C.lane, C.ybuf);
    if (logL > C.S.logzero) C.nlike++;
    return logL;
}

// Two independent trial points at once (the two ends of the initial bracket): the per-point work is a
// dependent chain (FMA -> compare -> reduction), so the second evaluation rides in the shadow of the first.
// Straight-line code: both likelihoods are computed unconditionally and masked afterwards.
template <int DPL, int NROWS>
__device__ __forceinline__ void eval_pair(ChainCtx<DPL, NROWS> &C, const double (&x0)[DPL], const double (&nh)[DPL],
                                          double tA, double tB, double &lA, double &lB)
{
    const PcLike &L = C.S.like;
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
    bool outA = false, outB = false;
    double sA = 0.0, sB = 0.0, s2A = 0.0, s2B = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        const double cA = x0[k] + tA * nh[k], cB = x0[k] + tB * nh[k];
        if (C.ld.on[k]) { outA |= (cA < 0.0) | (cA > 1.0); outB |= (cB < 0.0) | (cB > 1.0); }
        const double thA = C.ld.lo[k] + C.ld.span[k] * cA, thB = C.ld.lo[k] + C.ld.span[k] * cB;
        if (C.ld.on[k]) {
            if (L.kind == PC_LIKE_RASTRIGIN) {
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
    if (L.kind == PC_LIKE_RASTRIGIN) { lA = -sA; lB = -sB; }
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
template <int DPL, int NROWS, bool SPECIAL, int WPB = 1, int FW = 0>
__global__ PC_SLICE_ATTR __launch_bounds__(64 * WPB) void k_slice(PcState S, unsigned batch, int phi_lds, int mat_lds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // enqueued before the host knew how the previous nursery ended: only if the contraction left its go-ahead for THIS nursery
    if (S.spec_guard && S.ctl->spec_ok != (int)batch) return;
    __builtin_amdgcn_s_setprio(3);                 // a chain is one long dependent instruction stream: it goes first on its SIMD
#if defined(SLICE_DBG) && SLICE_DBG == 2
    const long long kc0 = clock64(), kw0 = wall_clock64();
#endif
    const int lane = threadIdx.x & 63, wv = (WPB > 1) ? (int)(threadIdx.x >> 6) : 0, chain = blockIdx.x * WPB + wv;
    const size_t per_wave = ((size_t)S.D + S.nr + (phi_lds ? (size_t)S.nr * (S.D + 1) : 0) + 1) & ~(size_t)1;   // doubles
    double *ybuf = (double *)smem + (size_t)wv * per_wave;   // [D] (corr gaussian only)
    int *sdeck = (int *)(ybuf + S.D);              // [nr] deck, only used when nr > 64
    int *sj = sdeck + S.nr;                        // [nr]
    double *tbuf = ybuf + S.D + S.nr;              // [nr][D+1] theta of every baby (when it fits: phi_lds)
    double *Mlds = (double *)smem + (size_t)WPB * per_wave;  // [D][D] inverse covariance, transposed (mat_lds), shared
    const int D = S.D, nr = S.nr, nT = S.nT;
    const double logzero = S.logzero;
    const bool seq_mode = SPECIAL && S.seq_mode != 0, graded = SPECIAL && S.ngrade > 1;

    LaneDims<DPL> ld;
    int slot, seed_cluster = 0;
    double contour;
    // FW: the loads of the whitening prologue are issued first -- raw direction `lane` (+64, ...) and the Cholesky factor of
    // cluster 0 (the seed's cluster unless there are several) -- and travel while the seed is chosen and the deck shuffled
    constexpr int FWN = FW > 0 ? FW : 1;
    constexpr int FWL = FW > 0 ? (FW * FW + 63) / 64 : 1;
    double vv0[FWN], Lpre[FWL];
    if constexpr (FW > 0) {
        const double *rawc = S.nhat_raw + (size_t)chain * S.nb_total * D * D;
#pragma unroll
        for (int d = 0; d < FWN; ++d) vv0[d] = (d < D && lane < nr) ? rawc[(size_t)lane * D + d] : 0.0;
#pragma unroll
        for (int q = 0; q < FWL; ++q) { const int e = lane + 64 * q; Lpre[q] = (e < D * D) ? S.chol[e] : 0.0; }
    }
    if constexpr (FW > 0) {
        int sel = 0, sl = 0;
        if (lane == 0) {                          // GenerateSeed (generate.F90:19-55), nested_sampling.F90:267-273
            select_seed(S, batch, chain, sel, sl);
            S.ch_cluster[chain] = sel; S.ch_seed_slot[chain] = sl;
            S.ch_contour[chain] = S.logLp[sel];                      // nested_sampling.F90:270
            S.ch_epoch[chain] = S.ctl->admin_epoch;
            if (chain == 0) { S.ctl->i_nursery = gridDim.x * WPB; S.ctl->batch_id = batch; }
        }
        seed_cluster = __builtin_amdgcn_readfirstlane(sel); slot = __builtin_amdgcn_readfirstlane(sl);
        contour = S.logLp[seed_cluster];
    } else { slot = S.ch_seed_slot[chain]; contour = S.ch_contour[chain]; }
#if defined(SLICE_DBG) && SLICE_DBG == 2
    const long long kp1 = clock64();
#endif
    double x0[DPL];
    {
        const double *seed = S.live + (size_t)slot * nT;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int dim = lane + 64 * k;
            ld.on[k] = dim < D;
            const double lo = (ld.on[k] && S.prior.lo) ? S.prior.lo[dim] : 0.0;
            const double hi = (ld.on[k] && S.prior.hi) ? S.prior.hi[dim] : 1.0;
            ld.lo[k] = lo; ld.span[k] = hi - lo;
            ld.mean[k] = (ld.on[k] && S.like.mean) ? S.like.mean[dim] : 0.0;
            x0[k] = ld.on[k] ? seed[dim] : 0.5;
        }
    }
    // pool mode: the babies land in this chain's rows of the phantom array; none of them is a phantom before the chain is consumed
    if (S.pool) for (int i = lane; i < nr; i += 64) S.ph_cuid[(size_t)S.pool_base + (size_t)chain * nr + i] = PC_CUID_NONE;
#if defined(SLICE_DBG) && SLICE_DBG == 2
    const long long kp2 = clock64();
#endif
    ChainCtx<DPL, NROWS> C{S, ld, lane, ybuf, 0, false, 0.0, 0.0, 0.0, 0.0};
    const bool corr = S.like.kind == PC_LIKE_CORR_GAUSSIAN;
    C.quad = (corr || S.like.kind == PC_LIKE_GAUSSIAN) && !(S.ablate & 1);
    C.qnorm = corr ? -((double)D * PC_LOG_TWO_PI + S.like.logdetcov) / 2.0 : S.like.norm;
    // correlated Gaussian: y = theta - mean and M.y travel with the chain (updated, not recomputed, at every
    // accepted point); the matrix is read from LDS when it fits
    const double *Mt = S.like.invcov;             // transposed: Mt[b*D + a] = M(a,b), lanes read consecutive a
    double yv[DPL], My[DPL], sv[DPL], Ms[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) { yv[k] = 0.0; My[k] = 0.0; sv[k] = 0.0; Ms[k] = 0.0; }
    auto matvec = [&](const double (&vec)[DPL], double (&out)[DPL]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) if (ld.on[k]) ybuf[lane + 64 * k] = vec[k];
        __syncthreads();                             // one wave per workgroup
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int a = lane + 64 * k;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            if (ld.on[k]) {
                int b = 0;
                for (; b + 3 < D; b += 4) {
                    t0 += Mt[(size_t)b * D + a] * ybuf[b]; t1 += Mt[(size_t)(b + 1) * D + a] * ybuf[b + 1];
                    t2 += Mt[(size_t)(b + 2) * D + a] * ybuf[b + 2]; t3 += Mt[(size_t)(b + 3) * D + a] * ybuf[b + 3];
                }
                for (; b < D; ++b) t0 += Mt[(size_t)b * D + a] * ybuf[b];
            }
            out[k] = (t0 + t1) + (t2 + t3);
        }
        __syncthreads();
    };
    // msp: M.s of every direction and M.y of the start point were formed by k_nhats_q on the matrix cores (nhat_Ms, ch_My):
    // no matrix here at all; y and M.y are carried along the chords for the whole chain (a rounding error per slice, the
    // size of the one a direct evaluation makes)
    const bool msp = corr && S.nhat_Ms != nullptr && C.quad;
    double Ms_next[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) Ms_next[k] = 0.0;
    if (corr) {
        if (mat_lds) {
            for (int e = threadIdx.x; e < D * D; e += 64 * WPB) Mlds[e] = S.like.invcov[e];
            __syncthreads();
            Mt = Mlds;
        }
#pragma unroll
        for (int k = 0; k < DPL; ++k) yv[k] = ld.on[k] ? (ld.lo[k] + ld.span[k] * x0[k]) - ld.mean[k] : 0.0;
        if (msp) {
#pragma unroll
            for (int k = 0; k < DPL; ++k) My[k] = ld.on[k] ? S.ch_My[(size_t)chain * D + lane + 64 * k] : 0.0;
        } else matvec(yv, My);
        double pa = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) pa += yv[k] * My[k];
        C.qa = wsum<DPL, NROWS>(pa);
    }

    // ---- deck: first direction stays, the others are Fisher-Yates shuffled (chordal_sampling.f90:135-142,
    //      random_utils.F90:505-532).  deck value for position p lives in lane p when nr <= 64.
    // seq_mode: stream positions of the shuffle draws (after the seed draws and every basis) and of the slice draws
    const unsigned long long seq_g0 = seq_mode ? S.ctl->seq + 2ull + (unsigned long long)S.n_dev : 0ull;
    unsigned long long seq_run = seq_g0 + (unsigned long long)(nr - 1);
    int deck = lane;
    const bool deck_in_regs = nr <= 64;
    if (deck_in_regs) {
        int jv = 0;
        if (lane >= 1 && lane < nr) {
            const double u = seq_mode ? pc_seq_uniform(S, seq_g0 + (unsigned long long)(nr - 1 - lane))
                                        : pc_uniform(S.k0, S.k1, PC_DOM_SHUFFLE, batch, (uint32_t)chain, (uint32_t)lane);
            int j = (int)ceil(u * lane);
            jv = j < 1 ? 1 : (j > lane ? lane : j);
        }
        for (int i = nr - 1; i >= 1; --i) {
            const int j = __builtin_amdgcn_readlane(jv, i);
            const int di = __builtin_amdgcn_readlane(deck, i), dj = __builtin_amdgcn_readlane(deck, j);
            deck = (lane == i) ? dj : ((lane == j) ? di : deck);
        }
    } else {
        for (int i = lane; i < nr; i += 64) {
            sdeck[i] = i;
            if (i >= 1) {
                const double u = seq_mode ? pc_seq_uniform(S, seq_g0 + (unsigned long long)(nr - 1 - i))
                                            : pc_uniform(S.k0, S.k1, PC_DOM_SHUFFLE, batch, (uint32_t)chain, (uint32_t)i);
                int j = (int)ceil(u * i);
                sj[i] = j < 1 ? 1 : (j > i ? i : j);
            }
        }
        __syncthreads();
        if (lane == 0)
            for (int i = nr - 1; i >= 1; --i) { const int j = sj[i], t = sdeck[i]; sdeck[i] = sdeck[j]; sdeck[j] = t; }
        __syncthreads();
    }

#if defined(SLICE_DBG) && SLICE_DBG == 2
    const long long kp3 = clock64();
#endif
    double ua = 0.0, ub = 0.0;                     // uniforms of 4 consecutive slices, 32 each
    double nh[DPL], nh_next[DPL], w_next;
    // ---- FW: all directions of the chain whitened up front, lane = direction (what a thread of k_nhats did for its
    //      vector, same loops): w = L n (chordal_sampling.f90:73), |w|, n^ = w / |w|, width 3 |w| (:80-82); results in LDS
    double *nhs = nullptr, *wsh = nullptr;
    if constexpr (FW > 0) {
        double *Lsh = tbuf + (phi_lds ? (size_t)nr * (D + 1) : 0);     // [FW][D] Cholesky factor, rows past D zero
        nhs = Lsh + (size_t)FW * D;                                      // [nr][D + 1]
        wsh = nhs + (size_t)nr * (D + 1);                               // [nr]
        if (seed_cluster != 0) {                                         // (uniform) several clusters: the seed's factor
            const double *Lg = S.chol + (size_t)seed_cluster * D * D;
#pragma unroll
            for (int q = 0; q < FWL; ++q) { const int e = lane + 64 * q; Lpre[q] = (e < D * D) ? Lg[e] : 0.0; }
        }
#pragma unroll
        for (int q = 0; q < FWL; ++q) { const int e = lane + 64 * q; if (e < FW * D) Lsh[e] = Lpre[q]; }
        __syncthreads();                                                // one wave per workgroup
        const double *rawc = S.nhat_raw + (size_t)chain * S.nb_total * D * D;   // direction v (generation order) at + v * D
        for (int v0 = 0; v0 < nr; v0 += 64) {
            const int v = v0 + lane;
            if (v < nr) {
                double vv[FWN], t[FWN];
#pragma unroll
                for (int d = 0; d < FWN; ++d) { vv[d] = (v0 == 0) ? vv0[d] : ((d < D) ? rawc[(size_t)v * D + d] : 0.0); t[d] = 0.0; }
#pragma unroll
                for (int bb = 0; bb < FWN; ++bb)
#pragma unroll
                    for (int aa = bb; aa < FWN; ++aa) t[aa] += Lsh[(size_t)aa * D + bb] * vv[bb];
#pragma unroll
                for (int aa = 0; aa < FWN; ++aa) if (aa >= D) t[aa] = 0.0;
                double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
                for (int d = 0; d < FWN; d += 4) { p0 += t[d] * t[d]; p1 += t[d + 1] * t[d + 1]; p2 += t[d + 2] * t[d + 2]; p3 += t[d + 3] * t[d + 3]; }
                const double wn = sqrt((p0 + p1) + (p2 + p3)), iw = 1.0 / wn;
#pragma unroll
                for (int d = 0; d < FWN; ++d) if (d < D) nhs[(size_t)v * (D + 1) + d] = t[d] * iw;
                wsh[v] = wn * 3.0;
            }
        }
        __syncthreads();
        const int v0 = deck_in_regs ? __builtin_amdgcn_readlane(deck, 0) : sdeck[0];
        nh_next[0] = (lane < D) ? nhs[(size_t)v0 * (D + 1) + lane] : 0.0;
        w_next = wsh[v0];
    } else
    {   // prefetch the first direction
        const int v0 = deck_in_regs ? __builtin_amdgcn_readlane(deck, 0) : sdeck[0];
        const double *p = S.nhat + ((size_t)chain * nr + v0) * D;
#pragma unroll
        for (int k = 0; k < DPL; ++k) nh_next[k] = ld.on[k] ? p[lane + 64 * k] : 0.0;
        w_next = S.nhat_w[(size_t)chain * nr + v0];
        if (msp) {
            const double *pm = S.nhat_Ms + ((size_t)chain * nr + v0) * D;
#pragma unroll
            for (int k = 0; k < DPL; ++k) Ms_next[k] = ld.on[k] ? pm[lane + 64 * k] : 0.0;
        }
    }

#ifdef SLICE_DBG
    long long scy[6] = {0, 0, 0, 0, 0, 0}; long long nev = 0, ev2 = 0;
#if SLICE_DBG == 2
    const long long kc1 = clock64();
#endif
#endif
    double w = w_next;
#pragma unroll
    for (int k = 0; k < DPL; ++k) nh[k] = nh_next[k];
    // loop-invariant addresses (the per-slice address arithmetic was a third of the slice's instructions)
    const double *nh_base = S.nhat + (size_t)chain * nr * D + lane;
    const double *nw_base = S.nhat_w + (size_t)chain * nr;
    const double *nm_base = msp ? S.nhat_Ms + (size_t)chain * nr * D + lane : nh_base;
    if (msp) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) Ms[k] = Ms_next[k];
    }
    double *row = S.babies + (size_t)chain * nr * nT;
    double *bl_row = S.baby_logL + (size_t)chain * nr;
    double *bl_col = S.baby_logL_T + chain;
    double *tb_row = tbuf + lane;
    const int o_p0 = S.p0, o_d0 = S.d0, o_b0 = S.b0, o_l0 = S.l0, nDer = S.nDer, Bstride = S.B;
    int nl_grade[PC_MAX_GRADE];                    // evaluations per grade (chordal_sampling.f90:84), static indices only
#pragma unroll
    for (int g = 0; g < PC_MAX_GRADE; ++g) nl_grade[g] = 0;
    for (int s = 0; s < nr; ++s, row += nT, bl_col += Bstride, tb_row += D + 1) {
#ifdef SLICE_DBG
        const long long c0 = clock64();
#endif
        const int nl_before = C.nlike;
        int my_grade = 0;
        if (graded) my_grade = pc_grade_of(S, deck_in_regs ? __builtin_amdgcn_readlane(deck, s) : sdeck[s]);
        if constexpr (FW > 0) {
            if (s + 1 < nr) {
                const int v1 = deck_in_regs ? __builtin_amdgcn_readlane(deck, s + 1) : sdeck[s + 1];
                nh_next[0] = (lane < D) ? nhs[(size_t)v1 * (D + 1) + lane] : 0.0;
                w_next = wsh[v1];
            }
        } else
        if (s + 1 < nr) {                           // prefetch the next direction (hidden under this slice)
            const int v1 = deck_in_regs ? __builtin_amdgcn_readlane(deck, s + 1) : sdeck[s + 1];
            const double *p = nh_base + v1 * D;
#pragma unroll
            for (int k = 0; k < DPL; ++k) nh_next[k] = ld.on[k] ? p[64 * k] : 0.0;
            w_next = nw_base[v1];
            if (msp) {
                const double *pm = nm_base + v1 * D;
#pragma unroll
                for (int k = 0; k < DPL; ++k) Ms_next[k] = ld.on[k] ? pm[64 * k] : 0.0;
            }
        }
        if ((s & 3) == 0 && !seq_mode) {          // one Philox call per lane covers 4 slices x 32 uniforms
            const uint32_t sl = (uint32_t)s + (uint32_t)(lane >> 4);
            pc_uniform2(S.k0, S.k1, PC_DOM_SLICE, batch, (uint32_t)chain,
                        (sl * PC_SLICE_STRIDE) / 2 + (uint32_t)(lane & 15), ua, ub);
        }
        uint32_t kdraw = 0;
        auto next_u = [&]() -> double {
            if (seq_mode) return pc_seq_uniform(S, seq_run++);
            const uint32_t k = kdraw++;
            if (k < 32u) {
                const int src = ((s & 3) << 4) + (int)(k >> 1);
                return (k & 1u) ? readlane_f64(ub, src) : readlane_f64(ua, src);
            }
            return pc_uniform(S.k0, S.k1, PC_DOM_SLICE, batch, (uint32_t)chain, (uint32_t)s * PC_SLICE_STRIDE + k);
        };

#ifdef SLICE_DBG
        const long long c1 = clock64();
#endif
        double cube[DPL], th[DPL];
        if (C.quad) {
            double pa = 0.0, pb = 0.0, pc = 0.0;
            if (!corr) {
#pragma unroll
                for (int k = 0; k < DPL; ++k) {
                    const double zA = ((ld.lo[k] + ld.span[k] * x0[k]) - S.like.mu) * S.like.inv_sigma;
                    const double zB = (ld.span[k] * nh[k]) * S.like.inv_sigma;
                    if (ld.on[k]) { pa += zA * zA; pb += zA * zB; pc += zB * zB; }
                }
                C.qa = wsum<DPL, NROWS>(pa);
            } else {
#pragma unroll
                for (int k = 0; k < DPL; ++k) sv[k] = ld.on[k] ? ld.span[k] * nh[k] : 0.0;
                if (!msp) matvec(sv, Ms);
#pragma unroll
                for (int k = 0; k < DPL; ++k) { pb += sv[k] * My[k]; pc += sv[k] * Ms[k]; }
            }
            C.qb = wsum<DPL, NROWS>(pb); C.qc = wsum<DPL, NROWS>(pc);
        }
        // initial bracket (chordal_sampling.f90:213-219)
        const double u0 = next_u();
        double tR = (1 - u0) * w, tL = -(u0 * w);
        double lR, lL;
        eval_pair<DPL, NROWS>(C, x0, nh, tR, tL, lR, lL);
#ifdef SLICE_DBG
        const long long c2 = clock64();
#endif
        // stepping out (:223-236)
        int istep = 0;
        while (lR >= contour && lR > logzero) { istep++; tR = w * istep; lR = eval_at<DPL, NROWS>(C, x0, nh, tR, cube, th); }
        istep = 0;
        while (lL >= contour && lL > logzero) { istep++; tL = -(w * istep); lL = eval_at<DPL, NROWS>(C, x0, nh, tL, cube, th); }
#ifdef SLICE_DBG
        const long long c3 = clock64();
#endif
        // shrinkage (:240-271)
        double lnew = logzero, t_last = 0.0;
        bool ok = false;
        int it0 = 0;
        if (C.quad && !seq_mode) {
            // The first four trial points at once.  Where trial q lands does not depend on the likelihood of the trials
            // before it, only on their positions (a rejected trial becomes the bracket end on its side of x0), and the
            // uniforms are known: so the four positions "if everything before was rejected" are computed up front and
            // their likelihoods -- closed form along the chord, a handful of dependent fp64 operations of ~32 cycles each
            // on a wave that has its SIMD to itself -- are evaluated side by side instead of one after the other.  The
            // first accepted one is the baby; the trials after it never happened (not counted, their draws given back):
            // the same result and the same likelihood count as the loop, a third of its latency.
            constexpr int NSP = 4;
            const uint32_t kd0 = kdraw;
            double tc[NSP], lc[NSP], tLb[NSP], tRb[NSP];
            bool oc[NSP];
            double tLc = tL, tRc = tR;
#pragma unroll
            for (int q = 0; q < NSP; ++q) {
                const double dl = fabs(tLc), dr = fabs(tRc);
                const double t = next_u() * (dr + dl) - dl;
                tc[q] = t;
                if (t > 0.0) tRc = t; else tLc = t;
                tLb[q] = tLc; tRb[q] = tRc;                         // the bracket after rejecting trial q
                bool outside = false;
#pragma unroll
                for (int k = 0; k < DPL; ++k) {
                    const double cb = x0[k] + t * nh[k];
                    if (ld.on[k]) outside |= (cb < 0.0) | (cb > 1.0);
                }
                oc[q] = __ballot(outside) != 0ull;
                lc[q] = C.qnorm - (C.qa + t * (2.0 * C.qb + t * C.qc)) / 2.0;
            }
            int acc = -1;
#pragma unroll
            for (int q = 0; q < NSP; ++q) {
                if (acc < 0) {
                    const double lg = oc[q] ? logzero : lc[q];     // calculate.f90:36-38
                    if (!oc[q] && lg > logzero) C.nlike++;
                    t_last = tc[q]; lnew = lg;
                    if (lg < contour || lg <= logzero) { tL = tLb[q]; tR = tRb[q]; }
                    else acc = q;
                }
            }
            if (acc >= 0) {
                ok = true; kdraw = kd0 + (uint32_t)acc + 1u;
#pragma unroll
                for (int k = 0; k < DPL; ++k) { cube[k] = x0[k] + t_last * nh[k]; th[k] = ld.lo[k] + ld.span[k] * cube[k]; }
            }
            it0 = NSP;
        }
        for (int it = it0; it <= 100 && !ok; ++it) {
            const double dl = fabs(tL), dr = fabs(tR);
            const double t = next_u() * (dr + dl) - dl;
            t_last = t;
            lnew = eval_at<DPL, NROWS>(C, x0, nh, t, cube, th);
#ifdef SLICE_DBG
            nev++;
#endif
            if (lnew < contour || lnew <= logzero) { if (t > 0.0) tR = t; else tL = t; }
            else ok = true;
        }
        if (!ok) lnew = logzero;                    // "Non deterministic loglikelihood"
        if (corr) {                                 // the next start point: y and M.y move along the chord
            C.qa = C.qa + t_last * (2.0 * C.qb + t_last * C.qc);
#pragma unroll
            for (int k = 0; k < DPL; ++k) { yv[k] += t_last * sv[k]; My[k] += t_last * Ms[k]; }
            if ((s & 15) == 15 && !msp) {            // resynchronise the carried products now and then
                matvec(yv, My);
                double pa = 0.0;
#pragma unroll
                for (int k = 0; k < DPL; ++k) pa += yv[k] * My[k];
                C.qa = wsum<DPL, NROWS>(pa);
            }
        }
#ifdef SLICE_DBG
        const long long c4 = clock64();
#endif
        // the baby becomes the next start point (chordal_sampling.f90:85-88)
        // The prefetched direction is taken over BEFORE this slice's stores are issued: its loads were
        // issued a whole slice ago, whereas a wait placed after the stores (vmcnt counts them too on
        // gfx9) would stall every slice for a full store round trip.
        w = w_next;
#pragma unroll
        for (int k = 0; k < DPL; ++k) { nh[k] = nh_next[k]; asm volatile("" : "+v"(nh[k])); }
        asm volatile("" : "+v"(w));
        if (msp) {
#pragma unroll
            for (int k = 0; k < DPL; ++k) { Ms[k] = Ms_next[k]; asm volatile("" : "+v"(Ms[k])); }
        }
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            x0[k] = cube[k];
            if (ld.on[k]) { row[lane + 64 * k] = cube[k]; row[o_p0 + lane + 64 * k] = th[k]; }
        }
        if (phi_lds) {
#pragma unroll
            for (int k = 0; k < DPL; ++k) if (ld.on[k]) tb_row[64 * k] = th[k];
        } else if (nDer > 0) {
            double phi0, phi1;
            like_phi<DPL, NROWS>(S, th, ld, lane, phi0, phi1);
            if (lane == 0) {
                row[o_d0] = phi0;
                if (nDer >= 2) row[o_d0 + 1] = phi1;
                for (int e = 2; e < nDer; ++e) row[o_d0 + e] = 0.0;
            }
        }
        if (lane == 0) {
            row[o_b0] = contour;                    // nested_sampling.F90:260
            row[o_l0] = lnew;
            bl_row[s] = lnew; *bl_col = lnew;
        }
        if (graded) {
#pragma unroll
            for (int g = 0; g < PC_MAX_GRADE; ++g) nl_grade[g] += (my_grade == g) ? C.nlike - nl_before : 0;
        }
#ifdef SLICE_DBG
        const long long c5 = clock64();
        scy[0] += c1 - c0; scy[1] += c2 - c1; scy[2] += c3 - c2; scy[3] += c4 - c3; scy[4] += c5 - c4;
#endif
    }
#if defined(SLICE_DBG) && SLICE_DBG == 2
    const long long kc2 = clock64();
#elif defined(SLICE_DBG)
    if (lane == 0 && chain == 0) { for (int x = 0; x < 5; ++x) S.ctl->dbg[x] += scy[x]; S.ctl->dbg[5] += nev; S.ctl->dbg[6] += scy[5]; S.ctl->dbg[7] += ev2; }
#endif
    if (lane == 0) S.ch_nlike[chain] = C.nlike;
    if (graded) {
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < PC_MAX_GRADE; ++g) S.ch_nlike_g[(size_t)chain * PC_MAX_GRADE + g] = nl_grade[g];
        }
    }
    if (seq_mode && lane == 0 && chain == 0) S.ctl->seq = seq_run;
    // derived parameters of all the babies at once, lane = slice (gaussian.f90:36-37, twin_gaussian.f90:48-52):
    // one sqrt / log per chain instead of one per slice on the chain's critical path
    if (S.nDer > 0 && phi_lds) {
        __syncthreads();                            // one wave per workgroup: orders the LDS writes above
        for (int s0 = 0; s0 < nr; s0 += 64) {
            const int s = s0 + lane;
            if (s >= nr) continue;
            double *row = S.babies + ((size_t)chain * nr + s) * nT;
            const double *tt = tbuf + (size_t)s * (D + 1);
            double phi0 = 0.0, phi1 = 0.0;
            if (S.like.kind == PC_LIKE_GAUSSIAN) {
                double r2 = 0.0;
                for (int d = 0; d < D; ++d) { const double z = tt[d] - S.like.mu; r2 += z * z; }
                phi0 = sqrt(r2);
                if (S.nDer >= 2) phi1 = pc_log_ball(phi0, D, S.like.log_vn);
            } else if (S.like.kind == PC_LIKE_TWIN_GAUSSIAN) {
                phi0 = (tt[0] > 0.5) ? 1.0 : -1.0;
            }
            row[S.d0] = phi0;
            if (S.nDer >= 2) row[S.d0 + 1] = phi1;
            for (int e = 2; e < S.nDer; ++e) row[S.d0 + e] = 0.0;
        }
    }
#if defined(SLICE_DBG) && SLICE_DBG == 2
    // whole-kernel view, all chains: cycles before / inside / after the slice loop (sums and maxima), launch-relative start
    {
        const long long kc3 = clock64(), kw3 = wall_clock64();
        unsigned long long *g = (unsigned long long *)S.ctl->dbg;
        if (lane == 0) {
            atomicAdd(&g[0], (unsigned long long)(kp1 - kc0)); atomicAdd(&g[1], (unsigned long long)(kp2 - kp1)); atomicAdd(&g[2], (unsigned long long)(kp3 - kp2));
            atomicAdd(&g[3], (unsigned long long)(kc1 - kp3)); atomicAdd(&g[4], (unsigned long long)(kc2 - kc1)); atomicAdd(&g[5], (unsigned long long)(kc3 - kc2));
            atomicMax(&g[6], (unsigned long long)(kw3 - kw0));       // longest chain, 100 MHz ticks
        }
    }
#endif
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
            static size_t doneq = 0;
            if (shq > doneq) { (void)hipFuncSetAttribute((const void *)k_nhats_q<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shq); doneq = shq; }
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

extern "C" int pc_launch_slice_fused(const PcState *S, unsigned batch, int nchains, hipStream_t st)
{
    if (!pc_slice_fusable(S)) return 1;
    const size_t sh0 = sizeof(double) * ((size_t)S->D + S->nr) + 16;
    const size_t tb = sizeof(double) * (size_t)S->nr * (S->D + 1);
    const int phi_lds = (S->nDer > 0 && sh0 + tb <= 48 * 1024) ? 1 : 0;
    const int D = S->D, FWv = D <= 8 ? 8 : (D <= 16 ? 16 : 24);
    const size_t sh = sh0 + (phi_lds ? tb : 0) + sizeof(double) * ((size_t)FWv * D + (size_t)S->nr * (D + 2));   // + L, directions, widths
    if (sh > 150 * 1024) return 1;
#define PC_SLICE_FUSED(NROWS, FW) { \
        if (sh > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_slice<1, NROWS, false, 1, FW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((k_slice<1, NROWS, false, 1, FW>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, 0); }
    if (D <= 8) PC_SLICE_FUSED(1, 8)
    else if (D <= 16) PC_SLICE_FUSED(1, 16)
    else PC_SLICE_FUSED(2, 24)
#undef PC_SLICE_FUSED
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
        if (sh4 > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_slice<2, 4, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
        hipLaunchKernelGGL((k_slice<2, 4, false, 4>), dim3(nchains / 4), dim3(256), sh4, st, *S, batch, phi_lds, mat_lds);
        return 0;
    }
    if (mat_lds) sh += mb;
#define PC_SLICE_LAUNCH1(DPL, NROWS, GR) { \
        if (sh > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_slice<DPL, NROWS, GR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((k_slice<DPL, NROWS, GR>), dim3(nchains), dim3(64), sh, st, *S, batch, phi_lds, mat_lds); }
#define PC_SLICE_LAUNCH(DPL, NROWS) { if (S->ngrade > 1 || S->seq_mode) PC_SLICE_LAUNCH1(DPL, NROWS, true) else PC_SLICE_LAUNCH1(DPL, NROWS, false) }
    if (D <= 16) PC_SLICE_LAUNCH(1, 1)
    else if (D <= 32) PC_SLICE_LAUNCH(1, 2)
    else if (D <= 64) PC_SLICE_LAUNCH(1, 4)
    else if (D <= 128) PC_SLICE_LAUNCH(2, 4)
    else if (D <= 256) PC_SLICE_LAUNCH(4, 4)
    else return 1;
#undef PC_SLICE_LAUNCH
#undef PC_SLICE_LAUNCH1
    return 0;
}
