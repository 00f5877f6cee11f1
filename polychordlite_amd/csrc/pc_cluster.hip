// pc_cluster.hip -- kNN clustering of the live points on the device.
//
// Restates do_clustering / NN_clustering / compute_knn / do_clustering_k / relabel
// (src/polychord/clustering.f90:15-324, utils.F90:713-749) and the data-parallel parts of
// add_cluster (run_time_info.f90:303-505):
//   k_similarity   S_ab = r_a + r_b - 2 x_a.x_b over the cube coordinates (calculate.f90:94-109),
//                  one thread per pair, products and sums kept un-fused in dimension order so that the
//                  neighbour ORDER equals the reference's;
//   k_knn_sort     compute_knn for every row of a (sub)set at once: the insertion rule of
//                  clustering.f90:156-172 (first slot with a strictly larger distance) is a stable sort
//                  by (distance, index); one workgroup sorts one row in LDS (bitonic), which also makes
//                  the "double k and recompute" step of NN_clustering free;
//   k_nn_cluster   one workgroup runs the n = 2..k loop of NN_clustering (:53-76): connected components
//                  of the "i in j's n-list or j in i's" graph by min-label hooking + pointer jumping in
//                  LDS, relabel by first appearance, early exits;
//   k_rebuild_lists / k_cluster_stats / k_ph_rehome   the array side of add_cluster: list order from
//                  (cluster, position) labels, per-cluster contour + live log-sum-exp, and phantoms
//                  re-homed to the cluster of their nearest live point (identify_cluster,
//                  run_time_info.f90:444-453, 913-949).
// The recursion over found clusters (clustering.f90:80-95) and the O(ncluster) evidence split
// (run_time_info.f90:458-503) are driven from the host (pc_engine.hip); they touch a few integers.
#include "pc_state.h"
#include <cstdlib>

__device__ __forceinline__ void similarity_body(const PcState &S, const int *pts /* slots in list order */, int n, double *Sm, int ybase, int ystride)
{
    const int a = blockIdx.x, D = S.D;
    const double *xa = S.live + (size_t)pts[a] * S.nT;
    double ra = 0.0;
    for (int d = 0; d < D; ++d) ra = __dadd_rn(ra, __dmul_rn(xa[d], xa[d]));
    for (int b = ybase + threadIdx.x; b < n; b += ystride) {
        const double *xb = S.live + (size_t)pts[b] * S.nT;
        double rb = 0.0, s = 0.0;
        for (int d = 0; d < D; ++d) { rb = __dadd_rn(rb, __dmul_rn(xb[d], xb[d])); s = __dadd_rn(s, __dmul_rn(xa[d], xb[d])); }
        Sm[(size_t)a * n + b] = __dadd_rn(__dadd_rn(ra, rb), -__dmul_rn(2.0, s));
    }
}

__global__ __launch_bounds__(256) void k_similarity(PcState S, const int *pts, int n, double *Sm)
{
    similarity_body(S, pts, n, Sm, blockIdx.y * 256, gridDim.y * 256);
}
// One descriptor per cluster that is looked at in an update: {cluster, points, offset of its n x n blocks, offset of its labels}.
// The first pass of do_clustering (clustering.f90:253-324) over ALL clusters of an update is three launches with the cluster
// in blockIdx.y instead of three launches and two host round trips per cluster (70 clusters, 130 updates per run at
// BASELINE configs[2]); only a cluster in which the pass finds a split goes through the per-cluster path, with its recursion.
struct ClusDesc { int c, n, off2, off1; };
__global__ __launch_bounds__(256) void k_similarity_b(PcState S, const ClusDesc *desc, double *Sm)
{
    const ClusDesc d = desc[blockIdx.y];
    if ((int)blockIdx.x >= d.n) return;
    similarity_body(S, S.cl_list + (size_t)d.c * S.Ncap, d.n, Sm + d.off2, 0, 256);
}

// rows of the sub-matrix S(gidx, gidx) sorted by (distance, local index): knn[a*m + r] = r-th neighbour
__device__ __forceinline__ void knn_sort_body(const double *Sm, int nroot, const int *gidx /* null: identity */, int m, int npow2, int *knn)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *kv = (double *)smem;
    int *ki = (int *)(kv + npow2);
    const int a = blockIdx.x, tid = threadIdx.x;
    const double *row = Sm + (size_t)(gidx ? gidx[a] : a) * nroot;
    for (int i = tid; i < npow2; i += 256) { kv[i] = (i < m) ? row[gidx ? gidx[i] : i] : PC_HUGE; ki[i] = (i < m) ? i : 0x7fffffff; }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const double x = kv[i], y = kv[l]; const int p = ki[i], q = ki[l];
                    const bool gt = (x > y) || (x == y && p > q);
                    if (gt == up) { kv[i] = y; kv[l] = x; ki[i] = q; ki[l] = p; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < m; i += 256) knn[(size_t)a * m + i] = ki[i];
}
__global__ __launch_bounds__(256) void k_knn_sort(const double *Sm, int nroot, const int *gidx, int m, int npow2, int *knn)
{
    knn_sort_body(Sm, nroot, gidx, m, npow2, knn);
}
__global__ __launch_bounds__(256) void k_knn_sort_b(const double *Sm, const ClusDesc *desc, int *knn)
{
    const ClusDesc d = desc[blockIdx.y];
    if ((int)blockIdx.x >= d.n) return;
    int npow2 = 2;
    while (npow2 < d.n) npow2 <<= 1;
    knn_sort_body(Sm + d.off2, d.n, nullptr, d.n, npow2, knn + d.off2);
}

// NN_clustering main loop for one point set (no recursion).  out[0] = number of clusters.
__device__ __forceinline__ void nn_cluster_body(const int *knn, int m, int *labels_out, int *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lab = (int *)smem;          // [m] component label (min index), then 1-based cluster label
    int *old = lab + m;              // [m] previous clustering
    int *rank = old + m;             // [m] scratch for the relabel prefix sums
    __shared__ int sh_changed, sh_num, sh_same, sh_carry;
    const int tid = threadIdx.x;
    auto root = [&](int x) { while (lab[x] != x) x = lab[x]; return x; };
    int k = m < 10 ? m : 10;
    const int k0 = k;
    int num = m;
    for (int i = tid; i < m; i += 1024) { old[i] = i + 1; labels_out[i] = i + 1; }
    __syncthreads();
    for (int nn = 2; nn <= k0; ++nn) {
        // ---- do_clustering_k (clustering.f90:100-130): components of the neighbour graph
        for (int i = tid; i < m; i += 1024) lab[i] = i;
        __syncthreads();
        while (true) {
            if (tid == 0) sh_changed = 0;
            __syncthreads();
            for (int e = tid; e < m * nn; e += 1024) {
                const int i = e / nn, mm = e % nn;
                const int j = knn[(size_t)i * m + mm];
                // neighbours(): j's first entry (itself, clustering.f90:186) appears in i's list
                if (knn[(size_t)j * m] != j) continue;         // duplicate points: handled below
                const int ra = root(i), rb = root(j);
                if (ra != rb) { atomicMin(&lab[ra > rb ? ra : rb], ra < rb ? ra : rb); sh_changed = 1; }
            }
            // points whose nearest entry is not themselves (exact duplicates): brute force
            for (int j = tid; j < m; j += 1024) {
                const int rep = knn[(size_t)j * m];
                if (rep == j) continue;
                for (int i = 0; i < m; ++i)
                    for (int mm = 0; mm < nn; ++mm)
                        if (knn[(size_t)i * m + mm] == rep) {
                            const int ra = root(i), rb = root(j);
                            if (ra != rb) { atomicMin(&lab[ra > rb ? ra : rb], ra < rb ? ra : rb); sh_changed = 1; }
                        }
            }
            __syncthreads();
            const int ch = sh_changed;
            __syncthreads();
            if (!ch) break;
        }
        for (int i = tid; i < m; i += 1024) rank[i] = root(i);
        __syncthreads();
        for (int i = tid; i < m; i += 1024) lab[i] = rank[i];
        __syncthreads();
        // ---- relabel (utils.F90:713-749): label = 1 + number of component minima below mine
        if (tid == 0) sh_carry = 0;
        __syncthreads();
        for (int base = 0; base < m; base += 1024) {
            const int i = base + tid;
            const int isroot = (i < m && lab[i] == i) ? 1 : 0;
            // block inclusive scan in `rank`
            __shared__ int scan[1024];
            scan[tid] = isroot;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) { const int v = tid >= off ? scan[tid - off] : 0; __syncthreads(); scan[tid] += v; __syncthreads(); }
            if (i < m) rank[i] = sh_carry + scan[tid];       // inclusive count of roots up to i
            __syncthreads();
            if (tid == 1023) sh_carry += scan[1023];
            __syncthreads();
        }
        num = sh_carry;
        if (tid == 0) sh_same = 1;
        __syncthreads();
        for (int i = tid; i < m; i += 1024) {
            const int l = rank[lab[i]];                      // 1-based label of my component's minimum
            labels_out[i] = l;
            if (l != old[i]) sh_same = 0;
        }
        __syncthreads();
        const int same = sh_same;
        if (num == 1) break;                                  // clustering.f90:62-63
        if (same) break;                                      // :64-65
        if (nn == k) k = (2 * k < m) ? 2 * k : m;             // :66-69 (the sorted lists already hold any k)
        for (int i = tid; i < m; i += 1024) old[i] = labels_out[i];
        __syncthreads();
    }
    if (tid == 0) { out[0] = num; sh_num = num; }
}
__global__ __launch_bounds__(1024) void k_nn_cluster(const int *knn, int m, int *labels_out, int *out)
{
    nn_cluster_body(knn, m, labels_out, out);
}
__global__ __launch_bounds__(1024) void k_nn_cluster_b(const int *knn, const ClusDesc *desc, int *labels, int *out)
{
    const ClusDesc d = desc[blockIdx.x];
    nn_cluster_body(knn + d.off2, d.n, labels + d.off1, out + blockIdx.x);
}

// cl_list / cl_n from the (cluster, position) labels of every slot
__global__ __launch_bounds__(256) void k_rebuild_lists(PcState S, int nc)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid < S.Ncap) { const int c = S.live_cluster[tid]; if (c >= 0) S.cl_list[(size_t)c * S.Ncap + S.live_pos[tid]] = tid; }
    (void)nc;
}

// per-cluster count, contour (find_min_loglikelihoods) and live log-sum-exp; one workgroup per cluster
__global__ __launch_bounds__(256) void k_cluster_stats(PcState S)
{
    const int c = blockIdx.x, tid = threadIdx.x;
    __shared__ double sv[256]; __shared__ int sk[256]; __shared__ int scount[256];
    double bv = PC_HUGE; int bk = 0x7fffffff, bs = -1, cnt = 0; double mx = -PC_HUGE;
    for (int s = tid; s < S.Ncap; s += 256) {
        if (S.live_cluster[s] != c) continue;
        cnt++;
        const double v = S.live_logL[s]; const int p = S.live_pos[s];
        if (v < bv || (v == bv && p < bk)) { bv = v; bk = p; bs = s; }
        mx = fmax(mx, v);
    }
    sv[tid] = bv; sk[tid] = bk; scount[tid] = cnt;
    __shared__ int ss[256]; __shared__ double smx[256];
    ss[tid] = bs; smx[tid] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            if (sv[tid + off] < sv[tid] || (sv[tid + off] == sv[tid] && sk[tid + off] < sk[tid])) { sv[tid] = sv[tid + off]; sk[tid] = sk[tid + off]; ss[tid] = ss[tid + off]; }
            scount[tid] += scount[tid + off]; smx[tid] = fmax(smx[tid], smx[tid + off]);
        }
        __syncthreads();
    }
    const double ref = smx[0];
    double sum = 0.0;
    for (int s = tid; s < S.Ncap; s += 256) if (S.live_cluster[s] == c) sum += exp(S.live_logL[s] - ref);
    __shared__ double ssum[256];
    ssum[tid] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) ssum[tid] += ssum[tid + off]; __syncthreads(); }
    if (tid == 0) {
        S.cl_n[c] = scount[0];
        S.logLp[c] = scount[0] > 0 ? sv[0] : PC_HUGE; S.imin_slot[c] = ss[0];
        S.lse_ref[c] = scount[0] > 0 ? ref : 0.0; S.lse_sum[c] = ssum[0];
    }
}

// every phantom goes to the cluster of its nearest live point and survives only above that cluster's
// contour (run_time_info.f90:444-453).  A workgroup takes 64 phantoms, four threads each; the live points pass through an
// LDS tile of 64 (a wave per phantom reading the live set from L2 took 0.6 ms per split at 20 k phantoms x 1000 points)
#define PHR_P 64
#define PHR_T 64
__global__ __launch_bounds__(256) void k_ph_rehome(PcState S, int nph, int nc, const unsigned *old_uids, int nold_uids)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, D = S.D, nT = S.nT, Ncap = S.Ncap;
    const int DP = D | 1;                                  // odd row stride: the four scanners of a phantom and the rows of a tile fall in different banks
    double *xs = (double *)smem;                           // [PHR_P][DP] phantoms of this block
    double *ys = xs + (size_t)PHR_P * DP;                  // [PHR_T][DP] tile of live points
    double *bd = ys + (size_t)PHR_T * DP;                  // [256] partial minima
    int *ykey = (int *)(bd + 256);                         // [PHR_T] cluster * Ncap + list position, or -1
    int *bk = ykey + PHR_T;                                // [256]
    int *alive = bk + 256;                                 // [PHR_P] the phantom's cluster still exists
    const int j0 = blockIdx.x * PHR_P;
    for (int e = tid; e < PHR_P * D; e += 256) { const int p = e / D, d = e % D, j = j0 + p; xs[(size_t)p * DP + d] = (j < nph) ? S.phantom[(size_t)j * nT + d] : 0.0; }
    if (tid < PHR_P) {
        // phantoms of clusters that no longer exist are gone (run_time_info.f90:367-368 saves live clusters only)
        const int j = j0 + tid;
        bool ok = false;
        if (j < nph) { const unsigned u = S.ph_cuid[j]; for (int q = 0; q < nold_uids; ++q) ok |= (old_uids[q] == u); }
        alive[tid] = ok ? 1 : 0;
    }
    const int p = tid >> 2, sc = tid & 3;
    double best = PC_HUGE; int bkey = 0x7fffffff;
    for (int t0 = 0; t0 < Ncap; t0 += PHR_T) {
        __syncthreads();
        for (int e = tid; e < PHR_T * D; e += 256) { const int q = e / D, d = e % D, s = t0 + q; ys[(size_t)q * DP + d] = (s < Ncap) ? S.live[(size_t)s * nT + d] : 0.0; }
        if (tid < PHR_T) { const int s = t0 + tid; const int c = (s < Ncap) ? S.live_cluster[s] : -1; ykey[tid] = (c >= 0) ? c * Ncap + S.live_pos[s] : -1; }
        __syncthreads();
        const double *x = xs + (size_t)p * DP;
        for (int q = sc; q < PHR_T; q += 4) {
            const int key = ykey[q];
            if (key < 0) continue;
            const double *y = ys + (size_t)q * DP;
            double d2 = 0.0;
            for (int d = 0; d < D; ++d) { const double t = x[d] - y[d]; d2 += t * t; }
            if (d2 < best || (d2 == best && key < bkey)) { best = d2; bkey = key; }
        }
    }
    bd[tid] = best; bk[tid] = bkey;
    __syncthreads();
    if (tid < PHR_P && j0 + tid < nph) {
        const int j = j0 + tid;
        double b = bd[4 * tid]; int k = bk[4 * tid];
        for (int u = 1; u < 4; ++u) { const double v = bd[4 * tid + u]; const int kk = bk[4 * tid + u]; if (v < b || (v == b && kk < k)) { b = v; k = kk; } }
        if (!alive[tid]) S.ph_cuid[j] = 0xFFFFFFFFu;
        else { const int c = k / Ncap; S.ph_cuid[j] = (S.ph_logL[j] > S.logLp[c]) ? S.cl_uid[c] : 0xFFFFFFFFu; }
    }
    (void)nc;
}

// The same for nDims <= 32 (round 5): lane = phantom with its coordinates in registers, two wavefronts to 64 phantoms that take every
// other live point of a tile -- all lanes read the SAME live point, one LDS broadcast per two coordinates.  The kernel above reads both
// operands of every difference from LDS (two reads per multiply-add: 150-190 us a split at configs[2], the LDS pipe's time; this one
// 35).  Rows are padded with zeros to DM coordinates on both sides: (0 - 0)^2 adds +0 to the same sum, term for term in the same order.
template <int DM>
__global__ __launch_bounds__(128) void k_ph_rehome_r(PcState S, int nph, const unsigned *old_uids, int nold_uids)
{
    constexpr int T = 128;
    __shared__ __attribute__((aligned(16))) double ys[T * DM];
    __shared__ int ykey[T];
    __shared__ double bd[64];
    __shared__ int bk[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, D = S.D, nT = S.nT, Ncap = S.Ncap;
    const int j = blockIdx.x * 64 + lane;
    double x[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) x[d] = (j < nph && d < D) ? S.phantom[(size_t)j * nT + d] : 0.0;
    double best = PC_HUGE; int bkey = 0x7fffffff;
    for (int t0 = 0; t0 < Ncap; t0 += T) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DM; ++u) {
            const int e = tid + 128 * u, q = e / DM, d = e - q * DM, s = t0 + q;
            ys[e] = (s < Ncap && d < D) ? S.live[(size_t)s * nT + d] : 0.0;
        }
        { const int s = t0 + tid; const int c = (s < Ncap) ? S.live_cluster[s] : -1; ykey[tid] = (c >= 0) ? c * Ncap + S.live_pos[s] : -1; }
        __syncthreads();
#pragma unroll 2
        for (int q = wv; q < T; q += 2) {
            const int key = ykey[q];
            if (key < 0) continue;
            const double *y = ys + q * DM;
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < DM; ++d) { const double t = x[d] - y[d]; d2 += t * t; }
            if (d2 < best || (d2 == best && key < bkey)) { best = d2; bkey = key; }
        }
    }
    if (wv == 1) { bd[lane] = best; bk[lane] = bkey; }
    __syncthreads();
    if (wv == 0 && j < nph) {
        const double v = bd[lane]; const int kk = bk[lane];
        if (v < best || (v == best && kk < bkey)) { best = v; bkey = kk; }
        // phantoms of clusters that no longer exist are gone (run_time_info.f90:367-368 saves live clusters only)
        const unsigned u = S.ph_cuid[j];
        bool ok = false;
        for (int q = 0; q < nold_uids; ++q) ok |= (old_uids[q] == u);
        if (!ok) S.ph_cuid[j] = 0xFFFFFFFFu;
        else { const int c = bkey / Ncap; S.ph_cuid[j] = (S.ph_logL[j] > S.logLp[c]) ? S.cl_uid[c] : 0xFFFFFFFFu; }
    }
}

// number of phantoms per cluster uid (fixed order: one workgroup, serial accumulation per cluster)
__global__ __launch_bounds__(256) void k_ph_count(PcState S, int nph, int nc, int *counts)
{
    __shared__ int sc[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const unsigned uid = S.cl_uid[c];
    int cnt = 0;
    for (int j = tid; j < nph; j += 256) cnt += (S.ph_cuid[j] == uid);
    sc[tid] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) sc[tid] += sc[tid + off]; __syncthreads(); }
    if (tid == 0) counts[c] = sc[0];
    (void)nc;
}

// add_cluster: the Cholesky factors and covariances of the clusters behind cluster p move up a block (run_time_info.f90:371-376); one
// workgroup, block after block (the blocks overlap in the order of the move)
__global__ __launch_bounds__(256) void k_shift_mats(PcState S, int p, int nc)
{
    const int DD = S.D * S.D;
    for (int c = p; c < nc - 1; ++c) {
        for (int e = threadIdx.x; e < DD; e += 256) {
            S.chol[(size_t)c * DD + e] = S.chol[(size_t)(c + 1) * DD + e];
            S.cov[(size_t)c * DD + e] = S.cov[(size_t)(c + 1) * DD + e];
        }
        __syncthreads();
    }
}

// the first pass for several runs in step: blockIdx.z = run, every run its own descriptors and scratch (PcManyRec::p: 0 descriptors,
// 1 similarity blocks, 2 neighbour lists, 3 labels, 4 verdicts; ia[1] = clusters looked at)
__global__ __launch_bounds__(256) void k_similarity_b_many(const PcManyRec *__restrict__ R)
{
    const PcManyView r = pc_many_view(R, blockIdx.z);
    if ((int)blockIdx.y >= r.ia[1]) return;
    const ClusDesc d = ((const ClusDesc *)r.p[0])[blockIdx.y];
    if ((int)blockIdx.x >= d.n) return;
    similarity_body(r.S, r.S.cl_list + (size_t)d.c * r.S.Ncap, d.n, (double *)r.p[1] + d.off2, 0, 256);
}
__global__ __launch_bounds__(256) void k_knn_sort_b_many(const PcManyRec *__restrict__ R)
{
    const PcManyView r = pc_many_view(R, blockIdx.z);
    if ((int)blockIdx.y >= r.ia[1]) return;
    const ClusDesc d = ((const ClusDesc *)r.p[0])[blockIdx.y];
    if ((int)blockIdx.x >= d.n) return;
    int npow2 = 2;
    while (npow2 < d.n) npow2 <<= 1;
    knn_sort_body((const double *)r.p[1] + d.off2, d.n, nullptr, d.n, npow2, (int *)r.p[2] + d.off2);
}
__global__ __launch_bounds__(1024) void k_nn_cluster_b_many(const PcManyRec *__restrict__ R)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    if ((int)blockIdx.x >= r.ia[1]) return;
    const ClusDesc d = ((const ClusDesc *)r.p[0])[blockIdx.x];
    nn_cluster_body((const int *)r.p[2] + d.off2, d.n, (int *)r.p[3] + d.off1, (int *)r.p[4] + blockIdx.x);
}

// One level of NN_clustering's recursion for many parts at once (Engine::refine_partitions): part b = m points of a cluster, given by
// their positions in the cluster's point order (pool + ioff), clustered on the sub-matrix of the cluster's similarity block the first
// pass left at Sm + off2 (n x n).  Descriptor: {off2, n, ioff, m, koff}: its neighbour lists go to knn + koff (m x m), its labels to
// labels + ioff, the number of clusters found to out[b].
struct SubDesc { int off2, n, ioff, m, koff; };
__global__ __launch_bounds__(256) void k_knn_sort_sub(const SubDesc *desc, const double *Sm, const int *pool, int *knn)
{
    const SubDesc d = desc[blockIdx.y];
    if ((int)blockIdx.x >= d.m) return;
    int npow2 = 2;
    while (npow2 < d.m) npow2 <<= 1;
    knn_sort_body(Sm + d.off2, d.n, pool + d.ioff, d.m, npow2, knn + d.koff);
}
__global__ __launch_bounds__(1024) void k_nn_cluster_sub(const SubDesc *desc, const int *knn, int *labels, int *out)
{
    const SubDesc d = desc[blockIdx.x];
    nn_cluster_body(knn + d.koff, d.m, labels + d.ioff, out + blockIdx.x);
}
// ... for several runs in step (PcManyRec::p: 0 descriptors, 1 similarity blocks, 2 pool, 3 neighbour lists, 4 labels, 5 verdicts; ia[1] parts)
__global__ __launch_bounds__(256) void k_knn_sort_sub_many(const PcManyRec *__restrict__ R)
{
    const PcManyView r = pc_many_view(R, blockIdx.z);
    if ((int)blockIdx.y >= r.ia[1]) return;
    const SubDesc d = ((const SubDesc *)r.p[0])[blockIdx.y];
    if ((int)blockIdx.x >= d.m) return;
    int npow2 = 2;
    while (npow2 < d.m) npow2 <<= 1;
    knn_sort_body((const double *)r.p[1] + d.off2, d.n, (const int *)r.p[2] + d.ioff, d.m, npow2, (int *)r.p[3] + d.koff);
}
__global__ __launch_bounds__(1024) void k_nn_cluster_sub_many(const PcManyRec *__restrict__ R)
{
    const PcManyView r = pc_many_view(R, blockIdx.y);
    if ((int)blockIdx.x >= r.ia[1]) return;
    const SubDesc d = ((const SubDesc *)r.p[0])[blockIdx.x];
    nn_cluster_body((const int *)r.p[3] + d.koff, d.m, (int *)r.p[4] + d.ioff, (int *)r.p[5] + blockIdx.x);
}

extern "C" {

static int sub_lds(int mmax, size_t &sh, size_t &sh2)
{
    int npow2 = 2;
    while (npow2 < mmax) npow2 <<= 1;
    sh = (size_t)npow2 * 12; sh2 = (size_t)mmax * 12 + 64;
    return (sh > 160 * 1024 || sh2 > 150 * 1024) ? 1 : 0;
}
int pc_launch_knn_cluster_sub(const int *d_desc, int nb, int mmax, const double *Sm, const int *pool, int *knn, int *labels, int *out, hipStream_t st)
{
    if (nb <= 0) return 0;
    size_t sh, sh2;
    if (sub_lds(mmax, sh, sh2)) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort_sub, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster_sub, sh2);
    hipLaunchKernelGGL(k_knn_sort_sub, dim3(mmax, nb), dim3(256), sh, st, (const SubDesc *)d_desc, Sm, pool, knn);
    hipLaunchKernelGGL(k_nn_cluster_sub, dim3(nb), dim3(1024), sh2, st, (const SubDesc *)d_desc, (const int *)knn, labels, out);
    return 0;
}
int pc_launch_knn_cluster_sub_many(const PcManyRec *dR, int R, int nb_max, int mmax, hipStream_t st)
{
    if (nb_max <= 0) return 0;
    size_t sh, sh2;
    if (sub_lds(mmax, sh, sh2)) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort_sub_many, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster_sub_many, sh2);
    hipLaunchKernelGGL(k_knn_sort_sub_many, dim3(mmax, nb_max, R), dim3(256), sh, st, dR);
    hipLaunchKernelGGL(k_nn_cluster_sub_many, dim3(nb_max, R), dim3(1024), sh2, st, dR);
    return 0;
}

// The chains still in the nursery follow the list of clusters through an update that split some of them (settings.epoch_discard
// = 0; oracle: remap_chains): map[c] = the place of the update's cluster c in the new list, -1 if it was split -- its chains are lost.
__global__ __launch_bounds__(256) void k_remap_chains(PcState S, const int *map, int nold, int n)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= n) return;
    const int c = S.ch_cluster[w];
    const int m = (c >= 0 && c < nold) ? map[c] : -1;
    S.ch_cluster[w] = m;
    if (m < 0) S.ch_epoch[w] = -1;
}
void pc_launch_remap_chains(const PcState *S, const int *map, int nold, int n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_remap_chains, dim3((n + 255) / 256), dim3(256), 0, st, *S, map, nold, n);
}

void pc_launch_shift_mats(const PcState *S, int p, int nc, hipStream_t st)
{
    if (p < nc - 1) hipLaunchKernelGGL(k_shift_mats, dim3(1), dim3(256), 0, st, *S, p, nc);
}

void pc_launch_similarity(const PcState *S, const int *pts, int n, double *Sm, hipStream_t st)
{
    hipLaunchKernelGGL(k_similarity, dim3(n, (n + 1023) / 1024), dim3(256), 0, st, *S, pts, n, Sm);
}

int pc_launch_knn_cluster(const double *Sm, int nroot, const int *gidx, int m, int *knn, int *labels, int *out, hipStream_t st)
{
    int npow2 = 2;
    while (npow2 < m) npow2 <<= 1;
    const size_t sh = (size_t)npow2 * 12;
    const size_t sh2 = (size_t)m * 12 + 64;
    if (sh > 160 * 1024 || sh2 > 150 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster, sh2);
    hipLaunchKernelGGL(k_knn_sort, dim3(m), dim3(256), sh, st, Sm, nroot, gidx, m, npow2, knn);
    hipLaunchKernelGGL(k_nn_cluster, dim3(1), dim3(1024), sh2, st, knn, m, labels, out);
    return 0;
}

// first pass over `nd` clusters at once (descriptors on the device, host copy for the grid): out[k] = clusters found in the k-th
int pc_launch_knn_cluster_batch(const PcState *S, const int *h_desc, const int *d_desc, int nd, double *Sm, int *knn, int *labels, int *out, hipStream_t st)
{
    if (nd <= 0) return 0;
    int nmax = 0;
    for (int k = 0; k < nd; ++k) nmax = h_desc[4 * k + 1] > nmax ? h_desc[4 * k + 1] : nmax;
    int npow2 = 2;
    while (npow2 < nmax) npow2 <<= 1;
    const size_t sh = (size_t)npow2 * 12, sh2 = (size_t)nmax * 12 + 64;
    if (sh > 160 * 1024 || sh2 > 150 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort_b, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster_b, sh2);
    const ClusDesc *dd = (const ClusDesc *)d_desc;
    hipLaunchKernelGGL(k_similarity_b, dim3(nmax, nd), dim3(256), 0, st, *S, dd, Sm);
    hipLaunchKernelGGL(k_knn_sort_b, dim3(nmax, nd), dim3(256), sh, st, (const double *)Sm, dd, knn);
    hipLaunchKernelGGL(k_nn_cluster_b, dim3(nd), dim3(1024), sh2, st, (const int *)knn, dd, labels, out);
    return 0;
}

// the same with the largest cluster given (no host copy of the descriptors at hand: a record of the runs in step launched on its own)
int pc_launch_knn_cluster_batch_dev(const PcState *S, const int *d_desc, int nd, int nmax, double *Sm, int *knn, int *labels, int *out, hipStream_t st)
{
    if (nd <= 0) return 0;
    int npow2 = 2;
    while (npow2 < nmax) npow2 <<= 1;
    const size_t sh = (size_t)npow2 * 12, sh2 = (size_t)nmax * 12 + 64;
    if (sh > 160 * 1024 || sh2 > 150 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort_b, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster_b, sh2);
    const ClusDesc *dd = (const ClusDesc *)d_desc;
    hipLaunchKernelGGL(k_similarity_b, dim3(nmax, nd), dim3(256), 0, st, *S, dd, Sm);
    hipLaunchKernelGGL(k_knn_sort_b, dim3(nmax, nd), dim3(256), sh, st, (const double *)Sm, dd, knn);
    hipLaunchKernelGGL(k_nn_cluster_b, dim3(nd), dim3(1024), sh2, st, (const int *)knn, dd, labels, out);
    return 0;
}
int pc_launch_knn_cluster_batch_many(const PcState *S, const PcManyRec *dR, int R, int nd_max, int nmax, hipStream_t st)
{
    (void)S;
    if (nd_max <= 0) return 0;
    int npow2 = 2;
    while (npow2 < nmax) npow2 <<= 1;
    const size_t sh = (size_t)npow2 * 12, sh2 = (size_t)nmax * 12 + 64;
    if (sh > 160 * 1024 || sh2 > 150 * 1024) return 1;
    pc_need_dyn_lds((const void *)k_knn_sort_b_many, sh);
    pc_need_dyn_lds((const void *)k_nn_cluster_b_many, sh2);
    hipLaunchKernelGGL(k_similarity_b_many, dim3(nmax, nd_max, R), dim3(256), 0, st, dR);
    hipLaunchKernelGGL(k_knn_sort_b_many, dim3(nmax, nd_max, R), dim3(256), sh, st, dR);
    hipLaunchKernelGGL(k_nn_cluster_b_many, dim3(nd_max, R), dim3(1024), sh2, st, dR);
    return 0;
}

void pc_launch_rebuild(const PcState *S, int nc, hipStream_t st)
{
    const int n = S->Ncap > S->maxc ? S->Ncap : S->maxc;
    hipLaunchKernelGGL(k_rebuild_lists, dim3((n + 255) / 256), dim3(256), 0, st, *S, nc);
    hipLaunchKernelGGL(k_cluster_stats, dim3(nc), dim3(256), 0, st, *S);
}

void pc_launch_ph_rehome(const PcState *S, int nph, int nc, const unsigned *old_uids, int nold_uids, int *counts, hipStream_t st)
{
    static const bool general_only = std::getenv("PC_PH_REHOME_GENERAL") != nullptr;      // (A/B and the test that holds the two kernels together)
    if (nph > 0 && S->D <= 32 && !general_only) {
        const dim3 g((nph + 63) / 64), b(128);
        switch ((S->D + 3) / 4) {
#define PC_R(n) case n: hipLaunchKernelGGL((k_ph_rehome_r<4 * n>), g, b, 0, st, *S, nph, old_uids, nold_uids); break;
            PC_R(1) PC_R(2) PC_R(3) PC_R(4) PC_R(5) PC_R(6) PC_R(7) PC_R(8)
#undef PC_R
        }
    } else if (nph > 0) {
        const int DP = S->D | 1;
        const size_t sh = sizeof(double) * ((size_t)(PHR_P + PHR_T) * DP + 256) + sizeof(int) * (PHR_T + 256 + PHR_P);
        pc_need_dyn_lds((const void *)k_ph_rehome, sh);
        hipLaunchKernelGGL(k_ph_rehome, dim3((nph + PHR_P - 1) / PHR_P), dim3(256), sh, st, *S, nph, nc, old_uids, nold_uids);
    }
    hipLaunchKernelGGL(k_ph_count, dim3(nc), dim3(256), 0, st, *S, nph, nc, counts);
}

}  // extern "C"
