// pc_merge.hip -- the exchange step of the repeat-sharded path: evidence and posterior of the UNION of
// independent nested-sampling runs, on the device.
//
// The reference's MPI farm (nested_sampling.F90:262-301, mpi_utils.F90:376-463) gathers the babies of its workers into
// ONE run.  This engine shards at the run level instead: every GPU carries a complete run of its own (own seed), the
// dead points of all runs are gathered (RCCL all-gather between processes, or straight from the runs' result buffers
// inside one process), and the union is itself a valid nested-sampling run with n(L) = sum over the runs of their
// live points at contour L.  The evidence recursion of update_evidence (run_time_info.f90:211-296) and the
// log-normal estimate of calculate_logZ_estimate (:652-678) are evaluated over the merged death sequence; posterior
// weights are logX_i - log(n_i + 1) + logL_i as in the reference's posterior stack (calculate.f90:53-79).
//
// Nothing here is a sort: every run's deaths already ascend in logL, so
//   * rank of a record in the union = its own index + binary searches in the other runs' logL columns (R-way merge path);
//   * live points of run q just before its k-th death, G_q[k] = #{entry contour < L_k} - k, from a histogram of where
//     each point's entry contour sits in the run's own death sequence (an entry contour IS the logL of a later death of
//     the same run, or logzero for the initial points) + one prefix sum;
//   * n_i = sum_q G_q[rank of record i among run q's deaths];
//   * the recursion over the merged sequence = two device-wide inclusive scans (plain sums of log n/(n+1), log n/(n+2);
//     log-sum-exp pairs for <Z X>/X) + two log-sum-exp reductions, all with a fixed chunking (bit-reproducible).
#include "pc_state.h"
#include "pc_keys.h"
#include "../../include/polychord_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>
#include <thread>
#include <atomic>
#include <algorithm>
#include <mutex>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the library is dlopen'ed (no link-time dependency on RCCL)

extern "C" int pc_run_many(const pchip_settings *, const pchip_like *, const pchip_prior *, int, const int *, int, int, pchip_result *);
extern "C" void pc_prepare_streams(int dev, int want);

namespace {

struct P2 { double a, b; };
struct OpAdd {
    static __device__ __forceinline__ P2 id() { return P2{0.0, 0.0}; }
    static __device__ __forceinline__ P2 comb(const P2 &x, const P2 &y) { return P2{x.a + y.a, x.b + y.b}; }   // x earlier, y later
};
struct OpLse {      // (m, s) stands for m + log s; neutral (NEGBIG, 0)
    static __device__ __forceinline__ P2 id() { return P2{NEGBIG, 0.0}; }
    static __device__ __forceinline__ P2 comb(const P2 &x, const P2 &y)
    {
        const double e = exp(-fabs(x.a - y.a));
        return P2{fmax(x.a, y.a), (x.a >= y.a) ? x.b + y.b * e : x.b * e + y.b};
    }
};
__device__ __forceinline__ double lsv(const P2 &p) { return p.b > 0.0 ? p.a + log(p.b) : NEGBIG; }

#define SC_NT 256
#define SC_IT 8
#define SC_CHUNK (SC_NT * SC_IT)

// phase 1: inclusive scan inside chunks of SC_CHUNK elements, chunk totals out
template <class Op>
__global__ __launch_bounds__(SC_NT) void k_scan_local(P2 *d, long long n, P2 *tot)
{
    __shared__ P2 sh[SC_NT];
    const int tid = threadIdx.x;
    const long long base = (long long)blockIdx.x * SC_CHUNK + (long long)tid * SC_IT;
    P2 v[SC_IT];
#pragma unroll
    for (int u = 0; u < SC_IT; ++u) v[u] = (base + u < n) ? d[base + u] : Op::id();
#pragma unroll
    for (int u = 1; u < SC_IT; ++u) v[u] = Op::comb(v[u - 1], v[u]);
    sh[tid] = v[SC_IT - 1];
    __syncthreads();
    for (int k = 1; k < SC_NT; k <<= 1) {          // Hillis-Steele over the 256 thread totals
        P2 o = Op::id();
        if (tid >= k) o = sh[tid - k];
        __syncthreads();
        if (tid >= k) sh[tid] = Op::comb(o, sh[tid]);
        __syncthreads();
    }
    if (tid > 0) {
        const P2 pre = sh[tid - 1];
#pragma unroll
        for (int u = 0; u < SC_IT; ++u) v[u] = Op::comb(pre, v[u]);
    }
#pragma unroll
    for (int u = 0; u < SC_IT; ++u) if (base + u < n) d[base + u] = v[u];
    if (tid == SC_NT - 1) tot[blockIdx.x] = sh[SC_NT - 1];
}
// phase 2: exclusive scan of the chunk totals, one workgroup, carry between groups of 256
template <class Op>
__global__ __launch_bounds__(SC_NT) void k_scan_totals(P2 *tot, int nb)
{
    __shared__ P2 sh[SC_NT];
    __shared__ P2 carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = Op::id();
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += SC_NT) {
        const int b = b0 + tid;
        sh[tid] = (b < nb) ? tot[b] : Op::id();
        __syncthreads();
        for (int k = 1; k < SC_NT; k <<= 1) {
            P2 o = Op::id();
            if (tid >= k) o = sh[tid - k];
            __syncthreads();
            if (tid >= k) sh[tid] = Op::comb(o, sh[tid]);
            __syncthreads();
        }
        const P2 c = carry;
        const P2 ex = tid ? Op::comb(c, sh[tid - 1]) : c;
        const P2 all = Op::comb(c, sh[SC_NT - 1]);
        __syncthreads();
        if (b < nb) tot[b] = ex;
        if (tid == 0) carry = all;
        __syncthreads();
    }
}
// phase 3: chunk prefix onto every element
template <class Op>
__global__ __launch_bounds__(SC_NT) void k_scan_apply(P2 *d, long long n, const P2 *tot)
{
    if (blockIdx.x == 0) return;
    const P2 pre = tot[blockIdx.x];
    const long long base = (long long)blockIdx.x * SC_CHUNK + threadIdx.x;
#pragma unroll
    for (int u = 0; u < SC_IT; ++u) { const long long i = base + (long long)u * SC_NT; if (i < n) d[i] = Op::comb(pre, d[i]); }
}
template <class Op> void device_scan(P2 *d, long long n, P2 *tot, hipStream_t st)
{
    const int nb = (int)((n + SC_CHUNK - 1) / SC_CHUNK);
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_scan_local<Op>, dim3(nb), dim3(SC_NT), 0, st, d, n, tot);
    if (nb > 1) {
        hipLaunchKernelGGL(k_scan_totals<Op>, dim3(1), dim3(SC_NT), 0, st, tot, nb);
        hipLaunchKernelGGL(k_scan_apply<Op>, dim3(nb), dim3(SC_NT), 0, st, d, n, tot);
    }
}

struct MergeDev {
    int R, nT, l0;
    long long n;
    const long long *off;        // [R+1] first record of each run
    const double *rows;          // [n][nT], runs one after the other, each ascending in logL
    const double *entry;         // [n]
    double *L;                   // [n] logL column, contiguous
    int *G;                      // [n + R] per run len+1 entries at off[q] + q: histogram, then live points before death k
    long long *perm;             // [n] merged position -> record
    double *Ls;                  // [n] merged logL
    int *nl;                     // [n] live points just before each merged death
};

__device__ __forceinline__ int run_of(const MergeDev &M, long long g)
{
    int q = 0;
    for (int r = 1; r < M.R; ++r) q = (g >= M.off[r]) ? r : q;
    return q;
}
__device__ __forceinline__ long long lower_bound_d(const double *a, long long len, double x)
{   // number of elements < x
    long long lo = 0, hi = len;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ long long upper_bound_d(const double *a, long long len, double x)
{   // number of elements <= x
    long long lo = 0, hi = len;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void k_merge_extract(MergeDev M)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < M.n) M.L[g] = M.rows[(size_t)g * M.nT + M.l0];
    if (g < M.n + M.R) M.G[g] = 0;
}
// where does each point's entry contour sit in its run's death sequence?
__global__ void k_merge_hist(MergeDev M)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M.n) return;
    const int q = run_of(M, g);
    const long long o = M.off[q], len = M.off[q + 1] - o;
    const double e = M.entry[g];
    long long pos = lower_bound_d(M.L + o, len, e);
    if (pos < len && M.L[o + pos] == e) pos++;       // alive from the death AFTER the one whose logL is its entry contour
    atomicAdd(&M.G[o + q + pos], 1);
}
// inclusive prefix of the histogram minus the deaths so far = live points before death k; one workgroup per run
__global__ __launch_bounds__(1024) void k_merge_livecount(MergeDev M)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long o = M.off[q] + q, len1 = M.off[q + 1] - M.off[q] + 1;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (long long b0 = 0; b0 < len1; b0 += 1024) {
        const long long k = b0 + tid;
        int v = (k < len1) ? M.G[o + k] : 0;
        for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(v, s); if (lane >= s) v += t; }
        if (lane == 63) wsum[wv] = v;
        __syncthreads();
        int pre = carry;
        for (int x = 0; x < wv; ++x) pre += wsum[x];
        v += pre;
        __syncthreads();
        if (k < len1) M.G[o + k] = v - (int)k;
        if (tid == 1023) carry = v;
        __syncthreads();
    }
}
// position of every record in the union and the live points of all runs just before it dies
__global__ void k_merge_rank(MergeDev M)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M.n) return;
    const int r = run_of(M, g);
    const long long a = g - M.off[r];
    const double x = M.L[g];
    long long rank = a;
    int nlive = M.G[M.off[r] + r + a];
    for (int q = 0; q < M.R; ++q) {
        if (q == r) continue;
        const long long o = M.off[q], len = M.off[q + 1] - o;
        // equal logL in two runs: the run with the lower number dies first (a strict total order)
        const long long k = (q < r) ? upper_bound_d(M.L + o, len, x) : lower_bound_d(M.L + o, len, x);
        rank += k;
        nlive += M.G[o + q + k];
    }
    M.perm[rank] = g; M.Ls[rank] = x; M.nl[rank] = nlive < 1 ? 1 : nlive;
}
__global__ void k_merge_dx(const int *nl, long long n, P2 *X)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)nl[i], l0 = log(v);
    X[i] = P2{l0 - log(v + 1.0), l0 - log(v + 2.0)};
}
// X: (logX, logXX) AFTER each death.  T: terms of <Z X> / X (run_time_info.f90:262-271 with the decay factored out)
__global__ void k_merge_t(const int *nl, const double *Ls, const P2 *X, long long n, P2 *T)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)nl[i], l0 = log(v), l1 = log(v + 1.0), l2 = log(v + 2.0);
    const double XXm = X[i].b - (l0 - l2);
    T[i] = P2{XXm + Ls[i] + l0 - l1 - l2 - X[i].a, 1.0};
}
// per merged death: log weight, the two evidence terms; per chunk of 1024 deaths: (max, sum) partials of both sums
// and the largest posterior log-weight
__global__ __launch_bounds__(1024) void k_merge_terms(const int *nl, const double *Ls, const P2 *X, const P2 *T, long long n,
                                                      double *logw, P2 *partA, P2 *partB, double *partM)
{
    __shared__ P2 sa[16], sb[16];
    __shared__ double sm[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long i = (long long)blockIdx.x * 1024 + tid;
    P2 a = OpLse::id(), b = OpLse::id();
    double pm = NEGBIG;
    if (i < n) {
        const double v = (double)nl[i], l0 = log(v), l1 = log(v + 1.0), l2 = log(v + 2.0), L = Ls[i];
        const double Xm = X[i].a - (l0 - l1), XXm = X[i].b - (l0 - l2);
        const double ZXm = i ? lsv(T[i - 1]) + X[i - 1].a : NEGBIG;
        const double log2v = 0.6931471805599453;
        a = P2{Xm + L - l1, 1.0};
        b = OpLse::comb(P2{log2v + ZXm + L - l1, ZXm > NEGBIG / 2 ? 1.0 : 0.0}, P2{log2v + XXm + 2.0 * L - l1 - l2, 1.0});
        logw[i] = Xm - l1;
        pm = Xm - l1 + L;
    }
    for (int s = 32; s > 0; s >>= 1) {
        const P2 oa = P2{__shfl_xor(a.a, s), __shfl_xor(a.b, s)}, ob = P2{__shfl_xor(b.a, s), __shfl_xor(b.b, s)};
        a = OpLse::comb(a, oa); b = OpLse::comb(b, ob);
        pm = fmax(pm, __shfl_xor(pm, s));
    }
    if (lane == 0) { sa[wv] = a; sb[wv] = b; sm[wv] = pm; }
    __syncthreads();
    if (tid == 0) {
        for (int x = 1; x < 16; ++x) { a = OpLse::comb(a, sa[x]); b = OpLse::comb(b, sb[x]); pm = fmax(pm, sm[x]); }
        partA[blockIdx.x] = a; partB[blockIdx.x] = b; partM[blockIdx.x] = pm;
    }
}
// weighted moments of theta and phi: thread = column, a workgroup walks a chunk of merged deaths (rows are read whole,
// coalesced across the columns); part[b][2 nP + 1] = sum w x, sum w x^2, sum w
#define MM_CHUNK 512
__global__ __launch_bounds__(256) void k_merge_moments(MergeDev M, const double *logw, double wmax, int p0, int nP, double *part)
{
    __shared__ double sw[MM_CHUNK];
    __shared__ long long sg[MM_CHUNK];
    const int tid = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * MM_CHUNK;
    const int m = (int)((M.n - i0) < MM_CHUNK ? (M.n - i0) : MM_CHUNK);
    for (int k = tid; k < m; k += 256) { sw[k] = exp(logw[i0 + k] + M.Ls[i0 + k] - wmax); sg[k] = M.perm[i0 + k]; }
    __syncthreads();
    double *out = part + (size_t)blockIdx.x * (2 * nP + 1);
    for (int c = tid; c < nP; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < m; ++k) { const double x = M.rows[(size_t)sg[k] * M.nT + p0 + c]; s1 += sw[k] * x; s2 += sw[k] * x * x; }
        out[c] = s1; out[nP + c] = s2;
    }
    if (tid == 0) { double s = 0.0; for (int k = 0; k < m; ++k) s += sw[k]; out[2 * nP] = s; }
}
// merged rows in death order, the birth column replaced by the entry contour (what a replay of the file needs)
__global__ __launch_bounds__(64) void k_merge_gather(MergeDev M, int b0, double *out)
{
    const long long i = blockIdx.x;
    const long long g = M.perm[i];
    const double *src = M.rows + (size_t)g * M.nT;
    double *dst = out + (size_t)i * M.nT;
    for (int e = threadIdx.x; e < M.nT; e += 64) dst[e] = (e == b0) ? M.entry[g] : src[e];
}


// ---- the lived records of ONE run, picked on the device.  A run hands back every dead point (pinned host memory); the
// points that never entered a live set (failed spawns, logweight = logzero) are no part of its death sequence.  The
// arrays go to the device whole (one DMA each) and are compacted there: flags + counts per 256 rows, the offsets of the
// blocks by one workgroup, one wave per surviving row.
#define PK_ROWS 256
__global__ __launch_bounds__(PK_ROWS) void k_pack_flag(const double *logw, long long nd, double logzero, int *blk_count)
{
    const long long i = (long long)blockIdx.x * PK_ROWS + threadIdx.x;
    const int keep = (i < nd) && (logw[i] > logzero);
    const unsigned long long m = __ballot(keep);
    __shared__ int wc[PK_ROWS / 64];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < PK_ROWS / 64; ++w) t += wc[w]; blk_count[blockIdx.x] = t; }
}
// exclusive offsets of the blocks in place, the total behind them (one workgroup; fixed order)
__global__ __launch_bounds__(1024) void k_pack_scan(int *blk, int nb, long long *total)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int b = b0 + tid, x = (b < nb) ? blk[b] : 0;
        int v = x;
        for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(v, s); if (lane >= s) v += t; }
        if (lane == 63) wsum[wv] = v;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wv; ++w) pre += wsum[w];
        v += pre;
        __syncthreads();
        if (b < nb) blk[b] = v - x;
        if (tid == 1023) carry = v;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}
// surviving row i of block b -> rows_out[off_b + rank in block]; entry contour next to it.  One wave per row of the block
// in turn (4 waves), lane = column
__global__ __launch_bounds__(PK_ROWS) void k_pack_scatter(const double *dead, const double *logw, const double *entry, long long nd, int nT,
                                                        double logzero, const int *blk_off, double *rows_out, double *entry_out, double *ownw_out)
{
    __shared__ int dst[PK_ROWS];
    __shared__ int wc[PK_ROWS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long i = (long long)blockIdx.x * PK_ROWS + tid;
    const int keep = (i < nd) && (logw[i] > logzero);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wc[wv] = __popcll(m);
    __syncthreads();
    int pre = blk_off[blockIdx.x];
    for (int w = 0; w < wv; ++w) pre += wc[w];
    dst[tid] = keep ? pre + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    if (keep) { entry_out[dst[tid]] = entry[i]; if (ownw_out) ownw_out[dst[tid]] = logw[i]; }
    __syncthreads();
    for (int r = wv; r < PK_ROWS; r += PK_ROWS / 64) {
        const int d = dst[r];
        if (d < 0) continue;
        const double *src = dead + ((size_t)blockIdx.x * PK_ROWS + r) * nT;
        double *o = rows_out + (size_t)d * nT;
        for (int e = lane; e < nT; e += 64) o[e] = src[e];
    }
}
// gathered blocks [rank][rows nmax x nT | entry nmax | own log weight nmax] -> the ranks' records one after the other (a rank's
// records: its runs one after the other); off = first record of each RANK
__global__ __launch_bounds__(64) void k_unpad(const double *recv, long long nmax, int nT, int R, const long long *off, double *rows, double *entry, double *ownw)
{
    const long long g = blockIdx.x;                // merged record index
    int q = 0;
    for (int r = 1; r < R; ++r) q = (g >= off[r]) ? r : q;
    const long long k = g - off[q];
    const double *blk = recv + (size_t)q * (size_t)nmax * (nT + 2);
    const double *src = blk + (size_t)k * nT;
    double *dst = rows + (size_t)g * nT;
    for (int e = threadIdx.x; e < nT; e += 64) dst[e] = src[e];
    if (threadIdx.x == 0) { entry[g] = blk[(size_t)nmax * nT + k]; ownw[g] = blk[(size_t)nmax * (nT + 1) + k]; }
}
// the union's weights when a run had clusters (pchip_merged::evidence_rule 1): record i keeps the prior-volume weight its OWN run gave
// it -- its cluster's volume over the cluster's live count, run_time_info.f90:211-296 -- over the number of runs; per chunk of 1024 the
// largest posterior log-weight
__global__ __launch_bounds__(1024) void k_merge_ownw(const long long *perm, const double *ownw, const double *Ls, long long n, double logR,
                                                     double *logw, double *partM)
{
    __shared__ double sm[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long i = (long long)blockIdx.x * 1024 + tid;
    double pm = NEGBIG;
    if (i < n) { const double w = ownw[perm[i]] - logR; logw[i] = w; pm = w + Ls[i]; }
    for (int s = 32; s > 0; s >>= 1) pm = fmax(pm, __shfl_xor(pm, s));
    if (lane == 0) sm[wv] = pm;
    __syncthreads();
    if (tid == 0) { for (int x = 1; x < 16; ++x) pm = fmax(pm, sm[x]); partM[blockIdx.x] = pm; }
}

// ---- RCCL, resolved at run time: the engine has no link-time dependency on a collective library (a single-GPU user needs
// none), and inside a torch process the RCCL that is already loaded is the one that is used.
struct Rccl {
    void *h = nullptr; bool tried = false; std::string where;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        static std::mutex m;
        std::lock_guard<std::mutex> g(m);
        if (tried) return h != nullptr;
        tried = true;
        const char *env = std::getenv("PCHIP_RCCL_LIB");
        const char *names[] = { env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (int pass = 0; pass < 2 && !h; ++pass)             // first a copy that is already in the process, then from disk
            for (const char *n : names) {
                if (!n || !*n) continue;
                // RTLD_LOCAL: with RTLD_GLOBAL the static objects of /opt/rocm's librocm_smi64 (a dependency of RCCL) were torn
                // down twice at process exit ("free(): invalid pointer" in ~map<amd::smi::DevInfoTypes, ...>, rc 134)
                int fl = RTLD_NOW | RTLD_LOCAL;
                if (const char *e = std::getenv("PCHIP_RCCL_DLOPEN")) { if (std::strstr(e, "global")) fl = RTLD_NOW | RTLD_GLOBAL; }   // (developer switch)
                h = dlopen(n, fl | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) { where = n; break; }
            }
        if (!h) return false;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) { h = nullptr; return false; }
        return true;
    }
};
Rccl &rccl() { static Rccl *r = new Rccl; return *r; }

// (device scratch and the merged rows come from the engine's block caches, pc_engine.hip)
extern "C" void *pc_cache_dev_alloc(size_t bytes);
extern "C" void pc_cache_dev_free(void *p);
extern "C" void *pc_cache_host_alloc(size_t bytes);
extern "C" void pc_cache_host_free(void *p);
// host side of a blocking copy: a pinned block of the cache (see pchip_merge_records on why not a vector)
template <class T> struct PinBuf {
    T *p; size_t n;
    explicit PinBuf(size_t n_) : p((T *)pc_cache_host_alloc(sizeof(T) * (n_ ? n_ : 1))), n(n_) {}
    ~PinBuf() { pc_cache_host_free(p); }
    PinBuf(const PinBuf &) = delete; PinBuf &operator=(const PinBuf &) = delete;
    T *data() { return p; } size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
};
struct DevBuf {
    std::vector<void *> v;
    template <class T> T *get(size_t n) { void *p = pc_cache_dev_alloc(sizeof(T) * (n ? n : 1)); if (!p) return nullptr; v.push_back(p); return (T *)p; }
    ~DevBuf() { for (void *p : v) pc_cache_dev_free(p); }
};

// two steps, because the caller sizes the destination with the counts of ALL runs (or ranks): count() uploads the weights,
// flags and scans (one host wait for the number of records); scatter() uploads the rows and writes the survivors where
// the caller wants them, on the run's device
struct PackJob {
    const pchip_result *r = nullptr; double logzero = 0.0; int device = 0, nT = 0, nb = 0; long long nd = 0, count = 0;
    double *d_logw = nullptr; int *d_blk = nullptr; long long *d_total = nullptr;
    bool count_records(DevBuf &B, hipStream_t st)
    {
        nd = r->ndead; nT = r->nTotal; nb = (int)((nd + PK_ROWS - 1) / PK_ROWS); count = 0;
        if (nd <= 0) return true;
        if (hipSetDevice(device) != hipSuccess) return false;
        d_logw = B.get<double>(nd); d_blk = B.get<int>(nb + 1); d_total = B.get<long long>(1);
        if (!d_logw || !d_blk || !d_total) return false;
        if (hipMemcpyAsync(d_logw, r->logweights, sizeof(double) * nd, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        hipLaunchKernelGGL(k_pack_flag, dim3(nb), dim3(PK_ROWS), 0, st, (const double *)d_logw, nd, logzero, d_blk);
        hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, d_blk, nb, d_total);
        if (hipMemcpyAsync(&count, d_total, sizeof(long long), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        return hipStreamSynchronize(st) == hipSuccess;
    }
    bool scatter(DevBuf &B, double *rows_out, double *entry_out, double *ownw_out, hipStream_t st)
    {
        if (nd <= 0 || count <= 0) return true;
        if (hipSetDevice(device) != hipSuccess) return false;
        double *d_dead = B.get<double>((size_t)nd * nT), *d_entry = B.get<double>(nd);
        if (!d_dead || !d_entry) return false;
        if (hipMemcpyAsync(d_dead, r->dead, sizeof(double) * (size_t)nd * nT, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipMemcpyAsync(d_entry, r->entry, sizeof(double) * nd, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        hipLaunchKernelGGL(k_pack_scatter, dim3(nb), dim3(PK_ROWS), 0, st, (const double *)d_dead, (const double *)d_logw, (const double *)d_entry,
                           nd, nT, logzero, (const int *)d_blk, rows_out, entry_out, ownw_out);
        return hipGetLastError() == hipSuccess;
    }
};

}  // namespace

struct pchip_comm { ncclComm_t comm = nullptr; int nranks = 1, rank = 0, device = 0; pchip_allgather_fn fn = nullptr; void *user = nullptr; };
// the one collective of the exchange: every rank's `bytes` at `send` -> all ranks' blocks, rank after rank, at `recv` (device memory both).
// RCCL (ncclAllGather on the stream), or the caller's own all-gather (pchip_comm_create_with: called with the stream drained, returns when
// `recv` is filled).  0, or a message and 2.
static int comm_all_gather(pchip_comm *c, const void *send, void *recv, size_t count, bool words, hipStream_t st, const char *what)
{
    if (c->fn) {
        if (hipStreamSynchronize(st) != hipSuccess) { std::fprintf(stderr, "polychord_hip: comm merge: %s: the stream failed before the exchange\n", what); return 2; }
        const int e = c->fn(c->user, send, recv, count * 8);
        if (e != 0) { std::fprintf(stderr, "polychord_hip: comm merge: %s: the caller's all-gather returned %d\n", what, e); return 2; }
        return 0;
    }
    const ncclResult_t e = rccl().AllGather(send, recv, count, words ? ncclInt64 : ncclDouble, c->comm, st);
    if (e != ncclSuccess) { std::fprintf(stderr, "polychord_hip: comm merge: %s: %s\n", what, rccl().GetErrorString(e)); return 2; }
    return 0;
}

// The lived records of a run picked where they were made (settings.device_records; called by the engine at the end of a run, on the
// run's device, behind everything that wrote the arrays): flags, offsets, scatter into `block` -- rows [cap][nT] | entry [cap] | own log
// weight [cap] | scratch (block counts and the total) -- and the count into *h_count (pinned) by a copy in stream order.
extern "C" size_t pc_records_block_bytes(long long cap, int nT)
{
    const long long nb = (cap + PK_ROWS - 1) / PK_ROWS;
    return sizeof(double) * (size_t)cap * (nT + 2) + sizeof(long long) * (size_t)(nb + 4);
}
extern "C" int pc_pack_lived_device(const double *dead, const double *logw, const double *entry, long long nd, int nT, double logzero,
                                    double *block, long long cap, long long *h_count, hipStream_t st)
{
    *h_count = 0;
    if (nd <= 0) return 0;
    const int nb = (int)((nd + PK_ROWS - 1) / PK_ROWS);
    double *rows = block, *ent = block + (size_t)cap * nT, *own = block + (size_t)cap * (nT + 1);
    long long *d_total = (long long *)(block + (size_t)cap * (nT + 2));
    int *d_blk = (int *)(d_total + 2);
    hipLaunchKernelGGL(k_pack_flag, dim3(nb), dim3(PK_ROWS), 0, st, logw, nd, logzero, d_blk);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, d_blk, nb, d_total);
    hipLaunchKernelGGL(k_pack_scatter, dim3(nb), dim3(PK_ROWS), 0, st, dead, logw, entry, nd, nT, logzero, (const int *)d_blk, rows, ent, own);
    if (hipMemcpyAsync(h_count, d_total, sizeof(long long), hipMemcpyDeviceToHost, st) != hipSuccess) return 2;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" {

void pchip_merged_free(pchip_merged *m)
{
    if (!m) return;
    if (m->rows) pc_cache_host_free(m->rows);
    pc_cache_host_free(m->logweights); pc_cache_host_free(m->nlive); std::free(m->post_mean); std::free(m->post_var);
    std::memset(m, 0, sizeof(*m));
}

// The evidence of R independent runs from the runs' OWN evidences (pchip_merged::evidence_rule 1).  Each run reports a log-normal
// Z_r: log<Z_r> = logZ_r + v_r / 2, log<Z_r^2> = 2 logZ_r + 2 v_r (run_time_info.f90:652-678 read backwards).  The mean of the runs'
// Z in linear space, Zbar = sum Z_r / R, has <Zbar> = sum <Z_r> / R and <Zbar^2> = (sum <Z_r^2> + (sum <Z_r>)^2 - sum <Z_r>^2) / R^2
// (independent runs); its variance in the log is the larger of that propagated one and of the scatter BETWEEN the runs (what a run does
// not know about itself: which modes it found), log(1 + s^2 / (R <Zbar>^2)) with s^2 the sample variance of the <Z_r>.  One run gives
// its own logZ and variance back.
static void runs_combined_evidence(const double *lz, const double *var, int R, double *logZ, double *varlogZ)
{
    double mx = -1.7e308;
    for (int r = 0; r < R; ++r) mx = std::max(mx, lz[r] + 0.5 * var[r]);
    double s1 = 0.0, s2 = 0.0, sq = 0.0;            // sums of <Z_r>, <Z_r>^2, <Z_r^2>, all relative to exp(mx)
    for (int r = 0; r < R; ++r) {
        const double m = std::exp(lz[r] + 0.5 * var[r] - mx);
        s1 += m; s2 += m * m; sq += std::exp(2.0 * lz[r] + 2.0 * var[r] - 2.0 * mx);
    }
    const double mean = s1 / R;
    const double second = (sq + s1 * s1 - s2) / ((double)R * R);
    double v = std::log(second) - 2.0 * std::log(mean);
    if (R > 1) {
        double ss = 0.0;
        for (int r = 0; r < R; ++r) { const double d = std::exp(lz[r] + 0.5 * var[r] - mx) - mean; ss += d * d; }
        const double between = std::log1p(ss / (double)(R - 1) / (double)R / (mean * mean));
        v = std::max(v, between);
    }
    if (!(v > 0.0)) v = 0.0;
    *logZ = std::log(mean) + mx - 0.5 * v;           // location of the log-normal with this mean and this variance of the log
    *varlogZ = v;
}

int pchip_merge_records(int nDims, int nDerived, int nruns, const long *counts, const double *rows, const double *entry,
                        int on_device, int want_rows, pchip_merged *out)
{
    return pchip_merge_records_ex(nDims, nDerived, nruns, counts, rows, entry, nullptr, nullptr, nullptr, nullptr, on_device, want_rows, out);
}

int pchip_merge_records_ex(int nDims, int nDerived, int nruns, const long *counts, const double *rows, const double *entry,
                           const double *ownw, const double *run_logZ, const double *run_varlogZ, const int *run_clustered,
                           int on_device, int want_rows, pchip_merged *out)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    std::memset(out, 0, sizeof(*out));
    const int nT = 2 * nDims + nDerived + 2, nP = nDims + nDerived, p0 = nDims, b0 = 2 * nDims + nDerived, l0 = b0 + 1;
    if (nruns < 1 || nruns > 4096) { std::fprintf(stderr, "polychord_hip: merge of %d runs\n", nruns); return 1; }
    std::vector<long long> off(nruns + 1, 0);
    for (int q = 0; q < nruns; ++q) { if (counts[q] < 0) return 1; off[q + 1] = off[q] + counts[q]; }
    const long long n = off[nruns];
    out->n = (long)n; out->nTotal = nT; out->nruns = nruns;
    out->post_mean = (double *)std::calloc(std::max(1, nP), sizeof(double)); out->post_var = (double *)std::calloc(std::max(1, nP), sizeof(double));
    if (n == 0) { out->logZ = -1e30; return 0; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "polychord_hip: no HIP device available -- the merge has no CPU path\n"); return 2; }
    if (on_device) {                                  // work where the records live
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, rows) == hipSuccess) (void)hipSetDevice(at.device); else (void)hipGetLastError();
    }
    hipStream_t st = nullptr;
    DevBuf B;
    auto fail = [&](const char *what) { std::fprintf(stderr, "polychord_hip: merge: %s (%s)\n", what, hipGetErrorString(hipGetLastError())); pchip_merged_free(out); return 7; };
    // which evidence the union quotes: a run that held more than one cluster at some time (pchip_result.ncluster_peak > 1) weighed its dead points by its
    // clusters' volumes, which a replay of the union from ranks and live counts does not know (10-D Rastrigin, dozens of clusters: the
    // replay sits 0.46 below the runs' own log Z, twenty of its own error bars) -- then the runs' own evidences and weights are used
    int nclustered = 0;
    if (run_clustered && ownw && run_logZ && run_varlogZ) for (int q = 0; q < nruns; ++q) nclustered += run_clustered[q] != 0;
    const bool own_rule = nclustered > 0;
    const double *d_rows = rows, *d_entry = entry, *d_ownw = ownw;
    if (!on_device) {
        double *r = B.get<double>((size_t)n * nT), *e = B.get<double>(n);
        if (!r || !e) return fail("out of device memory");
        if (hipMemcpy(r, rows, sizeof(double) * (size_t)n * nT, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
        if (hipMemcpy(e, entry, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
        d_rows = r; d_entry = e;
        if (own_rule) {
            double *w = B.get<double>(n);
            if (!w) return fail("out of device memory");
            if (hipMemcpy(w, ownw, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
            d_ownw = w;
        }
    }
    const int nbs = (int)((n + SC_CHUNK - 1) / SC_CHUNK), nbt = (int)((n + 1023) / 1024), nbm = (int)((n + MM_CHUNK - 1) / MM_CHUNK);
    MergeDev M{};
    M.R = nruns; M.nT = nT; M.l0 = l0; M.n = n; M.rows = d_rows; M.entry = d_entry;
    long long *d_off = B.get<long long>(nruns + 1);
    M.L = B.get<double>(n); M.G = B.get<int>(n + nruns); M.perm = B.get<long long>(n); M.Ls = B.get<double>(n); M.nl = B.get<int>(n);
    P2 *X = B.get<P2>(n), *T = B.get<P2>(n), *tot = B.get<P2>(nbs + 1), *pA = B.get<P2>(nbt), *pB = B.get<P2>(nbt);
    double *pM = B.get<double>(nbt), *d_logw = B.get<double>(n), *pmom = B.get<double>((size_t)nbm * (2 * nP + 1));
    if (!d_off || !M.L || !M.G || !M.perm || !M.Ls || !M.nl || !X || !T || !tot || !pA || !pB || !pM || !d_logw || !pmom) return fail("out of device memory");
    M.off = d_off;
    if (hipMemcpy(d_off, off.data(), sizeof(long long) * (nruns + 1), hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
    const int nb256 = (int)((n + nruns + 255) / 256);
    hipLaunchKernelGGL(k_merge_extract, dim3(nb256), dim3(256), 0, st, M);
    hipLaunchKernelGGL(k_merge_hist, dim3(nb256), dim3(256), 0, st, M);
    hipLaunchKernelGGL(k_merge_livecount, dim3(nruns), dim3(1024), 0, st, M);
    hipLaunchKernelGGL(k_merge_rank, dim3(nb256), dim3(256), 0, st, M);
    hipLaunchKernelGGL(k_merge_dx, dim3(nb256), dim3(256), 0, st, (const int *)M.nl, n, X);
    device_scan<OpAdd>(X, n, tot, st);
    hipLaunchKernelGGL(k_merge_t, dim3(nb256), dim3(256), 0, st, (const int *)M.nl, (const double *)M.Ls, (const P2 *)X, n, T);
    device_scan<OpLse>(T, n, tot, st);
    hipLaunchKernelGGL(k_merge_terms, dim3(nbt), dim3(1024), 0, st, (const int *)M.nl, (const double *)M.Ls, (const P2 *)X, (const P2 *)T, n, d_logw, pA, pB, pM);
    PinBuf<P2> hA(nbt), hB(nbt); PinBuf<double> hM(nbt);
    if (!hA.p || !hB.p || !hM.p) return fail("out of pinned memory");
    if (hipMemcpy(hA.data(), pA, sizeof(P2) * nbt, hipMemcpyDeviceToHost) != hipSuccess) return fail("evidence kernels");
    (void)hipMemcpy(hB.data(), pB, sizeof(P2) * nbt, hipMemcpyDeviceToHost); (void)hipMemcpy(hM.data(), pM, sizeof(double) * nbt, hipMemcpyDeviceToHost);
    auto comb = [](P2 x, P2 y) { const double e = std::exp(-std::fabs(x.a - y.a)); return P2{std::max(x.a, y.a), x.a >= y.a ? x.b + y.b * e : x.b * e + y.b}; };
    P2 a = hA[0], b = hB[0]; double wmax = hM[0];
    for (int k = 1; k < nbt; ++k) { a = comb(a, hA[k]); b = comb(b, hB[k]); wmax = std::max(wmax, hM[k]); }
    const double lZ = a.a + std::log(a.b), lZ2 = b.a + std::log(b.b);       // log <Z>, log <Z^2>
    out->logZ = 2.0 * lZ - 0.5 * lZ2; out->varlogZ = lZ2 - 2.0 * lZ;       // run_time_info.f90:652-678
    out->logZ_replay = out->logZ; out->varlogZ_replay = out->varlogZ; out->evidence_rule = 0; out->nclustered = nclustered;
    if (own_rule) {
        runs_combined_evidence(run_logZ, run_varlogZ, nruns, &out->logZ, &out->varlogZ);
        out->evidence_rule = 1;
        hipLaunchKernelGGL(k_merge_ownw, dim3(nbt), dim3(1024), 0, st, (const long long *)M.perm, d_ownw, (const double *)M.Ls, n, std::log((double)nruns), d_logw, pM);
        if (hipMemcpy(hM.data(), pM, sizeof(double) * nbt, hipMemcpyDeviceToHost) != hipSuccess) return fail("weight kernel");
        wmax = hM[0];
        for (int k = 1; k < nbt; ++k) wmax = std::max(wmax, hM[k]);
    }
    hipLaunchKernelGGL(k_merge_moments, dim3(nbm), dim3(256), 0, st, M, (const double *)d_logw, wmax, p0, nP, pmom);
    PinBuf<double> hm((size_t)nbm * (2 * nP + 1));
    if (!hm.p) return fail("out of pinned memory");
    if (hipMemcpy(hm.data(), pmom, sizeof(double) * hm.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail("moment kernel");
    double sw = 0.0;
    for (int k = 0; k < nbm; ++k) {
        const double *p = hm.data() + (size_t)k * (2 * nP + 1);
        sw += p[2 * nP];
        for (int c = 0; c < nP; ++c) { out->post_mean[c] += p[c]; out->post_var[c] += p[nP + c]; }
    }
    for (int c = 0; c < nP; ++c) { out->post_mean[c] /= sw; out->post_var[c] = out->post_var[c] / sw - out->post_mean[c] * out->post_mean[c]; }
    // (pinned blocks of the cache: a blocking copy into pageable memory has the driver pin the pages for the copy, and when the caller
    //  gives the arrays back -- megabytes: unmapped at once -- the kernel driver takes the process's queues off the device and puts
    //  them back some 20 ms later: the next call's first wait for the device, if it came at once, sat behind that)
    out->logweights = (double *)pc_cache_host_alloc(sizeof(double) * n); out->nlive = (int *)pc_cache_host_alloc(sizeof(int) * n);
    if (!out->logweights || !out->nlive) return fail("out of pinned memory");
    (void)hipMemcpy(out->logweights, d_logw, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(out->nlive, M.nl, sizeof(int) * n, hipMemcpyDeviceToHost);
    if (want_rows) {
        double *d_out = B.get<double>((size_t)n * nT);
        if (!d_out) return fail("out of device memory");
        hipLaunchKernelGGL(k_merge_gather, dim3((unsigned)n), dim3(64), 0, st, M, b0, d_out);
        out->rows = (double *)pc_cache_host_alloc(sizeof(double) * (size_t)n * nT);
        if (!out->rows) return fail("out of pinned memory");
        if (hipMemcpy(out->rows, d_out, sizeof(double) * (size_t)n * nT, hipMemcpyDeviceToHost) != hipSuccess) return fail("download");
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("kernels");
    out->t_merge_s = std::chrono::duration<double>(clk::now() - t0).count();
    return 0;
}

// Independent repeats of one problem spread over the HIP devices of this process (one host thread per run in flight,
// each run a complete engine on its own stream of its own device), merged on devices[0].  This is the front door of the
// repeat-sharded mode for a single process; between processes (one per GPU, torch.distributed / RCCL) the same merge
// is fed by an all-gather (polychordlite_amd/merge.py).
// mean of the runs' own log Z and its standard error (pchip_merged::runs_logZ_mean / _sem); one run: its own reported error
static void runs_evidence(const double *z, int n, double err_one, pchip_merged *out)
{
    double m = 0.0;
    for (int k = 0; k < n; ++k) m += z[k];
    m /= (double)(n > 0 ? n : 1);
    double v = 0.0;
    for (int k = 0; k < n; ++k) v += (z[k] - m) * (z[k] - m);
    out->runs_logZ_mean = m;
    out->runs_logZ_sem = n > 1 ? std::sqrt(v / (double)(n - 1) / (double)n) : err_one;
}

int pchip_run_repeats(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds,
                      int ndevices, const int *devices, int max_in_flight, pchip_result *results, pchip_merged *merged)
{
    return pchip_run_repeats_ex(s, like, prior, nseeds, seeds, ndevices, devices, max_in_flight, 1, results, merged);
}

int pchip_run_repeats_ex(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds,
                         int ndevices, const int *devices, int max_in_flight, int want_rows, pchip_result *results, pchip_merged *merged)
{
    using clk = std::chrono::steady_clock;
    if (nseeds < 1) return 1;
    int ndev_all = 0;
    if (hipGetDeviceCount(&ndev_all) != hipSuccess || ndev_all == 0) { std::fprintf(stderr, "polychord_hip: no HIP device available -- this engine has no CPU path\n"); return 2; }
    std::vector<int> devs;
    if (ndevices > 0 && devices) { for (int k = 0; k < ndevices; ++k) { if (devices[k] < 0 || devices[k] >= ndev_all) { std::fprintf(stderr, "polychord_hip: device %d of %d\n", devices[k], ndev_all); return 1; } devs.push_back(devices[k]); } }
    else devs.push_back(s->device >= 0 ? s->device % ndev_all : 0);
    // Built-in device likelihoods: one host thread per DEVICE keeps up to max_in_flight runs in flight there (pc_run_many).  Host
    // callbacks: one thread per run in flight, as the caller's callbacks have to be called from somewhere.
    const int per_dev = std::max(1, max_in_flight);
    std::atomic<int> next{0}, worst{0};
    const auto t0 = clk::now();
    const bool device_like = like->kind != PCHIP_LIKE_CALLBACK && prior->kind == 1 && !std::getenv("PC_REPEATS_THREADS");
    // (the runs leave their lived records on the device for the merge below: no second trip over the host link)
    pchip_settings s_loc = *s;
    // (set here, freed here: a caller that did not ask for the device block must not find its results pinning ndead x (nTotal + 2) doubles
    //  of device memory each until it frees them -- see below, behind the union)
    const bool records_are_mine = merged && !s->device_records && !std::getenv("PC_DEVICE_RECORDS_OFF");
    if (records_are_mine) s_loc.device_records = 1;
    s = &s_loc;
    if (device_like) {
        // seeds dealt round-robin to the devices: run k on devs[k % ndev] (what the merge below assumes)
        // (a device's runs may be shared out among a few scheduler threads: one thread's launch path saturates at about eight runs)
        // (runs with clustering: two groups going round by round side by side -- while one group's contractions hold a wavefront each, the
        //  other samples or updates; measured at BASELINE configs[2] / [3], sixteen runs: 969 -> ~780 ms, 541 -> ~400 ms.  One-cluster
        //  Gaussian runs fill the chip with one group, and two were slower: DESIGN section 5f)
        // (round 5: four groups of at least four runs each -- one main stream per hardware queue, which the groups now keep apart
        //  (CohortStreams, pc_engine.hip); sixteen runs of configs[2] / [3]: 495 / 386 ms with two groups, 430 / 340 with four; with six or
        //  eight groups main streams share queues again and the call takes twice as long.  And four groups whatever they hold: four runs
        //  as four groups of one 172 ms against 274 in one group, eight as four groups of two 264 against 325 in two)
        const int sched = std::getenv("PC_REPEATS_SCHED") ? std::max(1, std::atoi(std::getenv("PC_REPEATS_SCHED"))) : ((s->do_clustering && per_dev >= 2) ? std::min(4, per_dev) : 1);
        const int nd = (int)devs.size();
        // (several groups on a device: streams of known hardware-queue classes for all of them, classed now, while the device is idle)
        if (sched > 1) for (int d : devs) pc_prepare_streams(d, 4 * sched);
        auto work = [&](int wi) {
            const int di = wi % nd, part = wi / nd;
            std::vector<int> mine, idx;
            int j = 0;
            for (int k = di; k < nseeds; k += nd, ++j) if (j % sched == part) { mine.push_back(seeds[k]); idx.push_back(k); }
            if (mine.empty()) return;
            std::vector<pchip_result> res(mine.size());
            const int rc = pc_run_many(s, like, prior, (int)mine.size(), mine.data(), devs[di], std::max(1, (per_dev + sched - 1) / sched), res.data());
            if (rc != 0) { int z = 0; worst.compare_exchange_strong(z, rc); }
            for (size_t a = 0; a < idx.size(); ++a) results[idx[a]] = res[a];
        };
        std::vector<std::thread> th;
        for (int w = 1; w < nd * sched; ++w) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    } else {
        const int nworkers = std::min(nseeds, per_dev * (int)devs.size());
        auto work = [&](int wid) {
            // (run k must land on devs[k % ndev]: a worker takes the seeds of its own device only)
            const int di = wid % (int)devs.size();
            for (int k = next.fetch_add(1); k < nseeds; k = next.fetch_add(1)) {
                pchip_settings c = *s;
                c.seed = seeds[k]; c.device = devs[k % devs.size()];
                (void)di;
                const int rc = pchip_run_hooks(&c, like, prior, nullptr, &results[k]);
                if (rc != 0) { int z = 0; worst.compare_exchange_strong(z, rc); }
            }
        };
        std::vector<std::thread> th;
        for (int w = 1; w < nworkers; ++w) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    }
    const double t_runs = std::chrono::duration<double>(clk::now() - t0).count();
    if (worst.load() != 0) { for (int k = 0; k < nseeds; ++k) pchip_result_free(&results[k]); return worst.load(); }
    if (!merged) return 0;
    // the union of the points that entered a live set (failed spawns carry logweight = logzero), run after run: every run's
    // records are picked on the device it ran on and land in ONE buffer on devices[0] -- written there directly, or over
    // xGMI by a peer copy; nothing passes through host vectors
    const int nT = results[0].nTotal;
    std::vector<long> counts(nseeds, 0);
    int rc = 0;
    {
        std::vector<DevBuf> scratch(devs.size());                       // per device: freed with that device current
        std::vector<PackJob> jobs(nseeds);
        size_t ntot = 0;
        for (int k = 0; k < nseeds && rc == 0; ++k) {
            PackJob &J = jobs[k];
            J.r = &results[k]; J.logzero = s->logzero; J.device = devs[k % devs.size()];
            if (results[k].d_records) J.count = results[k].n_records;                 // (picked on the device when the run ended)
            else if (!J.count_records(scratch[k % devs.size()], nullptr)) rc = 7;
            counts[k] = (long)J.count; ntot += (size_t)J.count;
        }
        DevBuf U;
        double *rows_all = nullptr, *entry_all = nullptr, *ownw_all = nullptr;
        if (rc == 0) {
            (void)hipSetDevice(devs[0]);
            rows_all = U.get<double>(ntot * nT); entry_all = U.get<double>(ntot); ownw_all = U.get<double>(ntot);
            if (!rows_all || !entry_all || !ownw_all) rc = 7;
        }
        size_t o = 0;
        for (int k = 0; k < nseeds && rc == 0; ++k) {
            PackJob &J = jobs[k];
            DevBuf &B = scratch[k % devs.size()];
            if (results[k].d_records) {
                // device to device (a peer copy when the run was made on another device): rows, entry contours, own weights
                const pchip_result &r = results[k];
                const double *rr = r.d_records, *re = r.d_records + (size_t)r.records_cap * nT, *rw = r.d_records + (size_t)r.records_cap * (nT + 1);
                if (J.count > 0) {
                    if (r.records_device == devs[0]) {
                        (void)hipSetDevice(devs[0]);
                        if (hipMemcpyAsync(rows_all + o * nT, rr, sizeof(double) * (size_t)J.count * nT, hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||
                            hipMemcpyAsync(entry_all + o, re, sizeof(double) * J.count, hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||
                            hipMemcpyAsync(ownw_all + o, rw, sizeof(double) * J.count, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = 2;
                    } else {
                        (void)hipSetDevice(r.records_device);
                        if (hipMemcpyPeerAsync(rows_all + o * nT, devs[0], rr, r.records_device, sizeof(double) * (size_t)J.count * nT, nullptr) != hipSuccess ||
                            hipMemcpyPeerAsync(entry_all + o, devs[0], re, r.records_device, sizeof(double) * J.count, nullptr) != hipSuccess ||
                            hipMemcpyPeerAsync(ownw_all + o, devs[0], rw, r.records_device, sizeof(double) * J.count, nullptr) != hipSuccess) rc = 2;
                    }
                }
            }
            else if (J.device == devs[0]) { if (!J.scatter(B, rows_all + o * nT, entry_all + o, ownw_all + o, nullptr)) rc = 7; }
            else if (J.count > 0) {
                (void)hipSetDevice(J.device);
                double *tr = B.get<double>((size_t)J.count * nT), *te = B.get<double>(2 * (size_t)J.count);
                if (!tr || !te || !J.scatter(B, tr, te, te + J.count, nullptr)) { rc = 7; break; }
                if (hipMemcpyPeerAsync(rows_all + o * nT, devs[0], tr, J.device, sizeof(double) * (size_t)J.count * nT, nullptr) != hipSuccess ||
                    hipMemcpyPeerAsync(entry_all + o, devs[0], te, J.device, sizeof(double) * J.count, nullptr) != hipSuccess ||
                    hipMemcpyPeerAsync(ownw_all + o, devs[0], te + J.count, J.device, sizeof(double) * J.count, nullptr) != hipSuccess) rc = 2;
            }
            o += (size_t)J.count;
        }
        for (size_t d = 0; d < devs.size(); ++d) { (void)hipSetDevice(devs[d]); if (hipDeviceSynchronize() != hipSuccess) rc = rc ? rc : 2; }
        for (size_t d = 0; d < devs.size(); ++d) { (void)hipSetDevice(devs[d]); std::vector<void *> v; v.swap(scratch[d].v); for (void *p : v) pc_cache_dev_free(p); }
        // the runs' device blocks have been copied into the union's buffer (or the packing failed): unless the CALLER asked for them
        // (settings.device_records), they go back now, each with its own device current -- before the merge takes its working set
        if (records_are_mine)
            for (int k = 0; k < nseeds; ++k) {
                pchip_result &r = results[k];
                if (!r.d_records) continue;
                (void)hipSetDevice(r.records_device);
                pc_cache_dev_free(r.d_records);
                r.d_records = nullptr; r.n_records = 0; r.records_cap = 0;
            }
        (void)hipSetDevice(devs[0]);
        if (rc == 0) {
            std::vector<double> lz((size_t)nseeds), vz((size_t)nseeds);
            std::vector<int> cl((size_t)nseeds);
            for (int k = 0; k < nseeds; ++k) { lz[(size_t)k] = results[k].logZ; vz[(size_t)k] = results[k].varlogZ; cl[(size_t)k] = results[k].ncluster_peak > 1; }
            rc = pchip_merge_records_ex(s->nDims, s->nDerived, nseeds, counts.data(), rows_all, entry_all, ownw_all, lz.data(), vz.data(), cl.data(), 1, want_rows ? 1 : 0, merged);
        }
        else std::fprintf(stderr, "polychord_hip: run_repeats: packing the runs' records failed (%s)\n", hipGetErrorString(hipGetLastError()));
    }
    if (rc == 0) {
        merged->t_runs_s = t_runs;
        std::vector<double> zs((size_t)nseeds);
        for (int k = 0; k < nseeds; ++k) { merged->nlike += results[k].nlike; merged->ndead_all += results[k].ndead; zs[(size_t)k] = results[k].logZ; }
        runs_evidence(zs.data(), nseeds, std::sqrt(std::fabs(results[0].varlogZ)), merged);
    }
    else { for (int k = 0; k < nseeds; ++k) pchip_result_free(&results[k]); }        // (nothing is left for the caller to free on failure)
    return rc;
}

// ---- between processes (one per GPU): RCCL inside the library ---------------------------------------------------------
// The reference's farm exchanges babies with MPI point-to-point messages (mpi_utils.F90:376-463); here the only exchange of
// a repeat-sharded job is ONE all-gather of the runs' lived records at its end (counts first), over xGMI.
int pchip_comm_get_id(char *id128)
{
    if (!rccl().load()) { std::fprintf(stderr, "polychord_hip: librccl.so not found (PCHIP_RCCL_LIB names it)\n"); return 2; }
    ncclUniqueId id;
    const ncclResult_t e = rccl().GetUniqueId(&id);
    if (e != ncclSuccess) { std::fprintf(stderr, "polychord_hip: ncclGetUniqueId: %s\n", rccl().GetErrorString(e)); return 2; }
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int pchip_comm_create(const char *id128, int nranks, int rank, int device, pchip_comm **out)
{
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return 1;
    if (!rccl().load()) { std::fprintf(stderr, "polychord_hip: librccl.so not found (PCHIP_RCCL_LIB names it)\n"); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { std::fprintf(stderr, "polychord_hip: comm: device %d of %d\n", device, ndev); return 2; }
    if (hipSetDevice(device) != hipSuccess) return 2;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    pchip_comm *c = new pchip_comm;
    c->nranks = nranks; c->rank = rank; c->device = device;
    const ncclResult_t e = rccl().CommInitRank(&c->comm, nranks, id, rank);
    if (e != ncclSuccess) { std::fprintf(stderr, "polychord_hip: ncclCommInitRank: %s\n", rccl().GetErrorString(e)); delete c; return 2; }
    *out = c;
    return 0;
}

// A communicator over the CALLER's collective: `all_gather(user, send, recv, bytes)` must place every rank's `bytes` at `send` (device
// memory of `device`) into `recv` (device memory, nranks * bytes, rank after rank) on ALL ranks and return 0 when `recv` is complete.
// For hosts that bring their own transport -- a GPU-aware MPI_Allgather under the reference's MPI launcher (mpi_utils.F90), torch.distributed
// -- and for the tests, which drive pchip_comm_merge_many with several ranks on ONE GPU, where RCCL refuses to form a communicator.
int pchip_comm_create_with(pchip_allgather_fn all_gather, void *user, int nranks, int rank, int device, pchip_comm **out)
{
    *out = nullptr;
    if (!all_gather || nranks < 1 || rank < 0 || rank >= nranks) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { std::fprintf(stderr, "polychord_hip: comm: device %d of %d\n", device, ndev); return 2; }
    pchip_comm *c = new pchip_comm;
    c->nranks = nranks; c->rank = rank; c->device = device; c->fn = all_gather; c->user = user;
    *out = c;
    return 0;
}

void pchip_comm_destroy(pchip_comm *c)
{
    if (!c) return;
    if (c->comm) { (void)hipSetDevice(c->device); (void)rccl().CommDestroy(c->comm); }
    delete c;
}

const char *pchip_comm_library(void) { return rccl().load() ? rccl().where.c_str() : nullptr; }

int pchip_comm_merge(pchip_comm *c, const pchip_result *run, double logzero, int nDims, int nDerived, int want_rows, pchip_merged *out)
{
    return pchip_comm_merge_many(c, run, 1, logzero, nDims, nDerived, want_rows, out);
}

// words a rank sends about each of its runs ahead of the records: count, nlike, ndead, log Z, var log Z, did it ever hold several clusters?
#define META_W 6

int pchip_comm_merge_many(pchip_comm *c, const pchip_result *runs, int nruns, double logzero, int nDims, int nDerived, int want_rows,
                          pchip_merged *out)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    std::memset(out, 0, sizeof(*out));
    if (nruns < 1 || !runs) return 1;
    const int R = c ? c->nranks : 1, nT = 2 * nDims + nDerived + 2, me = c ? c->rank : 0;
    const bool coll = c && (c->comm || c->fn);
    // A rank that fails on its own (its records, its memory) still takes part in the exchange and says so there: the header's all-gather
    // carries -1 for it, a one-word all-gather behind the second phase's allocations its status, and every rank returns the error
    // together -- a rank that left early would leave the others waiting in ncclAllGather for ever.
    int local_err = 0;
    for (int j = 0; j < nruns; ++j)
        if (runs[j].ndead > 0 && runs[j].nTotal != nT) { std::fprintf(stderr, "polychord_hip: comm merge: the run's rows have %d columns, not %d\n", runs[j].nTotal, nT); local_err = 1; }
    int device = 0;
    if (c) device = c->device; else if (hipGetDevice(&device) != hipSuccess) { std::fprintf(stderr, "polychord_hip: no HIP device available -- the merge has no CPU path\n"); return 2; }
    if (hipSetDevice(device) != hipSuccess) { std::fprintf(stderr, "polychord_hip: no HIP device available -- the merge has no CPU path\n"); return 2; }
    hipStream_t st = nullptr;
    DevBuf B;
    auto fail = [&](const char *what, int code) { std::fprintf(stderr, "polychord_hip: comm merge: %s (%s)\n", what, hipGetErrorString(hipGetLastError())); return code; };

    std::vector<PackJob> J((size_t)nruns);
    long long mine_total = 0;
    for (int j = 0; j < nruns && !local_err; ++j) {
        J[(size_t)j].r = &runs[j]; J[(size_t)j].logzero = logzero; J[(size_t)j].device = device;
        if (runs[j].d_records && runs[j].records_device == device) J[(size_t)j].count = runs[j].n_records;      // (picked on the device when the run ended)
        else if (!J[(size_t)j].count_records(B, st)) local_err = fail("packing the run's records", 7);
        mine_total += J[(size_t)j].count;
    }
    if (local_err && !coll) return local_err;
    long long *d_status = nullptr;
    auto gather_words = [&](const long long *mine, int nw, std::vector<long long> &all, long long *d_buf, const char *what) -> int {
        // every rank's nw words -> all [R][nw]
        if (hipMemcpyAsync(d_buf, mine, sizeof(long long) * nw, hipMemcpyHostToDevice, st) != hipSuccess) return fail("upload", 2);
        if (const int e = comm_all_gather(c, d_buf, d_buf + nw, (size_t)nw, true, st, what)) return e;
        all.assign((size_t)nw * R, 0);
        if (hipMemcpyAsync(all.data(), d_buf + nw, sizeof(long long) * nw * R, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(what, 2);
        return 0;
    };
    // runs per rank (header), then META_W words per run
    std::vector<long long> hdr(1, local_err ? -1 : (long long)nruns), hdr_all;
    int nr_max = nruns;
    if (coll) {
        d_status = B.get<long long>((size_t)R + 1);
        if (!d_status) return fail("out of device memory", 7);      // (a few hundred bytes: a device in this state serves no collective either)
        if (const int e = gather_words(hdr.data(), 1, hdr_all, d_status, "ncclAllGather (header)")) return e;
        for (int q = 0; q < R; ++q) {
            if (hdr_all[(size_t)q] < 0) {
                if (q != me) std::fprintf(stderr, "polychord_hip: comm merge: rank %d could not pack its records\n", q);
                return local_err ? local_err : 7;
            }
            nr_max = std::max(nr_max, (int)hdr_all[(size_t)q]);
        }
    } else hdr_all = hdr;
    std::vector<long long> meta((size_t)META_W * nr_max, 0), meta_all;
    for (int j = 0; j < nruns; ++j) {
        long long *m = &meta[(size_t)META_W * j];
        const double z = runs[j].logZ, v = runs[j].varlogZ;
        m[0] = J[(size_t)j].count; m[1] = runs[j].nlike; m[2] = runs[j].ndead;
        std::memcpy(&m[3], &z, sizeof z); std::memcpy(&m[4], &v, sizeof v);
        m[5] = runs[j].ncluster_peak > 1;
    }
    if (coll) {
        long long *d_meta = B.get<long long>((size_t)META_W * nr_max * (R + 1));
        if (!d_meta) return fail("out of device memory", 7);
        if (const int e = gather_words(meta.data(), META_W * nr_max, meta_all, d_meta, "ncclAllGather (counts)")) return e;
    } else meta_all = meta;
    // the union's runs: rank 0's, then rank 1's, ...
    std::vector<long> counts;
    std::vector<double> lz, vz;
    std::vector<int> cl;
    std::vector<long long> off_rank(R + 1, 0);
    long long nmax = 1, nlike = 0, ndead_all = 0;
    for (int q = 0; q < R; ++q) {
        long long tot = 0;
        for (int j = 0; j < (int)hdr_all[(size_t)q]; ++j) {
            const long long *m = &meta_all[((size_t)q * nr_max + j) * META_W];
            counts.push_back((long)m[0]); tot += m[0]; nlike += m[1]; ndead_all += m[2];
            double z, v; std::memcpy(&z, &m[3], sizeof z); std::memcpy(&v, &m[4], sizeof v);
            lz.push_back(z); vz.push_back(v); cl.push_back((int)m[5]);
        }
        off_rank[q + 1] = off_rank[q] + tot; nmax = std::max(nmax, tot);
    }
    const int nruns_all = (int)counts.size();
    const size_t per = (size_t)nmax * (nT + 2);                 // one rank's block: rows [nmax][nT], entry [nmax], own log weight [nmax]
    double *send = B.get<double>(per);
    if (!send) local_err = fail("out of device memory", 7);
    else {
        long long o = 0;
        for (int j = 0; j < nruns && !local_err; ++j) {
            const pchip_result &r = runs[j];
            const long long cj = J[(size_t)j].count;
            if (r.d_records && r.records_device == device) {
                if (cj > 0 && (hipMemcpyAsync(send + (size_t)o * nT, r.d_records, sizeof(double) * (size_t)cj * nT, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                               hipMemcpyAsync(send + (size_t)nmax * nT + o, r.d_records + (size_t)r.records_cap * nT, sizeof(double) * cj, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                               hipMemcpyAsync(send + (size_t)nmax * (nT + 1) + o, r.d_records + (size_t)r.records_cap * (nT + 1), sizeof(double) * cj, hipMemcpyDeviceToDevice, st) != hipSuccess))
                    local_err = fail("copying the run's records", 2);
            }
            else if (!J[(size_t)j].scatter(B, send + (size_t)o * nT, send + (size_t)nmax * nT + o, send + (size_t)nmax * (nT + 1) + o, st)) local_err = fail("packing the run's records", 7);
            o += cj;
        }
    }
    if (local_err && !coll) return local_err;
    const double *rows_all = send, *entry_all = send + (size_t)nmax * nT, *ownw_all = send + (size_t)nmax * (nT + 1);
    if (coll) {
        double *recv = local_err ? nullptr : B.get<double>(per * R);
        long long *d_off = local_err ? nullptr : B.get<long long>(R + 1);
        const long long ntot = off_rank[R];
        const size_t na = (size_t)std::max<long long>(ntot, 1);
        double *ra = local_err ? nullptr : B.get<double>(na * nT), *ea = local_err ? nullptr : B.get<double>(2 * na);
        if (!local_err && (!recv || !d_off || !ra || !ea)) local_err = fail("out of device memory", 7);
        {   // the ranks' status words, all-gathered: go on only if every rank is fine
            std::vector<long long> w(1, local_err), all;
            if (const int e = gather_words(w.data(), 1, all, d_status, "ncclAllGather (status)")) return e;
            int worst = 0;
            for (int q = 0; q < R; ++q) if (all[(size_t)q] != 0 && !worst) { worst = (int)all[(size_t)q]; if (q != me) std::fprintf(stderr, "polychord_hip: comm merge: rank %d failed (code %d)\n", q, worst); }
            if (worst) return local_err ? local_err : worst;
        }
        if (const int e = comm_all_gather(c, send, recv, per, false, st, "ncclAllGather (records)")) return e;       // the exchange: one padded block per rank over xGMI
        if (hipMemcpyAsync(d_off, off_rank.data(), sizeof(long long) * (R + 1), hipMemcpyHostToDevice, st) != hipSuccess) return fail("upload", 2);
        if (ntot > 0) hipLaunchKernelGGL(k_unpad, dim3((unsigned)ntot), dim3(64), 0, st, (const double *)recv, nmax, nT, R, (const long long *)d_off, ra, ea, ea + na);
        rows_all = ra; entry_all = ea; ownw_all = ea + na;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail("exchange", 2);
    const int rc = pchip_merge_records_ex(nDims, nDerived, nruns_all, counts.data(), rows_all, entry_all, ownw_all, lz.data(), vz.data(), cl.data(), 1, want_rows, out);
    if (rc == 0) {
        out->nlike = (long)nlike; out->ndead_all = (long)ndead_all; out->t_merge_s = std::chrono::duration<double>(clk::now() - t0).count();
        runs_evidence(lz.data(), nruns_all, std::sqrt(std::fabs(vz[0])), out);
    }
    return rc;
}

}  // extern "C"
