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

struct DevBuf {
    std::vector<void *> v;
    template <class T> T *get(size_t n) { void *p = nullptr; if (hipMalloc(&p, sizeof(T) * (n ? n : 1)) != hipSuccess) { (void)hipGetLastError(); return nullptr; } v.push_back(p); return (T *)p; }
    ~DevBuf() { for (void *p : v) (void)hipFree(p); }
};

}  // namespace

extern "C" {

void pchip_merged_free(pchip_merged *m)
{
    if (!m) return;
    if (m->rows) (void)hipHostFree(m->rows);
    std::free(m->logweights); std::free(m->nlive); std::free(m->post_mean); std::free(m->post_var);
    std::memset(m, 0, sizeof(*m));
}

int pchip_merge_records(int nDims, int nDerived, int nruns, const long *counts, const double *rows, const double *entry,
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
    const double *d_rows = rows, *d_entry = entry;
    if (!on_device) {
        double *r = B.get<double>((size_t)n * nT), *e = B.get<double>(n);
        if (!r || !e) return fail("out of device memory");
        if (hipMemcpy(r, rows, sizeof(double) * (size_t)n * nT, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
        if (hipMemcpy(e, entry, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
        d_rows = r; d_entry = e;
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
    std::vector<P2> hA(nbt), hB(nbt); std::vector<double> hM(nbt);
    if (hipMemcpy(hA.data(), pA, sizeof(P2) * nbt, hipMemcpyDeviceToHost) != hipSuccess) return fail("evidence kernels");
    (void)hipMemcpy(hB.data(), pB, sizeof(P2) * nbt, hipMemcpyDeviceToHost); (void)hipMemcpy(hM.data(), pM, sizeof(double) * nbt, hipMemcpyDeviceToHost);
    auto comb = [](P2 x, P2 y) { const double e = std::exp(-std::fabs(x.a - y.a)); return P2{std::max(x.a, y.a), x.a >= y.a ? x.b + y.b * e : x.b * e + y.b}; };
    P2 a = hA[0], b = hB[0]; double wmax = hM[0];
    for (int k = 1; k < nbt; ++k) { a = comb(a, hA[k]); b = comb(b, hB[k]); wmax = std::max(wmax, hM[k]); }
    const double lZ = a.a + std::log(a.b), lZ2 = b.a + std::log(b.b);       // log <Z>, log <Z^2>
    out->logZ = 2.0 * lZ - 0.5 * lZ2; out->varlogZ = lZ2 - 2.0 * lZ;       // run_time_info.f90:652-678
    hipLaunchKernelGGL(k_merge_moments, dim3(nbm), dim3(256), 0, st, M, (const double *)d_logw, wmax, p0, nP, pmom);
    std::vector<double> hm((size_t)nbm * (2 * nP + 1));
    if (hipMemcpy(hm.data(), pmom, sizeof(double) * hm.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail("moment kernel");
    double sw = 0.0;
    for (int k = 0; k < nbm; ++k) {
        const double *p = hm.data() + (size_t)k * (2 * nP + 1);
        sw += p[2 * nP];
        for (int c = 0; c < nP; ++c) { out->post_mean[c] += p[c]; out->post_var[c] += p[nP + c]; }
    }
    for (int c = 0; c < nP; ++c) { out->post_mean[c] /= sw; out->post_var[c] = out->post_var[c] / sw - out->post_mean[c] * out->post_mean[c]; }
    out->logweights = (double *)std::malloc(sizeof(double) * n); out->nlive = (int *)std::malloc(sizeof(int) * n);
    (void)hipMemcpy(out->logweights, d_logw, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(out->nlive, M.nl, sizeof(int) * n, hipMemcpyDeviceToHost);
    if (want_rows) {
        double *d_out = B.get<double>((size_t)n * nT);
        if (!d_out) return fail("out of device memory");
        hipLaunchKernelGGL(k_merge_gather, dim3((unsigned)n), dim3(64), 0, st, M, b0, d_out);
        if (hipHostMalloc((void **)&out->rows, sizeof(double) * (size_t)n * nT) != hipSuccess) return fail("out of pinned memory");
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
int pchip_run_repeats(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds,
                      int ndevices, const int *devices, int max_in_flight, pchip_result *results, pchip_merged *merged)
{
    using clk = std::chrono::steady_clock;
    if (nseeds < 1) return 1;
    int ndev_all = 0;
    if (hipGetDeviceCount(&ndev_all) != hipSuccess || ndev_all == 0) { std::fprintf(stderr, "polychord_hip: no HIP device available -- this engine has no CPU path\n"); return 2; }
    std::vector<int> devs;
    if (ndevices > 0 && devices) { for (int k = 0; k < ndevices; ++k) { if (devices[k] < 0 || devices[k] >= ndev_all) { std::fprintf(stderr, "polychord_hip: device %d of %d\n", devices[k], ndev_all); return 1; } devs.push_back(devices[k]); } }
    else devs.push_back(s->device >= 0 ? s->device % ndev_all : 0);
    const int per_dev = std::max(1, max_in_flight), nworkers = std::min(nseeds, per_dev * (int)devs.size());
    std::atomic<int> next{0}, worst{0};
    const auto t0 = clk::now();
    auto work = [&](int wid) {
        const int dev = devs[wid % devs.size()];
        for (int k = next.fetch_add(1); k < nseeds; k = next.fetch_add(1)) {
            pchip_settings c = *s;
            c.seed = seeds[k]; c.device = dev;
            const int rc = pchip_run_hooks(&c, like, prior, nullptr, &results[k]);
            if (rc != 0) { int z = 0; worst.compare_exchange_strong(z, rc); }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nworkers; ++w) th.emplace_back(work, w);
    work(0);
    for (auto &t : th) t.join();
    const double t_runs = std::chrono::duration<double>(clk::now() - t0).count();
    if (worst.load() != 0) { for (int k = 0; k < nseeds; ++k) pchip_result_free(&results[k]); return worst.load(); }
    if (!merged) return 0;
    // the union of the points that entered a live set (failed spawns carry logweight = logzero), run after run
    const int nT = results[0].nTotal, l0 = nT - 1;
    std::vector<long> counts(nseeds, 0);
    size_t ntot = 0;
    for (int k = 0; k < nseeds; ++k) { for (long i = 0; i < results[k].ndead; ++i) counts[k] += results[k].logweights[i] > s->logzero; ntot += (size_t)counts[k]; }
    std::vector<double> rows(ntot * nT), entry(ntot);
    size_t o = 0;
    for (int k = 0; k < nseeds; ++k)
        for (long i = 0; i < results[k].ndead; ++i)
            if (results[k].logweights[i] > s->logzero) { std::memcpy(rows.data() + o * nT, results[k].dead + (size_t)i * nT, sizeof(double) * nT); entry[o] = results[k].entry[i]; o++; }
    (void)l0;
    (void)hipSetDevice(devs[0]);
    const int rc = pchip_merge_records(s->nDims, s->nDerived, nseeds, counts.data(), rows.data(), entry.data(), 0, 1, merged);
    if (rc == 0) { merged->t_runs_s = t_runs; for (int k = 0; k < nseeds; ++k) { merged->nlike += results[k].nlike; merged->ndead_all += results[k].ndead; } }
    return rc;
}

}  // extern "C"
