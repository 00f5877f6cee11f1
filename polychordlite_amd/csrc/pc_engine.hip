// pc_engine.hip -- host side of the MI355X nested-sampling engine.
//
// Replaces the administrator loop of NestedSampling (src/polychord/nested_sampling.F90:15-510) and
// the MPI live-point farm (mpi_utils.F90:322-600): the host only enqueues kernels and reads one
// small control block per round; every point, every evidence accumulator and every decision lives
// on the device.  One round = [K0 directions | K1 slice chains] (when the nursery is empty)
// + [K2 consume | K3 apply] + (on an update) [clean phantoms | covariance + Cholesky].
#include "pc_state.h"
#include "pc_resume.h"
#include "../../include/polychord_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <memory>
#include <string>
#include <chrono>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <atomic>
#include <mutex>
#include <thread>
#include <functional>
#include <exception>
#include <ucontext.h>
#include <array>
#include <sys/mman.h>
#include <unistd.h>

extern "C" size_t pc_records_block_bytes(long long cap, int nT);
extern "C" int pc_pack_lived_device(const double *, const double *, const double *, long long, int, double, double *, long long, long long *, hipStream_t);
extern "C" void *pc_cache_dev_alloc(size_t bytes);
extern "C" void pc_cache_dev_free(void *p);
extern "C" {
int pc_launch_generate_live(const PcState *, int, int, double *, double *, hipStream_t);
int pc_launch_nhats(const PcState *, unsigned, int, hipStream_t);
int pc_nhats_splittable(const PcState *);
int pc_launch_nhats_part(const PcState *, unsigned, int, int, hipStream_t, int);
int pc_launch_slice(const PcState *, unsigned, int, hipStream_t);
int pc_slice_fusable(const PcState *);
int pc_launch_slice_fused(const PcState *, unsigned, int, hipStream_t);
int pc_slice_t_ok(const PcState *, int);
int pc_launch_slice_many(const PcState *, const PcManyRec *, int, int, int, hipStream_t);
int pc_launch_nhats_many(const PcState *, const PcManyRec *, int, int, hipStream_t);
int pc_launch_nn_lists_many(const PcState *, const PcManyRec *, int, int, int, hipStream_t);
int pc_launch_consume_cl_many(const PcState *, const PcManyRec *, int, int, hipStream_t);
int pc_launch_reset_thresholds_many(const PcState *, const PcManyRec *, int, hipStream_t);
int pc_launch_knn_cluster_batch_many(const PcState *, const PcManyRec *, int, int, int, hipStream_t);
int pc_launch_knn_cluster_batch_dev(const PcState *, const int *, int, int, double *, int *, int *, int *, hipStream_t);
int pc_launch_knn_cluster_sub(const int *, int, int, const double *, const int *, int *, int *, int *, hipStream_t);
int pc_launch_knn_cluster_sub_many(const PcManyRec *, int, int, int, hipStream_t);
int pc_launch_slice_t(const PcState *, unsigned, int, hipStream_t);
int pc_bases_t_ok(const PcState *);
int pc_launch_slice_t_many(const PcState *, const PcManyRec *, int, unsigned, int, hipStream_t);
int pc_launch_bases_t_many(const PcState *, const PcManyRec *, int, unsigned, int, hipStream_t);
int pc_update_fused_grid(const PcState *, int, int);
int pc_launch_clean_many(const PcManyRec *, int, int, hipStream_t);
int pc_launch_final_par_many(const PcManyRec *, int, hipStream_t);
int pc_launch_sort_live_many(const PcState *, const PcManyRec *, int, hipStream_t);
int pc_launch_consume_par_many(const PcState *, const PcManyRec *, int, hipStream_t);
int pc_launch_apply_many(const PcState *, const PcManyRec *, int, unsigned, int, hipStream_t);
int pc_launch_update_fused_many(const PcState *, const PcManyRec *, int, int, int, int, hipStream_t);
int pc_launch_consume(const PcState *, int, int, hipStream_t);
void pc_launch_nn_lists(const PcState *, int, int, hipStream_t);
void pc_launch_shift_mats(const PcState *, int, int, hipStream_t);
void pc_launch_remap_chains(const PcState *, const int *, int, int, hipStream_t);
int pc_launch_consume_fast(const PcState *, int, hipStream_t);
int pc_fast_fits(const PcState *);
int pc_par_fits(const PcState *);
int pc_launch_sort_live(const PcState *, hipStream_t);
int pc_launch_consume_par(const PcState *, hipStream_t);
int pc_launch_final_par(const PcState *, hipStream_t);
int pc_consume_cl_fits(const PcState *, int);
int pc_consume_clp_fits(const PcState *, int);
int pc_launch_consume_cl(const PcState *, int, hipStream_t);
int pc_launch_killoff_cl(const PcState *, int, hipStream_t);
void pc_launch_ph_prepare(const PcState *, hipStream_t);
void pc_launch_apply(const PcState *, unsigned, int, hipStream_t);
void pc_launch_install_live(const PcState *, const double *, int, hipStream_t);
void pc_launch_clean(const PcState *, int, unsigned char *, int *, int *, double *, double *, unsigned *,
                     unsigned long long *, int *, hipStream_t);
void pc_launch_reset_thresholds(const PcState *, hipStream_t);
int pc_cov_nchunk(const PcState *, int);
void pc_launch_similarity(const PcState *, const int *, int, double *, hipStream_t);
int pc_launch_knn_cluster(const double *, int, const int *, int, int *, int *, int *, hipStream_t);
int pc_launch_knn_cluster_batch(const PcState *, const int *, const int *, int, double *, int *, int *, int *, hipStream_t);
void pc_launch_rebuild(const PcState *, int, hipStream_t);
void pc_launch_ph_rehome(const PcState *, int, int, const unsigned *, int, int *, hipStream_t);
void pc_launch_slice_tick(const PcState *, unsigned, int, void *, double *, int *, double *, const double *, const double *,
                          const double *, int, double *, int *, hipStream_t);
size_t pc_chain_state_size(void);
void pc_launch_init_state(const PcState *, double, hipStream_t);
int pc_post_blocks(void);
void pc_launch_post_moments(const PcState *, int, double *, double *, hipStream_t);
int pc_launch_covmats(const PcState *, int, int, double *, int *, double *, int *, double *, hipStream_t);
int pc_update_fused_ok(const PcState *, int);
int pc_update_fused_blocks(const PcState *, int);
int pc_update_fused_entries(const PcState *);
void pc_launch_update_fused(const PcState *, int, unsigned char *, int *, int *, double *, double *, unsigned *, unsigned long long *, double *, double *, int, hipStream_t);
}

// Fatal conditions unwind to the C ABI entry points (pchip_run_hooks, pchip_slice_chains), which release the run's
// resources, print the message and return the code: the reference's convention is message + `stop 1`
// (abort.F90:19-29), and the process-level half of it belongs to the caller of the C ABI (polychord_c_interface exits;
// a language binding raises) -- an engine inside somebody's interpreter must not take the process down itself.
enum { PC_RC_SETTINGS = 1, PC_RC_DEVICE = 2, PC_RC_NDIMS = 3, PC_RC_LDS = 4, PC_RC_STOPPED = 5, PC_RC_RESUME = 6, PC_RC_MEMORY = 7, PC_RC_LIMIT = 8 };
struct EngineError {
    int code; std::string msg;
};
[[noreturn]] static void engine_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
[[noreturn]] static void engine_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); std::vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw EngineError{code, buf};
}
static std::atomic<int> g_cap_clusters{128}, g_cap_phantoms{0};   // initial capacities (both grow on demand); tests shrink them
static std::atomic<int> g_inject_fault{0};           // polychord_hip_set_option("inject_fault", k): tests of the error paths, one-shot
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    engine_fail(PC_RC_DEVICE, "HIP error %s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// polychord_hip_set_batch_callback / polychord_hip_request_stop are process-level calls of the C ABI (no run handle, like
// the reference's setup_loglikelihood state).  A run takes its OWN copy of the batch callback when it is set up and owns
// its OWN stop flag; a stop request raises the flag of every run in flight at that moment (the registry below) and of no
// later one -- runs on other threads (pchip_run_repeats) neither lose a request nor inherit a stale one.
static std::mutex g_cb_mutex;
static polychord_batch_fn g_batch_fn = nullptr;     // polychord_hip_set_batch_callback
static void *g_batch_user = nullptr;
static std::mutex g_run_mutex;
static std::vector<std::atomic<int> *> g_run_stop;  // stop flags of the runs in flight
extern "C" double polychord_hip_keyed_uniform(unsigned seed, unsigned dom, unsigned shi, unsigned slo, unsigned idx);

namespace {

// Process-wide cache of device blocks and pinned host blocks.  A run allocates ~70 buffers; hipMalloc /
// hipFree cost O(100 us) each and hipFree synchronises the device, which is as long as the sampling
// itself at the metric config.  Blocks are keyed by (device, rounded size) and handed back on free;
// nothing relies on their contents (hipMalloc does not zero either).
static std::atomic<long long> g_dbg_miss_n[2], g_dbg_miss_ns[2], g_dbg_mk_stream_n{0}, g_dbg_mk_stream_ns{0};      // (PC_DEBUG=5: trips to the driver -- device / pinned blocks, streams)
struct BlockCache {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void *> free_;
    std::unordered_map<void *, std::pair<int, size_t>> owner;
    size_t cached = 0, limit;
    bool host;
    BlockCache(bool h, size_t lim) : limit(lim), host(h) {}
    // eight sizes per octave above 4 KB (at most an eighth more than asked for): arrays sized by what a run found -- its dead
    // points, its weights -- differ by a few rows from seed to seed, and every near miss was a trip to the driver (a pinned
    // allocation is milliseconds there once another runtime, PyTorch's, lives in the process)
    static size_t round_up(size_t b)
    {
        if (b < 4096) return (b + 255) & ~(size_t)255;
        int top = 63 - __builtin_clzll((unsigned long long)b);
        const size_t step = (size_t)1 << (top - 3);
        return (b + step - 1) & ~(step - 1);
    }
    void *get(size_t bytes)
    {
        int dev = 0;
        if (!host) (void)hipGetDevice(&dev);
        const size_t sz = round_up(bytes ? bytes : 1);
        if (!host && sz >= (256u << 10) && g_inject_fault.load() == 1) { g_inject_fault = 0; engine_fail(PC_RC_MEMORY, "out of device memory (%zu bytes): injected", sz); }
        {
            std::lock_guard<std::mutex> g(m);
            auto it = free_.find({dev, sz});
            if (it != free_.end()) { void *p = it->second; free_.erase(it); cached -= sz; return p; }
        }
        void *p = nullptr;
        const auto q0 = std::chrono::steady_clock::now();
        hipError_t e = host ? hipHostMalloc(&p, sz) : hipMalloc(&p, sz);
        g_dbg_miss_n[host ? 1 : 0]++; g_dbg_miss_ns[host ? 1 : 0] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - q0).count();
        if (e != hipSuccess) {          // give the cached blocks back and retry once
            trim();
            e = host ? hipHostMalloc(&p, sz) : hipMalloc(&p, sz);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); engine_fail(PC_RC_MEMORY, "out of %s memory (%zu bytes): %s", host ? "pinned host" : "device", sz, hipGetErrorString(e)); }
        std::lock_guard<std::mutex> g(m);
        owner[p] = {dev, sz};
        return p;
    }
    bool put(void *p)                   // false: not one of ours
    {
        {
            std::lock_guard<std::mutex> g(m);
            auto it = owner.find(p);
            if (it == owner.end()) return false;
            if (cached + it->second.second <= limit) { free_.insert({it->second, p}); cached += it->second.second; return true; }
            owner.erase(it);
        }
        if (host) (void)hipHostFree(p); else (void)hipFree(p);      // (over the limit: back to the driver, and not under the lock)
        return true;
    }
    // a run's hundred-odd blocks at its end in one visit: eight threads tearing runs down at once spent most of that time queueing
    // for the lock, block by block
    void put_many(const std::vector<void *> &ps)
    {
        std::vector<void *> over;
        {
            std::lock_guard<std::mutex> g(m);
            for (void *p : ps) {
                auto it = owner.find(p);
                if (it == owner.end()) continue;
                if (cached + it->second.second <= limit) { free_.insert({it->second, p}); cached += it->second.second; }
                else { owner.erase(it); over.push_back(p); }
            }
        }
        for (void *p : over) { if (host) (void)hipHostFree(p); else (void)hipFree(p); }
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(m);
        for (auto &kv : free_) { if (host) (void)hipHostFree(kv.second); else (void)hipFree(kv.second); owner.erase(kv.second); }
        free_.clear(); cached = 0;
    }
};
// never destroyed: the HIP runtime may already be gone when static destructors run
BlockCache &dcache() { static BlockCache *c = new BlockCache(false, (size_t)64 << 30); return *c; }
BlockCache &hcache() { static BlockCache *c = new BlockCache(true, (size_t)12 << 30); return *c; }

// PC_POISON=1 (tests): every device block is filled with 0x5A bytes when it is handed out -- ints become 1515870810,
// doubles 2.6e127 -- so that a buffer some path forgets to initialise fails loudly instead of working by the luck of
// what the previous run left in it
template <class T> T *dalloc(size_t n)
{
    static const bool poison = std::getenv("PC_POISON") != nullptr;
    T *p = (T *)dcache().get(sizeof(T) * (n ? n : 1));
    if (poison) { (void)hipMemset((void *)p, 0x5A, sizeof(T) * (n ? n : 1)); (void)hipDeviceSynchronize(); }
    return p;
}
static thread_local std::vector<void *> *tl_dfree_batch = nullptr;      // (Engine::destroy: the blocks are collected and given back together)
template <class T> void dfree(T *&p) { if (p) { if (tl_dfree_batch) tl_dfree_batch->push_back((void *)p); else dcache().put((void *)p); } p = nullptr; }
template <class T> T *halloc(size_t n) { return (T *)hcache().get(sizeof(T) * (n ? n : 1)); }
void hfree(void *p) { if (p && !hcache().put(p)) std::free(p); }

// streams and events are pooled for the same reason (creation / destruction synchronise with the driver)
struct HandlePool {
    std::mutex m;
    std::vector<std::pair<int, hipStream_t>> streams;
    std::vector<std::pair<int, hipEvent_t>> events;
    hipStream_t get_stream()
    {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < streams.size(); ++i)
                if (streams[i].first == dev) { hipStream_t s = streams[i].second; streams.erase(streams.begin() + i); return s; }
        }
        const auto q0 = std::chrono::steady_clock::now();
        hipStream_t s; HIPCHK(hipStreamCreate(&s));
        g_dbg_mk_stream_n++; g_dbg_mk_stream_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - q0).count();
        return s;
    }
    void put_stream(hipStream_t s) { int dev = 0; (void)hipGetDevice(&dev); std::lock_guard<std::mutex> g(m); streams.push_back({dev, s}); }
    template <class Pred> hipStream_t take_stream_if(Pred pred)          // a pooled stream of this device the predicate accepts, or null
    {
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(m);
        for (size_t i = 0; i < streams.size(); ++i)
            if (streams[i].first == dev && pred(streams[i].second)) { hipStream_t s = streams[i].second; streams.erase(streams.begin() + i); return s; }
        return nullptr;
    }
    hipEvent_t get_event()
    {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = events.size(); i-- > 0;)
                if (events[i].first == dev) { hipEvent_t e = events[i].second; events.erase(events.begin() + i); return e; }
        }
        hipEvent_t e; HIPCHK(hipEventCreate(&e)); return e;
    }
    void put_event(hipEvent_t e) { int dev = 0; (void)hipGetDevice(&dev); std::lock_guard<std::mutex> g(m); events.push_back({dev, e}); }
    // events that only order streams (no timestamps: a cheaper record)
    std::vector<std::pair<int, hipEvent_t>> sync_events;
    hipEvent_t get_sync_event()
    {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = sync_events.size(); i-- > 0;)
                if (sync_events[i].first == dev) { hipEvent_t e = sync_events[i].second; sync_events.erase(sync_events.begin() + i); return e; }
        }
        hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return e;
    }
    void put_sync_event(hipEvent_t e) { int dev = 0; (void)hipGetDevice(&dev); std::lock_guard<std::mutex> g(m); sync_events.push_back({dev, e}); }
};
HandlePool &hpool() { static HandlePool *p = new HandlePool; return *p; }

// A second stream is worth something only if the device runs it NEXT TO the first: the runtime deals streams onto a few hardware
// queues (GPU_MAX_HW_QUEUES, four by default), and two streams of one queue take turns -- after a sweep of sixty-four runs in
// step the pool held dozens of streams, and a single run that drew two of one queue was 2.3 ms (16 %) slower.  So a side stream
// is tried against its main stream once (two 40-us spinning kernels: together or one after the other?) and the verdict kept.
// ---- the small copies of runs in step, many at a time.  An update with clustering reads a dozen small arrays back and sends a dozen
// down (add_cluster: counts, evidences, the cross-volume matrix, labels); as hipMemcpyAsync calls that was 49 000 copies for sixteen
// 10-D Rastrigin runs (round 4) -- ~6 us of the driving thread each, and on the stream one after the other, ~4 us each, in front of
// every shared wait.  Pinned host memory is mapped into the device's address space (hipHostMalloc), so ONE kernel makes up to PC_COPY_N
// of them, either way, a workgroup per piece of at most 4 KB: its arguments are the (source, destination, bytes) of the pieces.
#define PC_COPY_N 96
#define PC_COPY_PIECE 4096u
struct PcCopyBatch { const void *src[PC_COPY_N]; void *dst[PC_COPY_N]; unsigned bytes[PC_COPY_N]; };
__global__ __launch_bounds__(256) void k_copy_batch(PcCopyBatch b)
{
    const int i = blockIdx.x;
    const unsigned n = b.bytes[i];
    const char *s = (const char *)b.src[i]; char *d = (char *)b.dst[i];
    if ((((uintptr_t)s | (uintptr_t)d) & 7u) == 0u) {
        const unsigned n8 = n >> 3;
        for (unsigned k = threadIdx.x; k < n8; k += 256) ((unsigned long long *)d)[k] = ((const unsigned long long *)s)[k];
        for (unsigned k = (n8 << 3) + threadIdx.x; k < n; k += 256) d[k] = s[k];
    } else for (unsigned k = threadIdx.x; k < n; k += 256) d[k] = s[k];
}
static void pc_copy_many(const std::vector<std::array<uintptr_t, 3>> &reqs, hipStream_t q)      // {dst, src, bytes}
{
    PcCopyBatch b; int n = 0;
    auto go = [&] { if (n) { hipLaunchKernelGGL(k_copy_batch, dim3(n), dim3(256), 0, q, b); n = 0; } };
    for (const auto &r : reqs) {
        size_t left = r[2]; uintptr_t d = r[0], s = r[1];
        while (left) {
            const unsigned c = (unsigned)std::min<size_t>(left, PC_COPY_PIECE);
            b.dst[n] = (void *)d; b.src[n] = (const void *)s; b.bytes[n] = c; ++n;
            d += c; s += c; left -= c;
            if (n == PC_COPY_N) go();
        }
    }
    go();
}

__global__ void k_engine_spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
static bool streams_overlap_test(hipStream_t a, hipStream_t b)
{
    auto both = [&] {
        (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_engine_spin, dim3(1), dim3(64), 0, a, 10000LL);              // 100 us of the 100 MHz clock
        hipLaunchKernelGGL(k_engine_spin, dim3(1), dim3(64), 0, b, 10000LL);
        (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    static std::atomic<bool> warmed{false};
    if (!warmed.exchange(true)) (void)both();                // (the kernel's first launch loads its code)
    const double two = std::min(both(), both());
    static const bool dbg = std::getenv("PC_DEBUG") && std::atoi(std::getenv("PC_DEBUG")) == 5;
    if (dbg) std::fprintf(stderr, "polychord_hip dbg streams %p %p: both %.1f us\n", (void *)a, (void *)b, two * 1e6);
    return two < 170e-6;                                    // side by side: ~120 us (one spin and the launches); in turn: ~225 us
}
// streams that take turns are streams of one hardware queue: every stream is put into its class once (one test against a
// member of each class known so far), and two streams run side by side when their classes differ
struct StreamClasses {
    std::mutex mm;
    std::map<void *, int> cls;
    std::vector<hipStream_t> reps;
    int known(hipStream_t x) { std::lock_guard<std::mutex> g(mm); auto it = cls.find((void *)x); return it == cls.end() ? -1 : it->second; }
    int classify(hipStream_t x)
    {
        std::lock_guard<std::mutex> g(mm);
        auto it = cls.find((void *)x);
        if (it != cls.end()) return it->second;
        int c = -1;
        for (size_t r = 0; r < reps.size() && c < 0; ++r) if (!streams_overlap_test(reps[r], x)) c = (int)r;
        if (c < 0) { c = (int)reps.size(); reps.push_back(x); }
        cls[(void *)x] = c;
        return c;
    }
};
static StreamClasses &sclasses() { static StreamClasses *p = new StreamClasses; return *p; }
// a pooled (or new) stream that runs next to all of `others` (null entries ignored): one whose class is known first, then the
// pool's unknown ones, classed as they come; the ones found wanting go back to the pool
static hipStream_t stream_avoiding(std::vector<int> avoid, bool known_only = false);
static hipStream_t stream_beside(std::initializer_list<hipStream_t> others, bool known_only = false)
{
    static const bool off = std::getenv("PC_SIDE_PICK_OFF") != nullptr;
    if (off) return hpool().get_stream();
    std::vector<int> avoid;
    for (hipStream_t o : others) if (o) { const int c = known_only ? sclasses().known(o) : sclasses().classify(o); if (c >= 0) avoid.push_back(c); }
    return stream_avoiding(avoid, known_only);
}
// ... next to every stream of the hardware-queue classes in `avoid`.  known_only: the device is at work (another scheduler group's runs):
// a class test now -- two spin kernels timed side by side -- would say "one queue" for whatever pair it is given, and the wrong class would
// stay with the stream; only streams classed earlier (pc_prepare_streams) are looked at, the best of them taken
static hipStream_t stream_avoiding(std::vector<int> avoid, bool known_only)
{
    auto fits = [&](int c) { return c >= 0 && std::find(avoid.begin(), avoid.end(), c) == avoid.end(); };
    if (hipStream_t k = hpool().take_stream_if([&](hipStream_t x) { return fits(sclasses().known(x)); })) return k;
    if (known_only) {
        if (hipStream_t k = hpool().take_stream_if([&](hipStream_t x) { return sclasses().known(x) >= 0; })) return k;
        return hpool().get_stream();
    }
    std::vector<hipStream_t> tried;
    hipStream_t pick = nullptr;
    for (int k = 0; k < 8 && !pick; ++k) {
        hipStream_t c = hpool().take_stream_if([&](hipStream_t x) { return sclasses().known(x) < 0; });
        static const bool dbg = std::getenv("PC_DEBUG") && std::atoi(std::getenv("PC_DEBUG")) == 5;
        const auto q0 = std::chrono::steady_clock::now();
        const bool made = !c;
        if (!c) { HIPCHK(hipStreamCreate(&c)); }
        const auto q1 = std::chrono::steady_clock::now();
        const int cc = sclasses().classify(c);
        if (dbg) std::fprintf(stderr, "polychord_hip dbg stream_beside: try %d, %s stream %p (%.2f ms), class %d (%.2f ms), avoid %d\n", k, made ? "new" : "pooled", (void *)c, std::chrono::duration<double>(q1 - q0).count() * 1e3, cc, std::chrono::duration<double>(std::chrono::steady_clock::now() - q1).count() * 1e3, avoid.empty() ? -1 : avoid[0]);
        if (fits(cc)) pick = c; else tried.push_back(c);
    }
    if (!pick) { pick = tried.back(); tried.pop_back(); }   // (none: any will do)
    for (hipStream_t t : tried) hpool().put_stream(t);
    return pick;
}
static hipStream_t side_stream_for(hipStream_t main_st) { return stream_beside({main_st}); }
// The hardware-queue classes the scheduler groups at work on a device have taken for their main and side streams.  Two groups side by
// side (clustered runs: pchip_run_repeats) each picked their streams from the pool on their own, and now and then both main streams sat
// on ONE hardware queue: their kernels took turns, sixteen twin-Gaussian runs 590 ms instead of 370 -- which of the two a process got
// depended on what it had done before (a merge, a torch.cuda.synchronize: whatever moved the pool's order).  A group now takes streams
// of classes no other group of its device holds, while there are any.
struct CohortStreams {
    std::mutex m;
    std::vector<std::pair<int, int>> used;      // (device, class; main streams' classes carry + 1000)
    std::vector<int> busy(int dev, bool mains_only = false)
    {
        std::vector<int> b;
        for (auto &u : used) if (u.first == dev && (!mains_only || u.second >= 1000)) b.push_back(u.second % 1000);
        return b;
    }
    void take(int dev, int c, bool main_stream = false) { if (c >= 0) used.emplace_back(dev, c + (main_stream ? 1000 : 0)); }
    // (the exact entry that was taken: by class alone a group handing back its SIDE stream's class could erase another group's MAIN entry of the
    //  same class, and busy(dev, mains_only) then misreports which hardware queues hold main streams)
    void give(int dev, int c, bool main_stream = false)
    {
        if (c < 0) return;
        const int v = c + (main_stream ? 1000 : 0);
        for (size_t i = 0; i < used.size(); ++i) if (used[i].first == dev && used[i].second == v) { used.erase(used.begin() + (long)i); return; }
    }
};
static CohortStreams &cstreams() { static CohortStreams *p = new CohortStreams; return *p; }
// what a scheduler group holds of the table above, given back when the group is done -- or when anything between the take and that point throws
struct CohortLease {
    int dev, cls_main = -1, cls_side = -1; bool held = false;
    explicit CohortLease(int d) : dev(d) {}
    void hold(int cm, int cs) { cls_main = cm; cls_side = cs; held = true; }      // (the caller holds cstreams().m: take() has just been called)
    void release() { if (!held) return; held = false; std::lock_guard<std::mutex> gq(cstreams().m); cstreams().give(dev, cls_main, true); cstreams().give(dev, cls_side); }
    ~CohortLease() { release(); }
};
// called before several scheduler groups start on a device, while it is idle: the pool gets at least `want` streams whose hardware-queue
// class is known (a group takes four: main, side, two for copies), so that no group has to class a stream while another group's kernels run
extern "C" void pc_prepare_streams(int dev, int want)
{
    if (hipSetDevice(dev) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> gq(cstreams().m);
    std::vector<hipStream_t> have;
    while (hipStream_t k = hpool().take_stream_if([](hipStream_t x) { return sclasses().known(x) >= 0; })) have.push_back(k);
    for (int tries = 0; (int)have.size() < want && tries < 4 * want; ++tries) {
        hipStream_t c = hpool().take_stream_if([](hipStream_t x) { return sclasses().known(x) < 0; });
        if (!c && hipStreamCreate(&c) != hipSuccess) break;
        (void)sclasses().classify(c);
        have.push_back(c);
    }
    for (hipStream_t k : have) hpool().put_stream(k);
}
std::atomic<int> g_active_runs{0};         // runs in flight in this process (pchip_run_repeats: one thread each)
std::atomic<int> g_active_dev[64];         // ... per HIP device (zero-initialised: static storage)

struct Timing { double t_gen = 0, t_loop = 0, t_final = 0, t_total = 0; long rounds = 0, updates = 0, batches = 0, compactions = 0; };

// HIP-event stopwatch per kernel class, on the engine's own stream (bench.py's roofline numbers)
enum { KT_NHATS = 0, KT_SLICE, KT_CONSUME, KT_APPLY, KT_CLEAN, KT_COV, KT_SIDE /* bases drawn ahead on the side stream */, KT_N };
struct KTimer {
    bool on = false;
    unsigned mask = 0xFFFFFFFFu;            // kernel classes that are timed
    unsigned stride = 1; unsigned seen[KT_N] = {0};   // ... every stride-th launch of a class (an event pair costs the stream ~6 us)
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> pool; size_t used = 0;
    struct Span { int k; hipEvent_t a, b; bool side; };
    std::vector<Span> open;
    double total_ms[KT_N] = {0}; long launches[KT_N] = {0};
    hipEvent_t get() { if (used == pool.size()) pool.push_back(hpool().get_event()); return pool[used++]; }
    hipEvent_t begin(int k) { if (!on || !((mask >> k) & 1u) || (seen[k]++ % stride) != 0 || open.size() >= MAX_OPEN) return nullptr; hipEvent_t e = get(); HIPCHK(hipEventRecord(e, st)); return e; }
    void end(int k, hipEvent_t a) { if (!on || !a) return; hipEvent_t e = get(); HIPCHK(hipEventRecord(e, st)); open.push_back({k, a, e, false}); }
    // the same around a launch on another stream of the run (the bases drawn ahead): collect() waits for that span's end itself
    hipEvent_t begin_on(int k, hipStream_t s2) { if (!on || !((mask >> k) & 1u) || (seen[k]++ % stride) != 0 || open.size() >= MAX_OPEN) return nullptr; hipEvent_t e = get(); HIPCHK(hipEventRecord(e, s2)); return e; }
    void end_on(int k, hipEvent_t a, hipStream_t s2) { if (!on || !a) return; hipEvent_t e = get(); HIPCHK(hipEventRecord(e, s2)); open.push_back({k, a, e, true}); }
    // Call after a stream synchronisation.  A round no longer synchronises, so the spans pile up between the moments that
    // do (an update with host work, a compaction, a growth, the end of the run); beyond MAX_OPEN open spans launches go
    // untimed (k_launches counts the timed ones) instead of creating events without bound.
    static constexpr size_t MAX_OPEN = 8192;
    void collect() {
        if (!on) return;
        for (auto &sp : open) { float ms = 0; if (sp.side) HIPCHK(hipEventSynchronize(sp.b)); HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b)); total_ms[sp.k] += ms; launches[sp.k]++; }
        open.clear(); used = 0;
    }
    void destroy() { for (auto e : pool) hpool().put_event(e); pool.clear(); }
};

// host copy of the device's counter RNG (pc_dev.h), used when the host evaluates the likelihood
static void h_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
static double h_uniform(uint32_t k0, uint32_t k1, uint32_t dom, uint32_t shi, uint32_t slo, uint32_t idx)
{
    uint32_t o[4];
    h_philox(idx >> 1, slo, shi, dom, k0, k1, o);
    const uint64_t w = (idx & 1u) ? (((uint64_t)o[2] << 32) | o[3]) : (((uint64_t)o[0] << 32) | o[1]);
    return ((double)(w >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// ---- the runs of a device in step (pchip_run_repeats with a built-in likelihood): a kernel launched on N streams at once costs
//      every one of them tens of microseconds on this device (tools/dev/ubench_queues.hip: an empty kernel 3 us alone, 36 us
//      each with sixteen streams busy; at most ~8 kernels at a time whatever the number of hardware queues), so the runs share
//      ONE stream and go round by round together: what each engine would launch in a phase of the round it writes down here, and
//      every kernel of the phase is launched ONCE for all of them (blockIdx.y = run, PcManyRec).  The same kernels' bodies on
//      the same states: the numbers of a run do not know whether it ran alone.
// ---- a run's host work that has to WAIT for the device in the middle (an update with clustering: counts down, verdicts back,
//      splits) as a fiber of the thread that drives the runs in step: where a run on its own synchronises its stream, a run in step
//      yields (Engine::sync_point); the driver goes through all the runs that have something to wait for, launches what they
//      wrote down ONCE for all of them, waits ONCE, and resumes them.  The waits of sixteen runs' updates cost what one run's do,
//      and the kernels between two waits are launched together.  (makecontext / swapcontext: no threads, no locks; an
//      exception inside a fiber is caught at its foot and rethrown by the driver.)
struct FiberCancelled {};      // thrown inside a suspended fiber that is resumed only to unwind (another run's update failed)
struct Fiber {
    ucontext_t ctx, ret;
    void *stack = nullptr; size_t stack_sz = 0;      // usable part; one PROT_NONE page below it (stacks grow down): an overflow faults
    void *map = nullptr; size_t map_sz = 0;          // instead of running into the heap
    std::function<void()> fn;
    bool started = false, done = false, cancel = false;
    std::exception_ptr err;
    // round_finish with clustering goes deep (update, kNN passes, add_cluster, the resume file's writer, HIP runtime calls): 8 MB of
    // address space, committed as touched
    void make_stack()
    {
        if (stack) return;
        const size_t page = (size_t)sysconf(_SC_PAGESIZE), want = (size_t)8 << 20;
        void *m = mmap(nullptr, want + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        (void)mprotect(m, page, PROT_NONE);
        map = m; map_sz = want + page; stack = (char *)m + page; stack_sz = want;
    }
    void free_stack() { if (map) munmap(map, map_sz); map = stack = nullptr; map_sz = stack_sz = 0; }
    static void foot(unsigned lo, unsigned hi)
    {
        Fiber *f = (Fiber *)(((uintptr_t)hi << 32) | (uintptr_t)lo);
        try { f->fn(); } catch (...) { f->err = std::current_exception(); }
        f->done = true;
        swapcontext(&f->ctx, &f->ret);
    }
    void resume()
    {
        if (!started) {
            getcontext(&ctx);
            ctx.uc_stack.ss_sp = stack; ctx.uc_stack.ss_size = stack_sz; ctx.uc_link = nullptr;
            const uintptr_t a = (uintptr_t)this;
            makecontext(&ctx, (void (*)())foot, 2, (unsigned)(a & 0xFFFFFFFFu), (unsigned)(a >> 32));
            started = true;
        }
        swapcontext(&ret, &ctx);
    }
    void yield() { swapcontext(&ctx, &ret); }
};

enum { CK_COMPACT = 0, CK_RESET, CK_CLUS1, CK_CLUSG, CK_BASES, CK_NHATS_G, CK_SLICE, CK_SLICE_G, CK_BASES_NEXT, CK_SORT, CK_NN, CK_CONSUME, CK_CONSUME_CL, CK_APPLY, CK_UPDATE, CK_COV, CK_FINAL, CK_N };      // (in the order they are launched)
// (_G: any device likelihood, the wavefront-per-chain kernels of a run on its own with the run in the grid; NN / CONSUME_CL: runs with several clusters)
struct Cohort {
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr;          // the bases of the NEXT nursery, next to this one's sampling and contraction
    hipEvent_t ev_up = nullptr, ev_next = nullptr; bool next_pending = false;
    // bases drawn TWO nurseries ahead: a sampling launch waits for the launch that drew ITS bases (numbered), not for the second stream's
    // latest -- with one event for the latest, sixteen runs' k_slice_t waited 140 us per round without an update for the 250 us of
    // deviates + bases launched a round before
    hipEvent_t ev_seq[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned long long seq_launched = 0, seq_waited = 0; bool seq_open = false;
    int seq_for_next() const { return (int)(seq_launched + 1); }      // (the number the bases written down now will be launched under)
    struct Rec { int kind; PcState S; void *p[10]; long long a[4]; int ia[6]; };      // a: what the runs of one launch must share; ia: each run's own (PcManyRec::ia)
    std::vector<Rec> pend;
    static constexpr int RING = 4;
    PcManyRec *h_stage[RING] = {}, *d_recs[RING] = {};
    // a slot's records are read by the kernels launched from it: its events are recorded behind the LAST of them, on both streams
    hipEvent_t ev[RING] = {}, ev2[RING] = {}; bool ev_used[RING] = {}, ev2_used[RING] = {};
    size_t cap = 0; int ring = 0;
    void slot_wait(int k)
    {
        if (ev_used[k]) { HIPCHK(hipEventSynchronize(ev[k])); ev_used[k] = false; }
        if (ev2_used[k]) { HIPCHK(hipEventSynchronize(ev2[k])); ev2_used[k] = false; }
    }
    // the bases of this nursery were drawn on the second stream: whatever reads them on the main stream comes behind them
    void wait_next() { if (next_pending) { HIPCHK(hipStreamWaitEvent(st, ev_next, 0)); next_pending = false; } }
    long n_fused = 0, n_single = 0;
    // the runs' copies to the host (dead rows, results) share two streams of the cohort, on hardware queues other than the two its
    // kernels use: a copy stream per run came from the pool, on whatever queue -- and where copies are shader blits (the HIP runtime
    // PyTorch ships: 48 us each) a round's kernels queued behind them
    hipStream_t stc[2] = {nullptr, nullptr}; int n_stc = 0;
    // copies to the host that belong behind what has been written down: made at the end of flush(), in the order they were asked for
    std::vector<std::function<void()>> post;
    // ... and copies to the device that what is written down reads: made at the start of flush()
    std::vector<std::function<void()>> pre;
    void rec(int kind, const PcState &S, std::initializer_list<void *> p, std::initializer_list<long long> a, std::initializer_list<int> ia)
    {
        pend.emplace_back();
        Rec &r = pend.back();
        r.kind = kind; r.S = S;
        int i = 0; for (void *x : p) r.p[i++] = x; for (; i < 10; ++i) r.p[i] = nullptr;
        i = 0; for (long long x : a) r.a[i++] = x; for (; i < 4; ++i) r.a[i] = 0;
        i = 0; for (int x : ia) r.ia[i++] = x; for (; i < 6; ++i) r.ia[i] = 0;
    }
    static void single(const Rec &r, hipStream_t st)
    {
        switch (r.kind) {
        case CK_COMPACT: pc_launch_clean(&r.S, r.ia[1], (unsigned char *)r.p[0], (int *)r.p[1], (int *)r.p[2], (double *)r.p[3], (double *)r.p[4], (unsigned *)r.p[5], (unsigned long long *)r.p[6], nullptr, st); break;
        case CK_BASES: case CK_BASES_NEXT: (void)pc_launch_nhats_part(&r.S, (unsigned)r.ia[0], (int)r.a[0], 1, st, 1); break;
        case CK_SLICE: (void)pc_launch_slice_t(&r.S, (unsigned)r.ia[0], (int)r.a[0], st); break;
        case CK_NHATS_G: (void)pc_launch_nhats(&r.S, (unsigned)r.ia[0], (int)r.a[0], st); break;
        case CK_SLICE_G: if (r.a[1]) (void)pc_launch_slice_fused(&r.S, (unsigned)r.ia[0], (int)r.a[0], st); else (void)pc_launch_slice(&r.S, (unsigned)r.ia[0], (int)r.a[0], st); break;
        case CK_NN: pc_launch_nn_lists(&r.S, r.ia[1], 1, st); break;      // (the run's CK_SORT of this round has been launched: kinds go in order)
        case CK_RESET: pc_launch_reset_thresholds(&r.S, st); break;
        case CK_CLUSG: (void)pc_launch_knn_cluster_sub((const int *)r.p[0], r.ia[1], r.ia[2], (const double *)r.p[1], (const int *)r.p[2], (int *)r.p[3], (int *)r.p[4], (int *)r.p[5], st); break;
        case CK_CLUS1: (void)pc_launch_knn_cluster_batch_dev(&r.S, (const int *)r.p[0], r.ia[1], r.ia[2], (double *)r.p[1], (int *)r.p[2], (int *)r.p[3], (int *)r.p[4], st); break;
        case CK_CONSUME_CL: (void)pc_launch_consume_cl(&r.S, r.a[0] ? 65 : 2, st); break;
        case CK_SORT: (void)pc_launch_sort_live(&r.S, st); break;
        case CK_CONSUME: (void)pc_launch_consume_par(&r.S, st); break;
        case CK_FINAL: (void)pc_launch_final_par(&r.S, st); break;
        case CK_APPLY: pc_launch_apply(&r.S, (unsigned)r.ia[0], (int)r.a[0], st); break;
        case CK_UPDATE: pc_launch_update_fused(&r.S, r.ia[1], (unsigned char *)r.p[0], (int *)r.p[1], (int *)r.p[2], (double *)r.p[3], (double *)r.p[4],
                                               (unsigned *)r.p[5], (unsigned long long *)r.p[6], (double *)r.p[7], (double *)r.p[8], (int)r.a[1], st); break;
        }
    }
    // ... the copies among them as (destination, source, bytes): one kernel for all of them (k_copy_batch), not a hipMemcpyAsync each
    std::vector<std::array<uintptr_t, 3>> post_copies, pre_copies;
    void run_post()
    {
        if (!post_copies.empty()) { std::vector<std::array<uintptr_t, 3>> c; c.swap(post_copies); pc_copy_many(c, st); }
        if (post.empty()) return;
        std::vector<std::function<void()>> p; p.swap(post); for (auto &f : p) f();
    }
    void run_pre()
    {
        if (!pre_copies.empty()) { std::vector<std::array<uintptr_t, 3>> c; c.swap(pre_copies); pc_copy_many(c, st); }
        if (pre.empty()) return;
        std::vector<std::function<void()>> p; p.swap(pre); for (auto &f : p) f();
    }
    void flush()
    {
        run_pre();
        if (pend.empty()) { run_post(); return; }
        const size_t n = pend.size();
        if (n > cap) {
            for (int k = 0; k < RING; ++k) {
                slot_wait(k);      // (kernels of earlier flushes may still be reading the old records, on either stream)
                // (from the block caches and the event pool: asking the driver -- and giving back to it at the end -- was 2 ms per call)
                if (h_stage[k]) hfree(h_stage[k]);
                if (d_recs[k]) dfree(d_recs[k]);
                h_stage[k] = halloc<PcManyRec>(2 * n);
                d_recs[k] = dalloc<PcManyRec>(2 * n);
                if (!ev[k]) ev[k] = hpool().get_sync_event();
                if (!ev2[k] && st2) ev2[k] = hpool().get_sync_event();
            }
            cap = 2 * n;
        }
        const int slot = ring++ % RING;
        slot_wait(slot);
        PcManyRec *hs = h_stage[slot], *dr = d_recs[slot];
        // records in launch order: by kind, and inside a kind by the arguments all runs of a launch must share
        std::vector<const Rec *> ord; ord.reserve(n);
        for (const Rec &r : pend) ord.push_back(&r);
        auto shape_less = [](const Rec *x, const Rec *y) {
            if (x->kind != y->kind) return x->kind < y->kind;
            const int c = std::memcmp(x->a, y->a, sizeof(x->a));
            if (c != 0) return c < 0;
            if (x->S.Ncap != y->S.Ncap) return x->S.Ncap < y->S.Ncap;
            if (x->S.B != y->S.B) return x->S.B < y->S.B;
            if (x->S.pool != y->S.pool) return x->S.pool < y->S.pool;
            return (x->S.prior.lo == nullptr) < (y->S.prior.lo == nullptr);
        };
        std::stable_sort(ord.begin(), ord.end(), shape_less);
        for (size_t i = 0; i < n; ++i) { hs[i].S = ord[i]->S; std::memcpy(hs[i].p, ord[i]->p, sizeof(ord[i]->p)); std::memcpy(hs[i].ia, ord[i]->ia, sizeof(ord[i]->ia)); }
        HIPCHK(hipMemcpyAsync(dr, hs, sizeof(PcManyRec) * n, hipMemcpyHostToDevice, st));
        bool up_marked = false, used_st2 = false;
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            while (j < n && !shape_less(ord[i], ord[j]) && !shape_less(ord[j], ord[i])) ++j;
            const Rec &f = *ord[i];
            const PcManyRec *d = dr + i;
            const int cnt = (int)(j - i), k = f.kind;
            hipStream_t q = st;
            if (k == CK_BASES_NEXT && st2) {         // on the second stream, behind the upload of the records
                if (!up_marked) { HIPCHK(hipEventRecord(ev_up, st)); up_marked = true; }
                HIPCHK(hipStreamWaitEvent(st2, ev_up, 0));
                q = st2; used_st2 = true;
            }
            if (k == CK_SLICE || k == CK_SLICE_G) {             // (its bases were drawn over there)
                int need = 0; bool numbered = st2 != nullptr;
                for (size_t x = i; x < j; ++x) { numbered = numbered && ord[x]->ia[3] > 0; need = std::max(need, ord[x]->ia[3]); }
                if (numbered && (unsigned long long)need <= seq_launched) {
                    if ((unsigned long long)need > seq_waited) { HIPCHK(hipStreamWaitEvent(st, ev_seq[need & 3], 0)); seq_waited = (unsigned long long)need; }
                } else wait_next();
            }
            int rc = 1;
            switch (k) {
            case CK_COMPACT: { int nbm = 0; for (size_t x = i; x < j; ++x) nbm = std::max(nbm, ord[x]->ia[2]); rc = pc_launch_clean_many(d, cnt, nbm, q); } break;
            case CK_BASES: case CK_BASES_NEXT: rc = pc_launch_bases_t_many(&f.S, d, cnt, 0u, (int)f.a[0], q); break;
            case CK_SLICE: rc = pc_launch_slice_t_many(&f.S, d, cnt, 0u, (int)f.a[0], q); break;
            case CK_NHATS_G: rc = pc_launch_nhats_many(&f.S, d, cnt, (int)f.a[0], q); break;
            case CK_SLICE_G: rc = pc_launch_slice_many(&f.S, d, cnt, (int)f.a[0], (int)f.a[1], q); break;
            case CK_NN: { int nl = 0; for (size_t x = i; x < j; ++x) nl = std::max(nl, ord[x]->ia[1]); rc = pc_launch_nn_lists_many(&f.S, d, cnt, nl, 1, q); } break;
            case CK_CONSUME_CL: rc = pc_launch_consume_cl_many(&f.S, d, cnt, (int)f.a[0], q); break;
            case CK_RESET: rc = pc_launch_reset_thresholds_many(&f.S, d, cnt, q); break;
            case CK_CLUSG: { int nbm = 0, nmx = 0; for (size_t x = i; x < j; ++x) { nbm = std::max(nbm, ord[x]->ia[1]); nmx = std::max(nmx, ord[x]->ia[2]); } rc = pc_launch_knn_cluster_sub_many(d, cnt, nbm, nmx, q); } break;
            case CK_CLUS1: { int ndm = 0, nmx = 0; for (size_t x = i; x < j; ++x) { ndm = std::max(ndm, ord[x]->ia[1]); nmx = std::max(nmx, ord[x]->ia[2]); } rc = pc_launch_knn_cluster_batch_many(&f.S, d, cnt, ndm, nmx, q); } break;
            case CK_SORT: rc = pc_launch_sort_live_many(&f.S, d, cnt, q); break;
            case CK_CONSUME: rc = pc_launch_consume_par_many(&f.S, d, cnt, q); break;
            case CK_FINAL: rc = pc_launch_final_par_many(d, cnt, q); break;
            case CK_APPLY: rc = pc_launch_apply_many(&f.S, d, cnt, 0u, (int)f.a[0], q); break;
            case CK_UPDATE: { int nbm = 0; for (size_t x = i; x < j; ++x) nbm = std::max(nbm, ord[x]->ia[2]); rc = pc_launch_update_fused_many(&f.S, d, cnt, nbm, (int)f.a[0], (int)f.a[1], q); } break;
            }
            if (rc == 0) n_fused += cnt;
            else for (size_t x = i; x < j; ++x) { single(*ord[x], q); n_single++; }
            if (k == CK_BASES_NEXT && st2) {
                HIPCHK(hipEventRecord(ev_next, st2)); next_pending = true;
                if (!seq_open) { seq_launched++; seq_open = true; }
                if (!ev_seq[seq_launched & 3]) ev_seq[seq_launched & 3] = hpool().get_sync_event();
                HIPCHK(hipEventRecord(ev_seq[seq_launched & 3], st2));
            }
            i = j;
        }
        HIPCHK(hipEventRecord(ev[slot], st)); ev_used[slot] = true;
        if (used_st2 && ev2[slot]) { HIPCHK(hipEventRecord(ev2[slot], st2)); ev2_used[slot] = true; }
        pend.clear();
        seq_open = false;
        run_post();
    }
    void destroy()
    {
        for (int k = 0; k < 4; ++k) if (ev_seq[k]) { hpool().put_sync_event(ev_seq[k]); ev_seq[k] = nullptr; }
        for (int k = 0; k < RING; ++k) {
            if (ev_used[k]) (void)hipEventSynchronize(ev[k]);
            if (ev2_used[k]) (void)hipEventSynchronize(ev2[k]);
            if (ev[k]) hpool().put_sync_event(ev[k]);
            if (ev2[k]) hpool().put_sync_event(ev2[k]);
            ev2[k] = nullptr; ev2_used[k] = false;
            if (h_stage[k]) hfree(h_stage[k]);
            if (d_recs[k]) dfree(d_recs[k]);
            ev[k] = nullptr; h_stage[k] = nullptr; d_recs[k] = nullptr; ev_used[k] = false;
        }
        cap = 0;
    }
};

static std::atomic<long long> g_dbg_compact_ns{0}, g_dbg_nursery_ns{0}, g_dbg_capacity_ns{0}, g_dbg_endb_ns{0}, g_dbg_destroy_ns{0}, g_dbg_evwait_ns{0}, g_dbg_d1{0}, g_dbg_d2{0};      // (PC_DEBUG=5: where round_enqueue's time goes)
struct Engine {
    Cohort *co = nullptr;               // not null: this run goes in step with others of its device, on their common stream
    Fiber *fib = nullptr;               // not null: this call runs as a fiber of the thread that drives the runs in step (waits are shared)
    pchip_settings cfg{};
    std::atomic<int> stop{0};                   // polychord_hip_request_stop reached this run
    polychord_batch_fn batch_fn = nullptr; void *batch_user = nullptr;     // the batch callback registered when the run was set up
    // host-callback mode (device proposes, host evaluates): pc_callback.hip
    polychord_loglike_fn cb_like = nullptr; polychord_prior_fn cb_prior = nullptr;
    bool callback_mode = false;
    void *d_cs = nullptr; double *d_x0s = nullptr, *d_prop = nullptr;
    int *d_decks = nullptr;
    // pinned host side of the propose / evaluate exchange: the kernel stores proposals and need flags here directly; the
    // answers (logL | theta | phi of every chain, one block) go back in one copy
    double *hp_prop = nullptr, *hp_ans = nullptr, *d_ans = nullptr; int *hp_need = nullptr;
    long long cb_evals = 0; long cb_ticks = 0;
    std::vector<int> cb_idx; std::vector<double> cb_c, cb_t, cb_p, cb_l;   // batch callback scratch
    // host mirror of the dead points for the dumper hook (nested_sampling.F90:546-590)
    polychord_dumper_fn dumper = nullptr;
    pchip_update_fn on_update = nullptr; void *hook_user = nullptr;
    long ndiscarded = 0;                        // prior samples rejected while generating the live points
    bool resume_static = true; unsigned resume_batch0 = 0;
    bool nph_stale = false;                     // h_ctl->nphantom is the count before the last clean (an upper bound)
    std::vector<double> hm_dead, hm_logw; int hm_ndead = 0;
    std::vector<unsigned> hm_cuid;              // cluster uid every dead point died in
    // boost_posterior: phantoms removed by a clean that were kept as posterior samples (run_time_info.f90:857-870)
    std::vector<double> pp_rows, pp_logpost; std::vector<unsigned> pp_cuid; std::vector<unsigned long long> pp_uid; int nd_last_update = 0;
    // cluster genealogy: child uid, parent uid, log of the evidence fraction the child received (add_cluster)
    std::vector<unsigned> split_child, split_parent; std::vector<double> split_logfrac;
    std::vector<double> h_lo, h_hi;
    PcState S{};
    hipStream_t st = nullptr;
    hipStream_t st_copy = nullptr;            // dead rows leave for the host while the run goes on
    bool st_copy_shared = false;              // ... one of the cohort's two (not this run's to give back)
    hipStream_t st_side = nullptr;            // the orthonormal bases of the next nursery, while this one is consumed
    hipEvent_t ev_main = nullptr;
    // ring of bases drawn ahead on the side stream: nursery b's live in raw_buf[b % raw_depth]
    static constexpr int RAW_RING = 4;
    struct RawSlot { hipEvent_t ready = nullptr, consumed = nullptr; unsigned batch = 0; int B = 0; bool valid = false, waited = false, used = false; int co_seq = 0; };
    RawSlot ring[RAW_RING];
    int raw_depth = 2;
    double *h_dead = nullptr; size_t h_dead_cap = 0, h_dead_copied = 0;
    PcCtl *h_ctl = nullptr;       // pinned mirror
    PcCtl *h_note = nullptr;      // pinned, device-visible: the contraction kernels publish the control block here (pc_publish_ctl)
    unsigned note_seq = 0;
    hipEvent_t ev_apply = nullptr;
    double *raw_buf[RAW_RING] = {nullptr, nullptr, nullptr, nullptr};
    // alternate phantom buffers + scratch for the update step
    double *ph2 = nullptr, *phL2 = nullptr; unsigned *phC2 = nullptr; unsigned long long *phU2 = nullptr;
    unsigned char *keep = nullptr; int *blk = nullptr, *d_total = nullptr;
    double *psum = nullptr, *mean = nullptr, *pcov = nullptr; int *pcnt = nullptr, *count = nullptr;
    size_t cov_chunks_cap = 0;
    double *upd_part = nullptr, *upd_shift = nullptr; size_t upd_part_cap = 0;   // fused update (pc_update.hip)
    double *d_lo = nullptr, *d_hi = nullptr, *d_invcovT = nullptr, *d_mean = nullptr;
    double *d_dynL = nullptr; int *d_dynN = nullptr; double *d_logn = nullptr;
    // clustering scratch (allocated on first use)
    double *c_Sm = nullptr; int *c_pts = nullptr, *c_gidx = nullptr, *c_knn = nullptr, *c_lab = nullptr, *c_out = nullptr, *c_cnt = nullptr;
    unsigned *c_olduid = nullptr; int c_cap = 0;
    long nsplits = 0; int ncluster_peak = 1;
    long path[PCHIP_PATH_COUNT] = {};          // launches per kernel variant (pchip_result.path): counted where the choice is made
    Timing tm;
    KTimer kt;
    int B = 0, dev = 0;
    bool fast_ok = false;
    bool cb_auto_batch = false; int B_small = 1; double cb_eval_seconds = -1.0;   // host callbacks: chains per nursery chosen from the measured cost of a call
    long long nlike_g[PC_MAX_GRADE] = {0};      // RTI%nlike per grade (grade 1 includes the prior samples)
    std::vector<int> h_nlike_g;                 // [B][PC_MAX_GRADE] of the batch in the nursery

    void alloc_phantom_side(int Pcap)
    {
        ph2 = dalloc<double>((size_t)Pcap * S.nT); phL2 = dalloc<double>(Pcap); phC2 = dalloc<unsigned>(Pcap);
        phU2 = dalloc<unsigned long long>(Pcap); keep = dalloc<unsigned char>(Pcap); blk = dalloc<int>((Pcap + 255) / 256 + 1);
    }

    // a small table of the set-up on its way to the device: through a pinned block of the cache, in stream order (a blocking copy
    // from pageable memory was 0.1 ms as a rule and 4 ... 11 ms now and then -- one run in eight or so -- while the driver pinned
    // the vector's pages); the blocks go back at the teardown
    std::vector<void *> setup_staged;
    void upload(void *dst, const void *src, size_t bytes)
    {
        void *h = halloc<char>(bytes);
        setup_staged.push_back(h);
        std::memcpy(h, src, bytes);
        HIPCHK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, st));
    }
    void setup(const pchip_settings &c, const pchip_like &like, const pchip_prior &prior)
    {
        cfg = c;
        { std::lock_guard<std::mutex> g(g_cb_mutex); batch_fn = g_batch_fn; batch_user = g_batch_user; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            engine_fail(PC_RC_DEVICE, "no HIP device available -- this engine has no CPU path");
        }
        dev = c.device >= 0 ? c.device % ndev : 0;
        HIPCHK(hipSetDevice(dev));
        st = co ? co->st : hpool().get_stream();
        st_copy = (co && co->stc[0]) ? co->stc[(co->n_stc++) & 1] : (co ? hpool().get_stream() : stream_beside({st}));      // (a run on its own: its copies on another hardware queue than its kernels)
        st_copy_shared = co && co->stc[0];
        kt.on = c.profile != 0 && !co; kt.st = st;      // (in step with other runs the launches are made elsewhere: nothing of its own to time)
        kt.mask = (c.profile == 1) ? 0xFFFFFFFFu : (((unsigned)c.profile >> 1) & 0x7Fu);   // 1: every class; else bit k+1 = class k
        kt.stride = std::max(1u, ((unsigned)c.profile >> 8) & 0xFFu);                       // bits 8..15: time every n-th launch of a class
        const int D = c.nDims, nDer = c.nDerived;
        S.D = D; S.nDer = nDer; S.nT = 2 * D + nDer + 2; S.nr = c.num_repeats; S.N = c.nlive;
        // grades (chordal_sampling.f90:119-130): bases per grade, first direction and first deviate of each
        S.ngrade = 1; S.g_off[0] = 0; S.g_nr[0] = c.num_repeats;
        for (int g = 1; g < PC_MAX_GRADE; ++g) { S.g_off[g] = 0; S.g_nr[g] = 0; }
        if (c.nGrade > 1 && c.grade_dims && c.grade_repeats) {
            if (c.nGrade > PC_MAX_GRADE) engine_fail(PC_RC_SETTINGS, "at most %d parameter grades", PC_MAX_GRADE);
            S.ngrade = c.nGrade;
            int off = 0, tot = 0;
            for (int g = 0; g < c.nGrade; ++g) {
                if (c.grade_dims[g] < 1 || c.grade_repeats[g] < 1) engine_fail(PC_RC_SETTINGS, "every grade needs at least one parameter and one repeat");
                S.g_off[g] = off; off += c.grade_dims[g]; S.g_nr[g] = c.grade_repeats[g]; tot += c.grade_repeats[g];
            }
            if (off != D) engine_fail(PC_RC_SETTINGS, "grade_dims must sum to nDims");
            S.nr = tot; cfg.num_repeats = tot;
        }
        S.nb_total = 0; S.n_dev = 0;
        for (int g = 0, col = 0; g < PC_MAX_GRADE; ++g) {
            const int Dg = D - S.g_off[g];
            S.g_nb[g] = g < S.ngrade ? (S.g_nr[g] + Dg - 1) / Dg : 0;
            S.g_col0[g] = col; S.g_e0[g] = (int)S.n_dev;
            col += S.g_nr[g]; S.nb_total += S.g_nb[g]; S.n_dev += (unsigned)S.g_nb[g] * Dg * Dg;
        }
        S.ch_nlike_g = nullptr;
        std::memset(nlike_g, 0, sizeof(nlike_g));
        S.p0 = D; S.d0 = 2 * D; S.b0 = 2 * D + nDer; S.l0 = S.b0 + 1;
        int nmax = c.nlive;
        for (int i = 0; i < c.n_nlives; ++i) nmax = std::max(nmax, c.nlives[i]);
        const int nprior = c.nprior <= 0 ? c.nlive : c.nprior;
        S.Ncap = std::max(nmax, nprior);
        B = c.batch > 0 ? c.batch : std::max(1, std::min(1024, c.nlive / 2));
        // (the contraction kernels stamp the nursery position into 16 bits of the host mirror, pc_state.h pc_note: never truncate silently)
        if (B > 65535) engine_fail(PC_RC_LIMIT, "batch = %d chains per nursery: at most 65535", B);
        if (c.sequential_rng) B = 1;
        // Host callbacks: every chain of a nursery is seeded from one snapshot, so about B / (2 nlive) of the evaluations
        // are spent on spawns that fail -- cheap for a compiled likelihood (then the round trips per nursery dominate and
        // many chains per nursery pay: nlive / 2), dear for an expensive one (nlive / 4, at most 64).  Which one this is
        // is measured while the live points are generated; buffers are sized for the larger choice.
        cb_auto_batch = c.batch <= 0 && !c.sequential_rng && (like.kind == PC_LIKE_CALLBACK || prior.kind != 1);
        if (cb_auto_batch) B_small = std::max(1, std::min(64, c.nlive / 4));
        S.B = B;
        S.maxc = c.do_clustering ? std::max(2, g_cap_clusters.load()) : 4;
        S.maxc_dead = 4096;
        S.Pcap = (int)std::min<long long>(2000000000LL / S.nT, 4LL * S.nr * S.Ncap + 4LL * B * S.nr + 1024);
        if (g_cap_phantoms.load() > 0) S.Pcap = std::max(g_cap_phantoms.load(), B * S.nr + 16);
        S.Dcap = 64 * S.Ncap + 4 * B + 1024;
        S.k0 = (uint32_t)c.seed; S.k1 = 0x504F4C59u;
        S.logzero = c.logzero; S.use_prec = c.precision_criterion > 0;
        S.log_prec = S.use_prec ? std::log(c.precision_criterion) : 0.0;
        S.log_cf = std::log(c.compression_factor);
        S.max_ndead = c.max_ndead; S.nfail = c.nfail <= 0 ? c.nlive : c.nfail;
        S.seed_override = 0; S.ablate = c.ablate; S.seq_mode = c.sequential_rng ? 1 : 0;
        S.epoch_discard = c.epoch_discard ? 1 : 0;
        if (S.seq_mode) { cfg.batch = 1; cfg.force_general = 1; }
        // dynamic nlive tables
        S.n_nlives = c.n_nlives;
        if (c.n_nlives > 0) {
            d_dynL = dalloc<double>(c.n_nlives); d_dynN = dalloc<int>(c.n_nlives);
            upload(d_dynL, c.loglikes, sizeof(double) * c.n_nlives);
            upload(d_dynN, c.nlives, sizeof(int) * c.n_nlives);
        }
        S.dyn_loglikes = d_dynL; S.dyn_nlives = d_dynN;
        // likelihood / prior
        S.like.kind = like.kind; S.like.mu = like.mu; S.like.sigma = like.sigma; S.like.logdetcov = like.logdetcov;
        S.like.norm = -(double)D * (std::log(like.sigma > 0 ? like.sigma : 1.0) + PC_LOG_TWO_PI / 2.0);
        S.like.inv_sigma = like.sigma > 0 ? 1.0 / like.sigma : 1.0;
        S.like.log_vn = 0.5 * D * std::log(3.14159265358979323846) - std::lgamma(1.0 + D / 2.0);
        S.like.invcov = nullptr; S.like.mean = nullptr;
        if (like.kind == PC_LIKE_CORR_GAUSSIAN) {
            std::vector<double> T((size_t)D * D);
            for (int a = 0; a < D; ++a) for (int b = 0; b < D; ++b) T[(size_t)b * D + a] = like.invcov[(size_t)a * D + b];
            d_invcovT = dalloc<double>((size_t)D * D); d_mean = dalloc<double>(D);
            upload(d_invcovT, T.data(), sizeof(double) * D * D);
            upload(d_mean, like.mean, sizeof(double) * D);
            S.like.invcov = d_invcovT; S.like.mean = d_mean;
        }
        callback_mode = (like.kind == PC_LIKE_CALLBACK) || (prior.kind != 1);
        if (callback_mode) {
            cb_like = like.fn; cb_prior = prior.fn;
            if (!cb_like) engine_fail(PC_RC_SETTINGS, "callback mode needs a loglikelihood function pointer");
        }
        S.prior.kind = prior.kind; S.prior.lo = nullptr; S.prior.hi = nullptr;
        if (prior.kind == 1 && prior.lo && prior.hi) {
            d_lo = dalloc<double>(D); d_hi = dalloc<double>(D);
            upload(d_lo, prior.lo, sizeof(double) * D);
            upload(d_hi, prior.hi, sizeof(double) * D);
            S.prior.lo = d_lo; S.prior.hi = d_hi;
        }
        // state arrays
        const int Ncap = S.Ncap, maxc = S.maxc, nT = S.nT, nr = S.nr;
        S.live = dalloc<double>((size_t)Ncap * nT); S.live_logL = dalloc<double>(Ncap);
        S.live_cluster = dalloc<int>(Ncap); S.live_pos = dalloc<int>(Ncap); S.live_entry = dalloc<double>(Ncap);
        S.cl_list = dalloc<int>((size_t)maxc * Ncap); S.cl_n = dalloc<int>(maxc);
        S.logZp = dalloc<double>(maxc); S.logXp = dalloc<double>(maxc); S.logZXp = dalloc<double>(maxc);
        S.logZp2 = dalloc<double>(maxc); S.logZpXp = dalloc<double>(maxc); S.logLp = dalloc<double>(maxc);
        S.XpXq = dalloc<double>(2 * (size_t)maxc * maxc); S.imin_slot = dalloc<int>(maxc);      // (the second half: k_consume_clp's copy of the matrix in linear space, made anew by every pass)
        S.lse_ref = dalloc<double>(maxc); S.lse_sum = dalloc<double>(maxc); S.death_thr = dalloc<double>(maxc);
        S.cl_uid = dalloc<unsigned>(maxc);
        S.chol = dalloc<double>((size_t)maxc * D * D); S.cov = dalloc<double>((size_t)maxc * D * D);
        S.logZp_dead = dalloc<double>(S.maxc_dead); S.logZp2_dead = dalloc<double>(S.maxc_dead); S.cl_uid_dead = dalloc<unsigned>(S.maxc_dead);
        S.phantom = dalloc<double>((size_t)S.Pcap * nT); S.ph_logL = dalloc<double>(S.Pcap);
        S.ph_cuid = dalloc<unsigned>(S.Pcap); S.ph_uid = dalloc<unsigned long long>(S.Pcap);
        alloc_phantom_side(S.Pcap);
        S.dead = dalloc<double>((size_t)S.Dcap * nT); S.dead_logw = dalloc<double>(S.Dcap);
        S.dead_postX = dalloc<double>(S.Dcap); S.dead_postZ = dalloc<double>(S.Dcap); S.dead_cuid = dalloc<unsigned>(S.Dcap);
        S.dead_entry = dalloc<double>(S.Dcap);
        S.babies = dalloc<double>((size_t)B * nr * nT); S.baby_logL = dalloc<double>((size_t)B * nr); S.baby_logL_T = dalloc<double>((size_t)B * nr);
        S.ch_cluster = dalloc<int>(B); S.ch_epoch = dalloc<int>(B); S.ch_nlike = dalloc<int>(B); S.ch_seed_slot = dalloc<int>(B);
        S.ch_contour = dalloc<double>(B);
        if (S.ngrade > 1) { S.ch_nlike_g = dalloc<int>((size_t)B * PC_MAX_GRADE); h_nlike_g.assign((size_t)B * PC_MAX_GRADE, 0); }
        {   // log k for the evidence update of the serial kernel
            std::vector<double> ln((size_t)Ncap + 4);
            ln[0] = -PC_HUGE;
            for (int k = 1; k < Ncap + 4; ++k) ln[k] = std::log((double)k);
            d_logn = dalloc<double>(ln.size());
            upload(d_logn, ln.data(), sizeof(double) * ln.size());
            S.logn = d_logn;
        }
        S.nn_list = nullptr; S.nn_slot_owner = nullptr; S.nn_chain_slot = nullptr; S.nn_pts = nullptr; S.nn_code = nullptr; S.nn_valid = 0;
        if (c.do_clustering) {      // candidate lists of the nearest-cluster search (k_nn_lists)
            S.nn_list = dalloc<int>((size_t)B * nr * PC_NN_K); S.nn_slot_owner = dalloc<int>(Ncap); S.nn_chain_slot = dalloc<int>(B);
            S.nn_pts = dalloc<double>((size_t)(Ncap + B) * D); S.nn_code = dalloc<int>((size_t)2 * Ncap + B + 64);      // (codes of the candidates, then the ranks of the live points and their number: k_sort_live)
        }
        S.nhat = dalloc<double>((size_t)B * nr * D); S.nhat_w = dalloc<double>((size_t)B * nr);
        static const bool ms_off = std::getenv("PC_MS_PRE_OFF") != nullptr;
        const bool ms_pre = S.like.kind == PC_LIKE_CORR_GAUSSIAN && D > 64 && D <= 128 && S.ngrade <= 1 && !S.seq_mode && !(S.ablate & 1) && !ms_off;
        S.nhat_Ms = ms_pre ? dalloc<double>((size_t)B * nr * D) : nullptr;
        S.ch_My = ms_pre ? dalloc<double>((size_t)B * D) : nullptr;
        static const bool split_off = std::getenv("PC_NHATS_SPLIT_OFF") != nullptr;
        // 64 < nDims <= 128, one grade: k_nhats_q<32, 1> leaves a basis register-major, 32 x 512 doubles
        const bool split_q = D > 64 && D <= 128 && S.ngrade <= 1 && !S.seq_mode && !split_off && !callback_mode;
        // 24 < nDims <= 64, one grade (round 5): k_nhats_q<8 / 16, 1> leaves a basis thread by thread, 16 HV^2 doubles
        const bool split_m = D > 24 && D <= 64 && S.ngrade <= 1 && !S.seq_mode && !split_off && !callback_mode;
        const size_t raw_n = split_q ? (size_t)B * S.nb_total * 32 * 512 : split_m ? (size_t)B * S.nb_total * (D <= 32 ? 1024 : 4096) : (size_t)B * S.nb_total * D * D + (size_t)B * 33 + 8;      // (+ the chains' deck records, 66 ints each: pc_deck_record, pc_sample.hip)
        S.nhat_raw = ((D <= 24 && !S.seq_mode && !split_off) || split_q || split_m) ? dalloc<double>(raw_n)
                   : (D > 128 ? dalloc<double>((size_t)B * S.nb_total * D * 256) : nullptr);   // k_nhats_big keeps its bases there
        // split launch: two buffers, so that the bases of nursery b + 1 can be drawn at any time while nursery b's are read
        // (nDims > 64: the bases cost more than the rest of a nursery's round -- they are drawn up to three nurseries ahead,
        //  through the updates as well)
        // (nDims <= 24: three -- the bases of nursery b + 2 are drawn under nursery b's contraction.  With two, nursery b + 1's were drawn
        //  there and k_slice(b + 1) waited for them across streams: kernel 39 us + ~10 us for the event to cross, a path as long as
        //  contraction + row copies + the host's look at the stamp, which is why enqueueing k_slice ahead changed nothing)
        static const int depth_env = std::getenv("PC_RAW_DEPTH") ? std::max(2, std::min(RAW_RING, std::atoi(std::getenv("PC_RAW_DEPTH")))) : 3;
        raw_depth = split_q ? RAW_RING : depth_env;
        raw_buf[0] = S.nhat_raw;
        for (int r = 1; r < raw_depth; ++r) raw_buf[r] = ((D <= 24 || split_q || split_m) && S.nhat_raw) ? dalloc<double>(raw_n) : nullptr;
        S.plan = dalloc<PcPlan>(B); S.slot_src = dalloc<int>(Ncap); S.slot_step = dalloc<int>(Ncap); S.slot_dead = dalloc<int>(Ncap); HIPCHK(hipMemsetAsync(S.slot_dead, 0xFF, sizeof(int) * Ncap, st)); S.defer_update = 0; S.sort_slot = dalloc<int>(Ncap + 64); S.sort_key = dalloc<unsigned long long>(Ncap + 64);
        S.ctl = dalloc<PcCtl>(1);
        d_total = dalloc<int>(1);
        if (callback_mode) {
            d_cs = (void *)dalloc<char>(pc_chain_state_size() * B); d_x0s = dalloc<double>((size_t)B * D); d_prop = dalloc<double>((size_t)B * D);
            const size_t nans = (size_t)B * (1 + D + std::max(1, nDer));
            d_ans = dalloc<double>(nans);
            d_decks = dalloc<int>((size_t)B * nr);
            hp_prop = halloc<double>((size_t)B * D); hp_ans = halloc<double>(nans); hp_need = halloc<int>(B);
            std::memset(hp_ans, 0, sizeof(double) * nans); std::memset(hp_need, 0, sizeof(int) * B);
            HIPCHK(hipMemset(d_cs, 0, pc_chain_state_size() * B));
        }
        h_ctl = halloc<PcCtl>(1);
        h_note = halloc<PcCtl>(1);                    // (hipHostMalloc memory is mapped and coherent: the device stores straight into it)
        std::memset(h_note, 0, sizeof(PcCtl)); note_seq = 0;
        { void *dp = nullptr; HIPCHK(hipHostGetDevicePointer(&dp, h_note, 0)); S.ctl_host = (PcCtl *)dp; }
        S.notify_seq = 0;
        // per-cluster initial values and the control block (initialise_run_time_info, run_time_info.f90:164-206)
        pc_launch_init_state(&S, c.logzero, st);
        PcCtl c0{};
        c0.status = PC_ST_RUNNING; c0.ncluster = 1; c0.logZ = c.logzero; c0.logZ2 = c.logzero;
        c0.logX_last_update = 0.0; c0.next_cluster_uid = 1; c0.live_logZ = c.logzero;
        *h_ctl = c0;
    }

    void grade_counts(long *o)
    {   // grade 1 carries the prior samples (generate.F90:294); with one grade it is simply nlike
        for (int g = 0; g < PC_MAX_GRADE; ++g) o[g] = 0;
        if (S.ngrade <= 1) { o[0] = (long)h_ctl->nlike; return; }
        long rest = 0;
        for (int g = 1; g < S.ngrade; ++g) { o[g] = (long)nlike_g[g]; rest += o[g]; }
        o[0] = (long)h_ctl->nlike - rest;
    }

    // per-grade likelihood counts (RTI%nlike, nested_sampling.F90:307): the chains the last segment consumed
    void tally_grades()
    {
        if (S.ngrade <= 1) return;
        for (int w = h_ctl->seg_lo; w <= h_ctl->seg_hi; ++w)
            for (int g = 0; g < PC_MAX_GRADE; ++g) nlike_g[g] += h_nlike_g[(size_t)w * PC_MAX_GRADE + g];
    }

    // ---- waiting for the device.  A run on its own synchronises its stream; a run in step, inside a fiber, yields to the driver,
    //      which launches what all runs have written down, waits once for all of them and resumes them (pc_run_cohort)
    void sync_point()
    {
        if (fib) {
            fib->yield();
            if (fib->cancel) {      // resumed to unwind: give back what this wait was for, then out through the frames of round_finish
                for (const Fetch &f : fetching) hfree(f.h);
                fetching.clear(); own_fetches.clear();
                for (void *h : staged_up) hfree(h);
                staged_up.clear();
                throw FiberCancelled{};
            }
            return;
        }
        if (co) co->flush();
        // (polling the stream wakes the host a few microseconds after the copy; the blocking wait sleeps on an interrupt)
        for (int spins = 0; spins < 200000; ++spins) { const hipError_t q = hipStreamQuery(st); if (q != hipErrorNotReady) { HIPCHK(q); break; } __builtin_ia32_pause(); }
        HIPCHK(hipStreamSynchronize(st));
    }
    // something launched here and now, behind whatever the runs in step have written down so far
    void direct_op() { if (co) co->flush(); }
    // device -> host, through a pinned block, in stream order behind everything asked for so far; the values are there after fetch_wait()
    struct Fetch { void *h; void *dst; size_t bytes; };
    std::vector<Fetch> fetching;
    void fetch_raw(void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return;
        void *h = halloc<char>(bytes);
        fetching.push_back({h, dst, bytes});
        hipStream_t q = st;
        if (co && batch_copies()) co->post_copies.push_back({(uintptr_t)h, (uintptr_t)src, (uintptr_t)bytes});
        else if (co) co->post.push_back([h, src, bytes, q] { HIPCHK(hipMemcpyAsync(h, src, bytes, hipMemcpyDeviceToHost, q)); });
        // a run on its own: the same rule as in step -- what is asked for comes down at the wait, in ONE kernel (an update with clustering
        // asks for a dozen small arrays: 890 copy commands a run at configs[2], ~7 us of the stream each)
        else if (batch_copies()) own_fetches.push_back({(uintptr_t)h, (uintptr_t)src, (uintptr_t)bytes});
        else HIPCHK(hipMemcpyAsync(h, src, bytes, hipMemcpyDeviceToHost, st));
    }
    template <class T> void fetch(std::vector<T> &v, const T *p, size_t n) { v.resize(n); fetch_raw(v.data(), p, sizeof(T) * n); }
    std::vector<std::array<uintptr_t, 3>> own_fetches;
    void fetch_wait()
    {
        if (!own_fetches.empty()) { std::vector<std::array<uintptr_t, 3>> c; c.swap(own_fetches); pc_copy_many(c, st); }
        sync_point();
        for (const Fetch &f : fetching) { std::memcpy(f.dst, f.h, f.bytes); hfree(f.h); }
        fetching.clear();
        for (void *h : staged_up) hfree(h);
        staged_up.clear();
    }
    // host -> device, through a pinned block, in stream order (the block goes back at the next wait)
    std::vector<void *> staged_up;
    void send_raw(void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return;
        direct_op();
        void *h = halloc<char>(bytes);
        staged_up.push_back(h);
        std::memcpy(h, src, bytes);
        // (runs in step: with the other copies at the head of the next launch -- what was written down before this call has been launched
        //  by direct_op(), and whatever is launched behind it, here and now or written down, goes through the same door)
        if (co && batch_copies()) co->pre_copies.push_back({(uintptr_t)dst, (uintptr_t)h, (uintptr_t)bytes});
        else HIPCHK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, st));
    }

    // host -> device for kernels that are WRITTEN DOWN after this call (runs in step: the copy is made at the start of the common launch,
    // so the runs' kernels stay together; the destination must not be read by anything this run wrote down before)
    void send_pre(void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return;
        if (!co) { send_raw(dst, src, bytes); return; }
        void *h = halloc<char>(bytes);
        staged_up.push_back(h);
        std::memcpy(h, src, bytes);
        hipStream_t q = st;
        if (batch_copies()) co->pre_copies.push_back({(uintptr_t)dst, (uintptr_t)h, (uintptr_t)bytes});
        else co->pre.push_back([dst, h, bytes, q] { HIPCHK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, q)); });
    }
    static bool batch_copies() { static const bool off = std::getenv("PC_COPY_BATCH_OFF") != nullptr; return !off; }

    void read_ctl()
    {
        fetch_raw(h_ctl_in(), S.ctl, sizeof(PcCtl));
        fetch_wait();
        *h_ctl = ctl_in;
        HIPCHK(hipGetLastError());                    // a kernel that could not be launched must not go unnoticed
        nph_stale = false;
        kt.collect();                                 // (the stream is idle: every open span is complete)
    }
    PcCtl ctl_in;
    PcCtl *h_ctl_in() { return &ctl_in; }

    // The outcome of a round without a copy and without a stream synchronisation: the contraction kernel stamps the
    // host mirror when it is done.  The row copies of the round (k_apply_*) may still be running when this returns;
    // everything the host enqueues next is ordered behind them by the stream.
    void launch_stamp() { S.notify_seq = ++note_seq; }
    unsigned long ready_spins = 0;
    // has the round's contraction kernel reported?  (non-blocking; a stream that finished without a stamp is an error)
    bool round_ready()
    {
        static const bool off = std::getenv("PC_NOTIFY_OFF") != nullptr;
        if (off || !S.ctl_host) { read_ctl(); return true; }
        const volatile unsigned long long *w = (const volatile unsigned long long *)h_note;
        unsigned long long got[PC_NOTE_WORDS];
        bool all = true;
        for (int i = 0; i < PC_NOTE_WORDS; ++i) { got[i] = w[i]; all = all && (unsigned)(got[i] >> 32) == note_seq; }
        if (!all) {
            if ((++ready_spins & 0x3FFFu) == 0x3FFFu) {
                // nothing after ~a millisecond: did the stream die (launch failure, fault)?  A finished stream without a
                // stamp is an error; a busy one just takes long (general contraction kernel, large nurseries)
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    bool ok = true;
                    for (int i = 0; i < PC_NOTE_WORDS; ++i) ok = ok && (unsigned)(w[i] >> 32) == note_seq;
                    if (!ok) engine_fail(PC_RC_DEVICE, "a contraction kernel finished without reporting (launch failure?)");
                } else if (q != hipErrorNotReady) engine_fail(PC_RC_DEVICE, "HIP error %s while waiting for a round", hipGetErrorString(q));
                if (ready_spins > (1ul << 22)) std::this_thread::yield();
            }
            return false;
        }
        PcCtl &c = *h_ctl;
        const int hi_before = c.i_nursery > 0 ? c.i_nursery - 1 : B - 1;       // the segment starts where the last one stopped
        c.status = (int)(got[0] & 0xFF); c.error = (int)((got[0] >> 8) & 0xFF); c.cluster_deleted = (int)((got[0] >> 16) & 1);
        c.upd_pending = (int)((got[0] >> 17) & 1); c.upd_marks = (int)((got[0] >> 18) & 0x3FFF);
        c.i_nursery = (int)((unsigned)got[1] & 0xFFFFu); c.upd_in = (int)(((unsigned)got[1] >> 16) & 0xFFFFu); c.ndead = (int)(unsigned)got[2]; c.nphantom = (int)(unsigned)got[3];
        c.ncluster = (int)(got[4] & 0xFFFF);
        {   // the low 16 bits of a counter that only grows
            int cand = (c.ncluster_dead & ~0xFFFF) | (int)((got[4] >> 16) & 0xFFFF);
            if (cand < c.ncluster_dead) cand += 0x10000;
            c.ncluster_dead = cand;
        }
        c.seg_hi = hi_before; c.seg_lo = c.i_nursery;
        nph_stale = false;
        return true;
    }

    void grow_dead(int nd)
    {
        if (co) co->flush();      // (what the runs in step have written down is launched before an array moves)
        HIPCHK(hipStreamSynchronize(st_copy));        // rows still travelling from the old array
        auto grow = [&](auto *&p, size_t per) {
            using T = std::remove_reference_t<decltype(*p)>;
            T *q = dalloc<T>((size_t)nd * per);
            HIPCHK(hipMemcpyAsync(q, p, sizeof(T) * (size_t)h_ctl->ndead * per, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            dfree(p); p = q;
        };
        grow(S.dead, S.nT); grow(S.dead_logw, 1); grow(S.dead_postX, 1); grow(S.dead_postZ, 1); grow(S.dead_cuid, 1); grow(S.dead_entry, 1);
        S.Dcap = nd;
        // the pinned mirror the rows are streamed into grows with the array (the same capacities in every run: the block comes
        // back from the cache; sized by the final count at the end it was a fresh multi-GB pinning and a second copy of every row)
        if (h_dead && (size_t)nd > h_dead_cap) { hfree(h_dead); h_dead_cap = (size_t)nd; h_dead = halloc<double>(h_dead_cap * S.nT); h_dead_copied = 0; }
    }

    // The reference reallocates its phantom arrays whenever they fill up (run_time_info.f90:747-757 via
    // array_utils reallocate); phantoms are only cleaned at updates, i.e. every -nlive log(compression_factor) deaths,
    // so a small compression_factor needs far more than the initial estimate.
    void grow_phantoms(long long need)
    {
        if (co) co->flush();      // (what the runs in step have written down is launched before an array moves)
        const long long hard = 1LL << 30;             // rows; Pcap and the phantom counters are ints
        if (need > hard) engine_fail(PC_RC_LIMIT, "more than %lld phantom points (%lld needed)", hard, need);
        long long np = std::max<long long>(need, 2LL * S.Pcap);
        np = std::min(np, hard);
        if (g_inject_fault.load() == 3) { g_inject_fault = 0; engine_fail(PC_RC_MEMORY, "out of device memory growing the phantom array to %lld rows (injected)", np); }
        const size_t used = (size_t)std::min<long long>(h_ctl->nphantom, S.Pcap);
        HIPCHK(hipStreamSynchronize(st));
        auto grow = [&](auto *&p, size_t per) {
            using T = std::remove_reference_t<decltype(*p)>;
            T *q = dalloc<T>((size_t)np * per);
            if (used) HIPCHK(hipMemcpyAsync(q, p, sizeof(T) * used * per, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            dfree(p); p = q;
        };
        grow(S.phantom, S.nT); grow(S.ph_logL, 1); grow(S.ph_cuid, 1); grow(S.ph_uid, 1);
        dfree(ph2); dfree(phL2); dfree(phC2); dfree(phU2); dfree(keep); dfree(blk);
        alloc_phantom_side((int)np);
        S.Pcap = (int)np;
    }

    // pool mode: the phantom array is full -- drop what the updates have invalidated (general clean: flags, scan, scatter into
    // the alternate buffers), grow if that is not enough.  Between nurseries only.
    long long pool_cursor = 0;
    double *babies_own = nullptr;
    // the phantom array is full: the phantoms that are still wanted move to the front of the alternate buffers.  In step with other
    // runs the clean is launched for all of them at once and waited for once (compact_wanted / compact_record / compact_finish)
    bool compact_worth_it() const { return S.pool && h_ctl->status == PC_ST_RUNNING && h_ctl->i_nursery == 0 && 2 * pool_cursor > (long long)S.Pcap; }      // (next to a run that has to: pc_run_cohort)
    bool compact_wanted() const { return S.pool && h_ctl->status == PC_ST_RUNNING && h_ctl->i_nursery == 0 && pool_cursor + (long long)B * S.nr > S.Pcap; }
    void compact_record() { co->rec(CK_COMPACT, S, {keep, blk, d_total, ph2, phL2, phC2, phU2}, {}, {0, (int)pool_cursor, ((int)pool_cursor + 255) / 256}); }
    void compact_finish(int total)
    {
        std::swap(S.phantom, ph2); std::swap(S.ph_logL, phL2); std::swap(S.ph_cuid, phC2); std::swap(S.ph_uid, phU2);
        pool_cursor = total; h_ctl->nphantom = total; nph_stale = false;
        if (pool_cursor + (long long)B * S.nr > S.Pcap / 2) grow_phantoms(std::max<long long>(2LL * S.Pcap, 2 * (pool_cursor + (long long)B * S.nr)));
        tm.compactions++;
    }
    void pool_compact()
    {
        const auto dbg_t0 = std::chrono::steady_clock::now();
        struct DbgT { std::chrono::steady_clock::time_point t0; ~DbgT() { g_dbg_compact_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } dbg_t{dbg_t0};
        if (co) co->flush();
        hipEvent_t e0 = kt.begin(KT_CLEAN);
        pc_launch_clean(&S, (int)pool_cursor, keep, blk, d_total, ph2, phL2, phC2, phU2, nullptr, st);
        kt.end(KT_CLEAN, e0);
        int total = 0;
        HIPCHK(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        kt.collect();
        compact_finish(total);
    }
    void ensure_capacity()
    {   // the next batch may append B*nr phantoms and B dead points
        if ((long long)h_ctl->ndead + B + S.Ncap + 16 > S.Dcap) grow_dead(S.Dcap * 2);
        if (S.pool) { if (h_ctl->ncluster_dead + h_ctl->ncluster + 8 > S.maxc_dead) grow_dead_clusters(2 * S.maxc_dead); return; }
        if (nph_stale && (long long)h_ctl->nphantom + (long long)B * S.nr > S.Pcap) read_ctl();   // pre-clean count: refresh
        if ((long long)h_ctl->nphantom + (long long)B * S.nr > S.Pcap) grow_phantoms((long long)h_ctl->nphantom + (long long)B * S.nr);
        // dead clusters: a segment can retire at most the clusters that are active when it starts
        if (h_ctl->ncluster_dead + h_ctl->ncluster + 8 > S.maxc_dead) grow_dead_clusters(2 * S.maxc_dead);
    }

    void grow_dead_clusters(int nd)
    {
        if (co) co->flush();      // (what the runs in step have written down is launched before an array moves)
        HIPCHK(hipStreamSynchronize(st));
        const size_t used = (size_t)std::min(h_ctl->ncluster_dead, S.maxc_dead);
        auto grow = [&](auto *&p) {
            using T = std::remove_reference_t<decltype(*p)>;
            T *q = dalloc<T>((size_t)nd);
            if (used) HIPCHK(hipMemcpy(q, p, sizeof(T) * used, hipMemcpyDeviceToDevice));
            dfree(p); p = q;
        };
        grow(S.logZp_dead); grow(S.logZp2_dead); grow(S.cl_uid_dead);
        S.maxc_dead = nd;
    }

    // The reference has no limit on the number of clusters (add_cluster reallocates every per-cluster array,
    // run_time_info.f90:392-418).  Here the per-cluster arrays are flat with a capacity: grow them the same way.
    void grow_clusters(int need)
    {
        const int mo = S.maxc, mn = std::max(need, 2 * mo), Ncap = S.Ncap, DD = S.D * S.D;
        if (mn > 16384) engine_fail(PC_RC_LIMIT, "more than 16384 clusters");
        if (co) co->flush();
        HIPCHK(hipStreamSynchronize(st));
        auto grow_d = [&](double *&p, double fill) {
            std::vector<double> v = dl(p, (size_t)mo); v.resize(mn, fill);
            dfree(p); p = dalloc<double>(mn); ul(p, v);
        };
        grow_d(S.logZp, cfg.logzero); grow_d(S.logZXp, cfg.logzero); grow_d(S.logZp2, cfg.logzero); grow_d(S.logZpXp, cfg.logzero);
        grow_d(S.logLp, cfg.logzero); grow_d(S.logXp, 0.0); grow_d(S.lse_ref, 0.0); grow_d(S.lse_sum, 0.0); grow_d(S.death_thr, -PC_HUGE);
        { std::vector<int> v = dl(S.cl_n, (size_t)mo); v.resize(mn, 0); dfree(S.cl_n); S.cl_n = dalloc<int>(mn); ul(S.cl_n, v); }
        { std::vector<int> v = dl(S.imin_slot, (size_t)mo); v.resize(mn, 0); dfree(S.imin_slot); S.imin_slot = dalloc<int>(mn); ul(S.imin_slot, v); }
        { std::vector<unsigned> v = dl(S.cl_uid, (size_t)mo); v.resize(mn, 0u); dfree(S.cl_uid); S.cl_uid = dalloc<unsigned>(mn); ul(S.cl_uid, v); }
        {   // the cross-volume matrix changes its leading dimension
            std::vector<double> o = dl(S.XpXq, (size_t)mo * mo), v((size_t)mn * mn, 0.0);
            for (int a = 0; a < mo; ++a) std::copy(o.begin() + (size_t)a * mo, o.begin() + (size_t)(a + 1) * mo, v.begin() + (size_t)a * mn);
            dfree(S.XpXq); S.XpXq = dalloc<double>(2 * (size_t)mn * mn); ul(S.XpXq, v);
        }
        {
            int *q = dalloc<int>((size_t)mn * Ncap);
            HIPCHK(hipMemcpy(q, S.cl_list, sizeof(int) * (size_t)mo * Ncap, hipMemcpyDeviceToDevice));
            dfree(S.cl_list); S.cl_list = q;
        }
        auto grow_mat = [&](double *&p) {
            double *q = dalloc<double>((size_t)mn * DD);
            HIPCHK(hipMemcpy(q, p, sizeof(double) * (size_t)mo * DD, hipMemcpyDeviceToDevice));
            std::vector<double> id((size_t)(mn - mo) * DD, 0.0);
            for (int c = 0; c < mn - mo; ++c) for (int a = 0; a < S.D; ++a) id[(size_t)c * DD + (size_t)a * S.D + a] = 1.0;
            HIPCHK(hipMemcpy(q + (size_t)mo * DD, id.data(), sizeof(double) * id.size(), hipMemcpyHostToDevice));
            dfree(p); p = q;
        };
        grow_mat(S.chol); grow_mat(S.cov);
        if (c_cnt) { dfree(c_cnt); dfree(c_olduid); c_cnt = dalloc<int>(mn); c_olduid = dalloc<unsigned>(mn); }
        dfree(psum); dfree(pcnt); dfree(pcov); dfree(mean); dfree(count); cov_chunks_cap = 0;   // sized with maxc: covmats() reallocates
        S.maxc = mn;
    }

    // dump (nested_sampling.F90:546-590): live and dead points as [theta, phi, birth, logL] rows,
    // posterior log-weights normalised to logsumexp 0
    // stage 0: update, 1: after the kill-off, 2: the initial live points before any death (prior files)
    void call_dumper(int stage = 0)
    {
        if (!(dumper && stage != 2) && !on_update) return;
        const int nT = S.nT, D = S.D, nDer = S.nDer, npars = D + nDer + 2, nd = h_ctl->ndead;
        HIPCHK(hipStreamSynchronize(st));
        if (nd > hm_ndead) {
            std::vector<double> rows((size_t)(nd - hm_ndead) * nT), lw(nd - hm_ndead);
            hm_cuid.resize(nd);
            HIPCHK(hipMemcpy(hm_cuid.data() + hm_ndead, S.dead_cuid + hm_ndead, sizeof(unsigned) * (nd - hm_ndead), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(rows.data(), S.dead + (size_t)hm_ndead * nT, sizeof(double) * rows.size(), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(lw.data(), S.dead_logw + hm_ndead, sizeof(double) * lw.size(), hipMemcpyDeviceToHost));
            hm_dead.resize((size_t)nd * npars); hm_logw.resize(nd);
            for (int i = hm_ndead; i < nd; ++i) {
                const double *r = rows.data() + (size_t)(i - hm_ndead) * nT;
                double *o = hm_dead.data() + (size_t)i * npars;
                std::memcpy(o, r + S.p0, sizeof(double) * (D + nDer)); o[D + nDer] = r[S.b0]; o[D + nDer + 1] = r[S.l0];
                hm_logw[i] = lw[i - hm_ndead] + r[S.l0];
            }
            hm_ndead = nd;
        }
        std::vector<double> lwn(hm_logw.begin(), hm_logw.begin() + nd);
        double m = -PC_HUGE;
        for (double v : lwn) m = std::max(m, v);
        double sum = 0.0;
        for (double v : lwn) sum += std::exp(v - m);
        const double lse = m + std::log(sum);
        for (double &v : lwn) v -= lse;
        // live points ordered by cluster, then list position
        auto lc = dl(S.live_cluster, S.Ncap); auto lp = dl(S.live_pos, S.Ncap); auto lr = dl(S.live, (size_t)S.Ncap * nT);
        std::vector<std::pair<long long, int>> ord;
        for (int s = 0; s < S.Ncap; ++s) if (lc[s] >= 0) ord.push_back({(long long)lc[s] * S.Ncap + lp[s], s});
        std::sort(ord.begin(), ord.end());
        std::vector<double> live(std::max<size_t>(1, ord.size()) * npars);
        for (size_t k = 0; k < ord.size(); ++k) {
            const double *r = lr.data() + (size_t)ord[k].second * nT;
            double *o = live.data() + k * npars;
            std::memcpy(o, r + S.p0, sizeof(double) * (D + nDer)); o[D + nDer] = r[S.b0]; o[D + nDer + 1] = r[S.l0];
        }
        const double lz = std::max(-PC_HUGE, 2 * h_ctl->logZ - 0.5 * h_ctl->logZ2), var = h_ctl->logZ2 - 2 * h_ctl->logZ;
        double dummy = 0.0;
        if (dumper && stage != 2) dumper(nd, (int)ord.size(), npars, live.data(), nd > 0 ? hm_dead.data() : &dummy, nd > 0 ? lwn.data() : &dummy, lz, std::sqrt(std::fabs(var)));
        if (on_update) {
            const int nc = h_ctl->ncluster, ncd = std::min(h_ctl->ncluster_dead, S.maxc_dead);
            std::vector<int> lcl(std::max<size_t>(1, ord.size()));
            for (size_t k = 0; k < ord.size(); ++k) lcl[k] = lc[ord[k].second];
            auto zp = dl(S.logZp, std::max(1, nc)), zp2 = dl(S.logZp2, std::max(1, nc));
            auto zd = dl(S.logZp_dead, std::max(1, ncd)), zd2 = dl(S.logZp2_dead, std::max(1, ncd));
            auto cn = dl(S.cl_n, std::max(1, nc));
            std::vector<double> e1(std::max(1, nc)), s1(std::max(1, nc)), e2(std::max(1, ncd)), s2(std::max(1, ncd));
            for (int i = 0; i < nc; ++i) { e1[i] = 2 * zp[i] - 0.5 * zp2[i]; s1[i] = std::sqrt(std::fabs(zp2[i] - 2 * zp[i])); }   // run_time_info.f90:652-678
            for (int i = 0; i < ncd; ++i) { e2[i] = 2 * zd[i] - 0.5 * zd2[i]; s2[i] = std::sqrt(std::fabs(zd2[i] - 2 * zd[i])); }
            pchip_update u{};
            u.final_call = stage; u.ndiscarded = ndiscarded; u.ndead = nd; u.nlive = (int)ord.size(); u.npars = npars;
            u.dead = nd > 0 ? hm_dead.data() : &dummy; u.logpost = nd > 0 ? hm_logw.data() : &dummy;
            u.live = live.data(); u.live_cluster = lcl.data();
            u.logZ = lz; u.logZerr = std::sqrt(std::fabs(var)); u.nlike = h_ctl->nlike;
            long ng[PC_MAX_GRADE]; int gdims[PC_MAX_GRADE];
            grade_counts(ng);
            for (int g = 0; g < S.ngrade; ++g) gdims[g] = (g + 1 < S.ngrade ? S.g_off[g + 1] : D) - S.g_off[g];
            u.ngrade = S.ngrade; u.nlike_grade = ng; u.grade_dims = gdims; u.grade_repeats = S.g_nr;
            u.ncluster = nc; u.ncluster_dead = ncd; u.nlive_p = cn.data();
            u.logZp = e1.data(); u.logZperr = s1.data(); u.logZp_dead = e2.data(); u.logZperr_dead = s2.data();
            auto ua = dl(S.cl_uid, std::max(1, nc)); auto ud = dl(S.cl_uid_dead, std::max(1, ncd));
            unsigned dummy_u = 0u;
            u.dead_cluster = nd > 0 ? hm_cuid.data() : &dummy_u; u.cluster_uid = ua.data(); u.cluster_uid_dead = ud.data();
            u.nsplit = (int)split_child.size(); u.split_child = split_child.data(); u.split_parent = split_parent.data();
            u.split_logfrac = split_logfrac.data();
            u.n_extra = (int)pp_logpost.size(); u.extra = pp_rows.data(); u.extra_logpost = pp_logpost.data(); u.extra_cluster = pp_cuid.data(); u.extra_uid = pp_uid.data();
            on_update(hook_user, &u);
        }
    }

    // clean_phantoms + calculate_covmats (nested_sampling.F90:326-368 minus file output / clustering)
    // clean_phantoms also feeds the posterior (run_time_info.f90:845-870): a phantom that falls below the contour is
    // kept with probability thin_posterior = boost_posterior / num_repeats (generate.F90:311-316) and carries the
    // weight of the first point of its cluster that died since the last update with a larger logL.  Host side: the
    // feature is off by default and touches each phantom once per update.
    void collect_phantom_posteriors(int nph)
    {
        const double thin = cfg.boost_posterior < 0.0 ? 1.0 : cfg.boost_posterior / (double)S.nr;
        const int nd0 = nd_last_update, nd = h_ctl->ndead, nT = S.nT, np = S.D + S.nDer;
        nd_last_update = nd;
        if (!(thin > 0.0) || nph <= 0 || nd <= nd0) return;
        HIPCHK(hipStreamSynchronize(st));
        auto kp = dl(keep, nph); auto phL = dl(S.ph_logL, nph); auto phC = dl(S.ph_cuid, nph); auto phU = dl(S.ph_uid, nph);
        const int m = nd - nd0;
        std::vector<double> dL(m), dW = dl(S.dead_logw + nd0, m); auto dC = dl(S.dead_cuid + nd0, m);
        HIPCHK(hipMemcpy2D(dL.data(), sizeof(double), S.dead + (size_t)nd0 * nT + S.l0, sizeof(double) * nT, sizeof(double), m, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<std::pair<double, double>>> stack;      // per cluster: (logL, logweight), death order
        for (int i = 0; i < m; ++i) if (dW[i] > cfg.logzero) stack[dC[i]].push_back({dL[i], dW[i]});
        std::vector<int> pick; std::vector<double> pickw;
        for (int j = 0; j < nph; ++j) {
            if (kp[j]) continue;
            const unsigned long long uid = phU[j];
            if (!(polychord_hip_keyed_uniform((unsigned)cfg.seed, PC_DOM_PHANTOM, (unsigned)(uid >> 32), (unsigned)uid, 0u) < thin)) continue;
            auto it = stack.find(phC[j]);
            if (it == stack.end()) continue;
            const auto &v = it->second;
            // deaths of one cluster arrive in ascending logL: first entry above the phantom
            size_t lo = 0, hi = v.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (v[mid].first > phL[j]) hi = mid; else lo = mid + 1; }
            if (lo == v.size()) continue;
            pick.push_back(j); pickw.push_back(v[lo].second);
        }
        if (pick.empty()) return;
        std::vector<double> row(nT);
        for (size_t k = 0; k < pick.size(); ++k) {
            HIPCHK(hipMemcpy(row.data(), S.phantom + (size_t)pick[k] * nT, sizeof(double) * nT, hipMemcpyDeviceToHost));
            const size_t o = pp_rows.size();
            pp_rows.resize(o + np + 2);
            std::memcpy(pp_rows.data() + o, row.data() + S.p0, sizeof(double) * np);
            pp_rows[o + np] = row[S.b0]; pp_rows[o + np + 1] = row[S.l0];
            pp_logpost.push_back(pickw[k] + row[S.l0]); pp_cuid.push_back(phC[pick[k]]); pp_uid.push_back(phU[pick[k]]);
        }
    }

    // Sequential-stream test mode with posteriors = T: clean_phantoms draws one Bernoulli uniform for every phantom it
    // removes (run_time_info.f90:857-859, even when thin_posterior = 0), all from the one generator: the engine's stream
    // position moves on by as many, so that the next nursery draws what the reference binary's draws.  (The trials of
    // update_posteriors for equals = T depend on the order of the reference's arrays and are not emulated: with
    // equals = T the sequential mode is a valid run, not the reference's.)
    void seq_consume(unsigned long long n)
    {
        if (!n) return;
        h_ctl->seq += n;
        HIPCHK(hipMemcpy(&S.ctl->seq, &h_ctl->seq, sizeof(unsigned long long), hipMemcpyHostToDevice));
    }

    void do_update(bool deferred = false)
    {
        tm.updates += deferred ? std::max(1, h_ctl->upd_marks) : 1;
        // the round loop only sees the compact notification; whoever looks at evidences, counters or cluster ids gets the block
        const bool seq_post = S.seq_mode && (cfg.posteriors || cfg.equals);
        // (clustering alone: the block comes with the counts below, in the same wait)
        const bool ctl_late = cfg.do_clustering && !(dumper || on_update || cfg.resume_write || cfg.boost_posterior != 0.0 || seq_post);
        if (!ctl_late && (dumper || on_update || cfg.do_clustering || cfg.resume_write || cfg.boost_posterior != 0.0 || seq_post)) { const int st_keep = h_ctl->status; read_ctl(); h_ctl->status = st_keep; }
        // (the reference makes update_posteriors -- clean_phantoms with it -- BEFORE it writes files and calls the dumper,
        //  nested_sampling.F90:325-336: when phantoms can join the posterior (boost_posterior) the hook waits for this update's)
        const bool hook_late = cfg.boost_posterior != 0.0 && (cfg.posteriors || cfg.equals);
        if (!hook_late) call_dumper();
        const int nph = S.pool ? (int)pool_cursor : h_ctl->nphantom;
        static const bool fused_off = std::getenv("PC_UPDATE_FUSED_OFF") != nullptr;
        if (!fused_off && !(cfg.ablate & 8) && nph > 0 && !cfg.do_clustering && cfg.boost_posterior == 0.0 && pc_update_fused_ok(&S, h_ctl->ncluster)) {
            // one cluster, nDims < 32: clean + covariance + Cholesky in three launches (pc_update.hip)
            const size_t need = (size_t)pc_update_fused_blocks(&S, nph) * pc_update_fused_entries(&S);
            if (need > upd_part_cap) { dfree(upd_part); upd_part_cap = 2 * need; upd_part = dalloc<double>(upd_part_cap); }
            if (!upd_shift) {
                upd_shift = dalloc<double>(S.D);
                std::vector<double> half(S.D, 0.5);               // first update: moments about the centre of the hypercube
                HIPCHK(hipMemcpyAsync(upd_shift, half.data(), sizeof(double) * S.D, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
            }
            path[PCHIP_PATH_UPDATE_FUSED]++;
            hipEvent_t e0 = kt.begin(KT_CLEAN);
            if (co) co->rec(CK_UPDATE, S, {keep, blk, d_total, ph2, phL2, phC2, phU2, upd_part, upd_shift}, {(long long)pc_update_fused_grid(&S, nph, deferred ? 1 : 0), deferred ? 1LL : 0LL}, {0, nph, (nph + 255) / 256});
            else pc_launch_update_fused(&S, nph, keep, blk, d_total, ph2, phL2, phC2, phU2, upd_part, upd_shift, deferred ? 1 : 0, st);
            kt.end(KT_CLEAN, e0);
            if (cfg.resume_write || dumper || on_update || seq_post) {
                // (in step with other runs the update was only written down: the copy of its count comes behind its launch)
                std::vector<int> tot;
                fetch(tot, (const int *)d_total, 1);
                fetch_wait();
                const int total = tot[0];
                h_ctl->nphantom = total;
                if (seq_post) seq_consume((unsigned long long)(nph - total));
            } else nph_stale = true;
            if (!S.pool) { std::swap(S.phantom, ph2); std::swap(S.ph_logL, phL2); std::swap(S.ph_cuid, phC2); std::swap(S.ph_uid, phU2); }
            write_resume();
            return;
        }
        if (deferred) engine_fail(PC_RC_DEVICE, "deferred update without the fused update path");
        path[PCHIP_PATH_UPDATE_STEPS]++;
        hipEvent_t e0 = kt.begin(KT_CLEAN);
        // (in step with other runs: the clean of all runs that update in this round is one launch, like the pool compaction)
        if (co && nph > 0) co->rec(CK_COMPACT, S, {keep, blk, d_total, ph2, phL2, phC2, phU2}, {}, {0, nph, (nph + 255) / 256});
        else { direct_op(); pc_launch_clean(&S, nph, keep, blk, d_total, ph2, phL2, phC2, phU2, nullptr, st); }
        kt.end(KT_CLEAN, e0);
        if (cfg.boost_posterior != 0.0 && (cfg.posteriors || cfg.equals)) collect_phantom_posteriors(nph);
        if (hook_late) call_dumper();
        // The surviving count is written to the control block on the device.  Without clustering / resume files
        // nothing on the host needs it before the next round's read-back, so the update costs no extra sync: the
        // covariance grid is sized with the pre-clean count and the kernels clamp to the device value.
        const bool need_count = cfg.do_clustering || cfg.resume_write || dumper || on_update || cfg.boost_posterior != 0.0 || seq_post;
        int total = nph;
        std::vector<int> tot, cn;
        if (need_count) {
            fetch(tot, (const int *)d_total, 1);
            if (cfg.do_clustering) fetch(cn, (const int *)S.cl_n, (size_t)h_ctl->ncluster);      // (the clusters' sizes for do_clustering: the same wait)
            if (ctl_late) fetch_raw(&ctl_in, S.ctl, sizeof(PcCtl));
        } else nph_stale = true;
        std::swap(S.phantom, ph2); std::swap(S.ph_logL, phL2); std::swap(S.ph_cuid, phC2); std::swap(S.ph_uid, phU2);
        if (co) co->rec(CK_RESET, S, {}, {}, {}); else pc_launch_reset_thresholds(&S, st);
        if (need_count) {
            fetch_wait();
            if (ctl_late) { const int st_keep = h_ctl->status; *h_ctl = ctl_in; h_ctl->status = st_keep; nph_stale = false; }
            total = tot[0];
            h_ctl->nphantom = total;
            if (seq_post) seq_consume((unsigned long long)(nph - total));
        }
        if (cfg.do_clustering) do_clustering(cn);
        hipEvent_t e1 = kt.begin(KT_COV);
        covmats(total, h_ctl->ncluster);
        kt.end(KT_COV, e1);
        write_resume();                               // nested_sampling.F90:337
    }

    // ---- kNN clustering (clustering.f90:253-324); heavy parts on the device (pc_cluster.hip)
    static int relabel_host(std::vector<int> &lab)
    {   // utils.F90:713-749
        std::vector<int> map; std::vector<int> out(lab.size());
        for (size_t i = 0; i < lab.size(); ++i) {
            int f = -1;
            for (size_t k = 0; k < map.size(); ++k) if (map[k] == lab[i]) { f = (int)k; break; }
            if (f < 0) { map.push_back(lab[i]); f = (int)map.size() - 1; }
            out[i] = f + 1;
        }
        lab.swap(out);
        return (int)map.size();
    }

    // NN_clustering on the subset `gidx` (indices into the root cluster's point order); recursion
    // over the found clusters as in clustering.f90:80-95
    int nn_clustering(int nroot, const std::vector<int> &gidx, std::vector<int> &labels)
    {
        const int m = (int)gidx.size();
        labels.assign(m, 1);
        if (m <= 1) return 1;
        send_raw(c_gidx, gidx.data(), sizeof(int) * m);
        direct_op();
        if (pc_launch_knn_cluster(c_Sm, nroot, c_gidx, m, c_knn, c_lab, c_out, st)) engine_fail(PC_RC_LDS, "cluster of %d points too large for the LDS kNN sort", m);
        std::vector<int> numv;
        fetch_raw(labels.data(), c_lab, sizeof(int) * m);
        fetch(numv, (const int *)c_out, 1);
        fetch_wait();
        int num = numv[0];
        if (num > 1) {
            int ic = 1;
            while (ic <= num) {
                std::vector<int> pts, sub;
                for (int j = 0; j < m; ++j) if (labels[j] == ic) { pts.push_back(j); sub.push_back(gidx[j]); }
                std::vector<int> sl;
                const int nnew = nn_clustering(nroot, sub, sl);
                for (size_t a = 0; a < pts.size(); ++a) labels[pts[a]] = num + sl[a];
                if (nnew == 1) ic++;
                num = relabel_host(labels);
            }
        }
        return num;
    }

    void ensure_cluster_scratch()
    {
        if (c_cap >= S.Ncap) return;
        c_cap = S.Ncap;
        c_Sm = dalloc<double>((size_t)c_cap * c_cap); c_pts = dalloc<int>(c_cap); c_gidx = dalloc<int>(c_cap);
        c_knn = dalloc<int>((size_t)c_cap * c_cap); c_lab = dalloc<int>(c_cap); c_out = dalloc<int>(4);
        c_cnt = dalloc<int>(S.maxc); c_olduid = dalloc<unsigned>(S.maxc);
    }

    // (in stream order, through pinned blocks: a run in step shares the wait with the others)
    template <class T> std::vector<T> dl(const T *p, size_t n)
    {
        std::vector<T> v;
        fetch(v, p, n);
        fetch_wait();
        return v;
    }
    template <class T> void ul(T *p, const std::vector<T> &v) { send_raw(p, v.data(), sizeof(T) * v.size()); }

    // What the host's half of a split reads: the points' (cluster, position) labels and the clusters' volumes, evidences, thresholds,
    // ids, cross-volume rows -- the first nc entries / rows; what lies behind them stays what it is on the device.  Asked for with the
    // verdicts of the update's first clustering pass (do_clustering: the same wait), it serves every split of the update: a split
    // leaves on the host exactly what it sends up, so the splits of an update cost one wait each (the phantoms' counts), not two.
    struct ClusterMirror {
        bool valid = false; int maxc = 0;
        std::vector<int> lc, lp; std::vector<double> Xp, ZXp, Zp, Zp2, ZpXp, thr, XQ; std::vector<unsigned> uid;
    } cmir;
    void cmir_ask()
    {
        const int nc = h_ctl->ncluster, maxc = S.maxc, Ncap = S.Ncap;
        cmir.lc.resize(Ncap); cmir.lp.resize(Ncap);
        for (std::vector<double> *v : {&cmir.Xp, &cmir.ZXp, &cmir.Zp, &cmir.Zp2, &cmir.ZpXp, &cmir.thr}) v->resize(maxc);
        cmir.XQ.resize((size_t)maxc * maxc); cmir.uid.resize(maxc);
        fetch_raw(cmir.lc.data(), S.live_cluster, sizeof(int) * Ncap); fetch_raw(cmir.lp.data(), S.live_pos, sizeof(int) * Ncap);
        fetch_raw(cmir.Xp.data(), S.logXp, sizeof(double) * nc); fetch_raw(cmir.ZXp.data(), S.logZXp, sizeof(double) * nc);
        fetch_raw(cmir.Zp.data(), S.logZp, sizeof(double) * nc); fetch_raw(cmir.Zp2.data(), S.logZp2, sizeof(double) * nc);
        fetch_raw(cmir.ZpXp.data(), S.logZpXp, sizeof(double) * nc); fetch_raw(cmir.thr.data(), S.death_thr, sizeof(double) * nc);
        fetch_raw(cmir.XQ.data(), S.XpXq, sizeof(double) * (size_t)nc * maxc); fetch_raw(cmir.uid.data(), S.cl_uid, sizeof(unsigned) * nc);
        cmir.maxc = maxc;
    }

    // add_cluster (run_time_info.f90:303-505): cluster p splits into nnew clusters appended at the end
    void add_cluster(int p, const std::vector<int> &labels, int nnew)
    {
        const int nc = h_ctl->ncluster, nold = nc - 1, ncn = nc + nnew - 1, Ncap = S.Ncap;
        if (g_inject_fault.load() == 2) { g_inject_fault = 0; engine_fail(PC_RC_LIMIT, "more than %d clusters (injected)", nc); }
        if (ncn > S.maxc) { grow_clusters(ncn); cmir.valid = false; }
        const int maxc = S.maxc;
        nsplits++;
        // everything the host's half of the split reads: there since the update's first pass, or asked for now in ONE wait (a run in step
        // shares it with the others)
        if (!cmir.valid || cmir.maxc != maxc) { cmir_ask(); fetch_wait(); cmir.valid = true; }
        std::vector<int> &lc = cmir.lc, &lp = cmir.lp; std::vector<double> &Xp = cmir.Xp, &ZXp = cmir.ZXp, &Zp = cmir.Zp, &Zp2 = cmir.Zp2, &ZpXp = cmir.ZpXp, &thr = cmir.thr, &XQ = cmir.XQ;
        std::vector<unsigned> &uid = cmir.uid;
        auto uln = [&](auto *dst, const auto &v, size_t n) { send_raw(dst, v.data(), sizeof(v[0]) * n); };
        // position of every split point inside its new cluster = rank among equal labels in list order
        std::vector<int> posnew(labels.size()), cnt(nnew, 0);
        for (size_t a = 0; a < labels.size(); ++a) posnew[a] = cnt[labels[a] - 1]++;
        for (int s = 0; s < Ncap; ++s) {
            const int c = lc[s];
            if (c < 0) continue;
            if (c == p) { const int a = lp[s]; lc[s] = nold + labels[a] - 1; lp[s] = posnew[a]; }
            else if (c > p) lc[s] = c - 1;
        }
        ul(S.live_cluster, lc); ul(S.live_pos, lp);
        // per-cluster state: old clusters keep their order at 0..nold-1 (old_save/old_target, :371-376)
        std::vector<unsigned> olduid(uid.begin(), uid.begin() + nc);
        send_raw(c_olduid, olduid.data(), sizeof(unsigned) * nc);
        const double logXp = Xp[p], logXp2 = XQ[(size_t)p * maxc + p], logZp = Zp[p], logZp2 = Zp2[p], logZXp = ZXp[p], logZpXp = ZpXp[p];
        std::vector<double> rowpq;
        for (int q = 0; q < nc; ++q) if (q != p) rowpq.push_back(XQ[(size_t)p * maxc + q]);
        auto shift = [&](std::vector<double> &v) { for (int c = p; c < nc - 1; ++c) v[c] = v[c + 1]; };
        shift(Xp); shift(ZXp); shift(Zp); shift(Zp2); shift(ZpXp); shift(thr);
        for (int c = p; c < nc - 1; ++c) uid[c] = uid[c + 1];
        {
            std::vector<double> t(XQ);
            for (int a = 0, na = 0; a < nc; ++a) { if (a == p) continue; for (int b = 0, nb = 0; b < nc; ++b) { if (b == p) continue; XQ[(size_t)na * maxc + nb] = t[(size_t)a * maxc + b]; nb++; } na++; }
        }
        // (the Cholesky factors and covariances of the clusters behind p move up a block: one launch, not two copies per cluster)
        direct_op();
        pc_launch_shift_mats(&S, p, nc, st);
        for (int k = 0; k < nnew; ++k) { uid[nold + k] = h_ctl->next_cluster_uid++; thr[nold + k] = -PC_HUGE; }
        uln(S.cl_uid, uid, (size_t)ncn); uln(S.death_thr, thr, (size_t)ncn);
        // lists, contours, live log-sum-exp of every cluster; then the phantoms find their new homes
        direct_op();
        pc_launch_rebuild(&S, ncn, st);
        pc_launch_ph_rehome(&S, h_ctl->nphantom, ncn, c_olduid, nc, c_cnt, st);
        std::vector<int> nph, nlv;
        fetch(nph, (const int *)c_cnt, ncn); fetch(nlv, (const int *)S.cl_n, ncn);
        fetch_wait();
        // 5) evidences and volumes split in proportion to nlive + nphantom (:458-503)
        std::vector<double> logni(nnew), logni1(nnew);
        for (int k = 0; k < nnew; ++k) { logni[k] = std::log((double)(nlv[nold + k] + nph[nold + k]) + 0.0); logni1[k] = std::log((double)(nlv[nold + k] + nph[nold + k]) + 1.0); }
        double mx = logni[0];
        for (int k = 1; k < nnew; ++k) mx = std::max(mx, logni[k]);
        double sm = 0.0;
        for (int k = 0; k < nnew; ++k) sm += std::exp(logni[k] - mx);
        const double logn = mx + std::log(sm);
        const double logn1 = logn > 0.0 ? logn + std::log(std::exp(0.0 - logn) + 1.0) : 0.0 + std::log(std::exp(logn - 0.0) + 1.0);
        for (int k = 0; k < nnew; ++k) { split_child.push_back(uid[nold + k]); split_parent.push_back(olduid[p]); split_logfrac.push_back(logni[k] - logn); }
        for (int k = 0; k < nnew; ++k) {
            const int c = nold + k;
            Xp[c] = logXp + logni[k] - logn; ZXp[c] = logZXp + logni[k] - logn; Zp[c] = logZp + logni[k] - logn;
            Zp2[c] = logZp2 + logni[k] + logni1[k] - logn - logn1; ZpXp[c] = logZpXp + logni[k] + logni1[k] - logn - logn1;
            for (int q = 0; q < nold; ++q) { XQ[(size_t)c * maxc + q] = rowpq[q] + logni[k] - logn; XQ[(size_t)q * maxc + c] = XQ[(size_t)c * maxc + q]; }
        }
        for (int a = 0; a < nnew; ++a)
            for (int b = 0; b < nnew; ++b)
                XQ[(size_t)(nold + a) * maxc + nold + b] = (a == b) ? logXp2 + logni[a] + logni1[a] - logn - logn1
                                                                     : logXp2 + logni[a] + logni[b] - logn - logn1;
        uln(S.logXp, Xp, (size_t)ncn); uln(S.logZXp, ZXp, (size_t)ncn); uln(S.logZp, Zp, (size_t)ncn); uln(S.logZp2, Zp2, (size_t)ncn); uln(S.logZpXp, ZpXp, (size_t)ncn);
        uln(S.XpXq, XQ, (size_t)ncn * maxc);
        h_ctl->ncluster = ncn;
        ncluster_peak = std::max(ncluster_peak, ncn);
    }

    // do_clustering (clustering.f90:253-324).  First pass: every cluster with more than two points at once (three launches, the
    // counts down and the verdicts back: two host waits per update); a cluster in which the pass finds more than one group
    // goes through the per-cluster path with its recursion and add_cluster, in the reference's order
    int *c_desc = nullptr, *c_bout = nullptr; int c_desc_cap = 0;
    int *c_map = nullptr; int c_map_cap = 0;
    int *c_gdesc = nullptr, *c_gpool = nullptr, *c_glab = nullptr, *c_gout = nullptr; int c_g_cap = 0;
    // NN_clustering's recursion (clustering.f90:80-95) level by level.  The reference re-clusters every cluster it finds, alone, until one
    // pass over it finds a single cluster; the labels it returns are the final parts numbered by first appearance (relabel after every
    // step, utils.F90:713-749).  A part's own clustering depends on its points only, so the order in which the parts are looked at does
    // not matter: all parts of all clusters of this update that are still open are clustered in ONE launch per level (and, in step
    // with other runs, together with theirs), on the similarity blocks the first pass left behind -- two or three waits per update
    // instead of one per part.  desc: the first pass' descriptors {cluster, n, off2, off1}; out: clusters it found; lab0: its labels.
    bool refine_partitions(const std::vector<int> &desc, const std::vector<int> &which, const std::vector<int> &out, const std::vector<int> &lab0,
                           std::vector<std::vector<int>> &final_labels, std::vector<int> &final_num)
    {
        struct Part { int k; std::vector<int> idx; };              // k: descriptor; idx: positions in the cluster's point order
        const int nd = (int)which.size();
        std::vector<std::vector<std::vector<int>>> done((size_t)nd);       // final parts of every split cluster
        std::vector<Part> work;
        auto split_by = [&](int k, const std::vector<int> &idx, const int *lab /* 1-based, one per entry of idx */, int num) {
            std::vector<std::vector<int>> parts((size_t)num);
            for (size_t a = 0; a < idx.size(); ++a) parts[(size_t)lab[a] - 1].push_back(idx[a]);
            for (auto &pt : parts) { if (pt.size() > 1) work.push_back(Part{k, std::move(pt)}); else if (!pt.empty()) done[(size_t)k].push_back(std::move(pt)); }
        };
        for (int k = 0; k < nd; ++k) {
            if (out[k] <= 1) continue;
            const int n = desc[4 * k + 1], o1 = desc[4 * k + 3];
            std::vector<int> all((size_t)n);
            for (int i = 0; i < n; ++i) all[i] = i;
            split_by(k, all, lab0.data() + o1, out[k]);
        }
        while (!work.empty()) {
            std::vector<Part> cur; cur.swap(work);
            const int nb = (int)cur.size();
            std::vector<int> gdesc((size_t)5 * nb), pool;
            int mmax = 0; long long koff = 0;
            for (int b = 0; b < nb; ++b) {
                const int k = cur[b].k, m = (int)cur[b].idx.size();
                gdesc[5 * b + 0] = desc[4 * k + 2]; gdesc[5 * b + 1] = desc[4 * k + 1]; gdesc[5 * b + 2] = (int)pool.size(); gdesc[5 * b + 3] = m; gdesc[5 * b + 4] = (int)koff;
                pool.insert(pool.end(), cur[b].idx.begin(), cur[b].idx.end());
                mmax = std::max(mmax, m); koff += (long long)m * m;
            }
            if (koff > (long long)c_cap * c_cap) return false;           // (cannot happen: the parts of a cluster are disjoint)
            if (c_g_cap < std::max(nb, (int)pool.size())) {
                dfree(c_gdesc); dfree(c_gpool); dfree(c_glab); dfree(c_gout);
                c_g_cap = std::max(2 * std::max(nb, (int)pool.size()), S.Ncap);
                c_gdesc = dalloc<int>((size_t)5 * c_g_cap); c_gpool = dalloc<int>(c_g_cap); c_glab = dalloc<int>(c_g_cap); c_gout = dalloc<int>(c_g_cap);
            }
            send_pre(c_gdesc, gdesc.data(), sizeof(int) * gdesc.size());
            send_pre(c_gpool, pool.data(), sizeof(int) * pool.size());
            if (co) co->rec(CK_CLUSG, S, {c_gdesc, c_Sm, c_gpool, c_knn, c_glab, c_gout}, {}, {0, nb, mmax});
            else if (pc_launch_knn_cluster_sub(c_gdesc, nb, mmax, c_Sm, c_gpool, c_knn, c_glab, c_gout, st)) engine_fail(PC_RC_LDS, "cluster of %d points too large for the LDS kNN sort", mmax);
            std::vector<int> labs, nums;
            fetch(labs, (const int *)c_glab, pool.size()); fetch(nums, (const int *)c_gout, (size_t)nb);
            fetch_wait();
            for (int b = 0; b < nb; ++b) {
                if (nums[b] > 1) split_by(cur[b].k, cur[b].idx, labs.data() + gdesc[5 * b + 2], nums[b]);
                else done[(size_t)cur[b].k].push_back(std::move(cur[b].idx));
            }
        }
        for (int k = 0; k < nd; ++k) {
            if (out[k] <= 1) continue;
            const int n = desc[4 * k + 1], j = which[k];
            std::vector<int> part_of((size_t)n, -1), newlab(done[(size_t)k].size(), 0);
            for (size_t q = 0; q < done[(size_t)k].size(); ++q) for (int i : done[(size_t)k][q]) part_of[(size_t)i] = (int)q;
            int next = 0;
            final_labels[(size_t)j].assign((size_t)n, 0);
            for (int i = 0; i < n; ++i) { int &l = newlab[(size_t)part_of[(size_t)i]]; if (l == 0) l = ++next; final_labels[(size_t)j][(size_t)i] = l; }
            final_num[(size_t)j] = next;
        }
        return true;
    }
    bool do_clustering(std::vector<int> cn = std::vector<int>())
    {
        ensure_cluster_scratch();
        bool found = false;
        cmir.valid = false;                              // (the contraction has moved volumes and evidences since the last update)
        struct MirrorEnds { ClusterMirror &m; ~MirrorEnds() { m.valid = false; } } mirror_ends{cmir};
        const int nold = h_ctl->ncluster;
        if (c_desc_cap < nold) { dfree(c_desc); dfree(c_bout); c_desc_cap = std::max(2 * nold, 64); c_desc = dalloc<int>((size_t)4 * c_desc_cap); c_bout = dalloc<int>(c_desc_cap); }
        if ((int)cn.size() != nold) cn = dl(S.cl_n, (size_t)nold);
        std::vector<int> desc, verdict(nold, 1);
        std::vector<std::vector<int>> final_labels((size_t)nold); std::vector<int> final_num((size_t)nold, 1);
        bool refined = false;
        {
            int o1 = 0; long long o2 = 0;
            std::vector<int> which;
            for (int c = 0; c < nold; ++c)
                if (cn[c] > 2) { desc.push_back(c); desc.push_back(cn[c]); desc.push_back((int)o2); desc.push_back(o1); which.push_back(c); o1 += cn[c]; o2 += (long long)cn[c] * cn[c]; }
            const int nd = (int)which.size();
            static const bool batch_off = std::getenv("PC_CLUSTER_BATCH_OFF") != nullptr;
            if (nd > 0 && !batch_off && o2 <= (long long)c_cap * c_cap) {
                send_pre(c_desc, desc.data(), sizeof(int) * desc.size());
                int nmax1 = 0;
                for (int k = 0; k < nd; ++k) nmax1 = std::max(nmax1, desc[4 * k + 1]);
                // (in step with other runs: the first pass of all runs that update in this round in three launches)
                if (co) co->rec(CK_CLUS1, S, {c_desc, c_Sm, c_knn, c_lab, c_bout}, {}, {0, nd, nmax1});
                else if (pc_launch_knn_cluster_batch(&S, desc.data(), c_desc, nd, c_Sm, c_knn, c_lab, c_bout, st)) engine_fail(PC_RC_LDS, "a cluster too large for the LDS kNN sort");
                std::vector<int> out, lab0;
                fetch(out, (const int *)c_bout, (size_t)nd);
                fetch(lab0, (const int *)c_lab, (size_t)o1);       // (the first pass' labels of every cluster: a few KB, the same wait)
                cmir_ask();                                        // (and what a split will read, should the pass find one)
                fetch_wait();
                cmir.valid = true;
                for (int k = 0; k < nd; ++k) verdict[which[k]] = out[k];
                refined = refine_partitions(desc, which, out, lab0, final_labels, final_num);
            } else for (int c = 0; c < nold; ++c) verdict[c] = cn[c] > 2 ? 2 : 1;      // (no first pass: look at every cluster)
        }
        int ic = 0;
        std::vector<int> cmap((size_t)nold);             // where the update's cluster j is in the list now; -1: it was split
        for (int j = 0; j < nold; ++j) cmap[(size_t)j] = j;
        auto split_at = [&](int p) { for (int &m : cmap) { if (m == p) m = -1; else if (m > p) m -= 1; } };
        for (int j = 0; j < nold; ++j) {                 // j: the cluster's number when the update began; ic: its number now
            if (ic >= h_ctl->ncluster) break;
            const int n = cn[j];
            if (refined && n > 2 && verdict[j] > 1) {
                // (the recursion of clustering.f90:80-95 was made for all clusters and all runs level by level: refine_partitions)
                if (final_num[j] > 1) { found = true; add_cluster(ic, final_labels[j], final_num[j]); split_at(ic); }
                else ic++;
            } else if (n > 2 && verdict[j] > 1) {
                direct_op();
                HIPCHK(hipMemcpyAsync(c_pts, S.cl_list + (size_t)ic * S.Ncap, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
                pc_launch_similarity(&S, c_pts, n, c_Sm, st);
                std::vector<int> gidx(n), labels;
                for (int i = 0; i < n; ++i) gidx[i] = i;
                const int num = nn_clustering(n, gidx, labels);
                if (num > 1) { found = true; add_cluster(ic, labels, num); split_at(ic); }
                else ic++;
            } else ic++;
        }
        if (found) {
            if (cfg.epoch_discard) h_ctl->admin_epoch++;         // nested_sampling.F90:331-333 as written: every chain in flight is lost
            else if (h_ctl->i_nursery > 0) {
                // the engine's rule: the chains seeded in clusters this update left alone stay in the nursery, under their new numbers
                if (c_map_cap < nold) { dfree(c_map); c_map_cap = std::max(2 * nold, 64); c_map = dalloc<int>(c_map_cap); }
                send_raw(c_map, cmap.data(), sizeof(int) * (size_t)nold);
                direct_op();
                pc_launch_remap_chains(&S, c_map, nold, h_ctl->i_nursery, st);
            }
            h_ctl->status = PC_ST_RUNNING;
            send_raw(S.ctl, h_ctl, sizeof(PcCtl));
        }
        return found;
    }

    void covmats(int nph, int nc)
    {
        const size_t nchunk = pc_cov_nchunk(&S, nph);
        if (nchunk * nc > cov_chunks_cap) {
            dfree(psum); dfree(pcnt); dfree(pcov); dfree(mean); dfree(count);
            cov_chunks_cap = nchunk * nc * 2;
            psum = dalloc<double>(cov_chunks_cap * S.D); pcnt = dalloc<int>(cov_chunks_cap);
            pcov = dalloc<double>(cov_chunks_cap * S.D * S.D);
            mean = dalloc<double>((size_t)S.maxc * S.D); count = dalloc<int>(S.maxc);
        }
        direct_op();
        if (pc_launch_covmats(&S, nph, nc, psum, pcnt, mean, count, pcov, st)) {
            engine_fail(PC_RC_LDS, "covariance tile exceeds LDS (nDims too large)");
        }
    }

    // prior + likelihood on the host for one hypercube point (calculate.f90:40-41)
    double host_eval(const double *cube, double *theta, double *phi)
    {
        const int D = S.D;
        if (cb_prior) cb_prior(const_cast<double *>(cube), theta, D);
        else {
            if ((int)h_lo.size() != D) {
                h_lo.assign(D, 0.0); h_hi.assign(D, 1.0);
                if (d_lo) { HIPCHK(hipMemcpy(h_lo.data(), d_lo, sizeof(double) * D, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(h_hi.data(), d_hi, sizeof(double) * D, hipMemcpyDeviceToHost)); }
            }
            for (int d = 0; d < D; ++d) theta[d] = h_lo[d] + (h_hi[d] - h_lo[d]) * cube[d];
        }
        cb_evals++;
        return cb_like(theta, D, phi, S.nDer);
    }

    // prior + likelihood for n hypercube points at once: one call of the registered batch callback, else n scalar calls
    void host_eval_batch(int n, const double *cubes, double *thetas, double *phis, double *logLs)
    {
        const int D = S.D, nDer = S.nDer, pd = std::max(1, nDer);
        if (batch_fn) { batch_fn(batch_user, n, D, nDer, cubes, thetas, phis, logLs); cb_evals += n; return; }
        for (int i = 0; i < n; ++i) logLs[i] = host_eval(cubes + (size_t)i * D, thetas + (size_t)i * D, phis + (size_t)i * pd);
    }

    void generate_live_callback()
    {   // GenerateLivePoints with host evaluations; same Philox streams as k_generate_live
        const int nprior = cfg.nprior <= 0 ? cfg.nlive : cfg.nprior, nT = S.nT, D = S.D;
        std::vector<double> rows((size_t)nprior * nT, 0.0);
        int have = 0; uint32_t attempt = 0; long long nlike = 0;
        double t_eval = 0.0;
        const int pd = std::max(1, S.nDer);
        std::vector<double> bc, bt, bp, bl;
        while (have < nprior) {
            // as many attempts as points are still missing (all of them valid is the common case); attempt numbers,
            // and with them the Philox streams, are those of one-at-a-time generation
            const int m = nprior - have;
            bc.resize((size_t)m * D); bt.resize((size_t)m * D); bp.assign((size_t)m * pd, 0.0); bl.resize(m);
            for (int i = 0; i < m; ++i)
                for (int d = 0; d < D; ++d) bc[(size_t)i * D + d] = h_uniform(S.k0, S.k1, PC_DOM_LIVEGEN, 0u, attempt + (uint32_t)i, (uint32_t)d);
            const auto te0 = std::chrono::steady_clock::now();
            host_eval_batch(m, bc.data(), bt.data(), bp.data(), bl.data());
            t_eval += std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count();
            if (stop.load(std::memory_order_relaxed)) return;
            for (int i = 0; i < m; ++i) {
                if (!(bl[i] > cfg.logzero)) continue;
                double *row = rows.data() + (size_t)have * nT;
                std::copy(bc.begin() + (size_t)i * D, bc.begin() + (size_t)(i + 1) * D, row);
                std::copy(bt.begin() + (size_t)i * D, bt.begin() + (size_t)(i + 1) * D, row + S.p0);
                std::copy(bp.begin() + (size_t)i * pd, bp.begin() + (size_t)i * pd + S.nDer, row + S.d0);
                row[S.b0] = cfg.logzero; row[S.l0] = bl[i];
                have++; nlike++;
            }
            attempt += (uint32_t)m;
            if (attempt > 1000u * (uint32_t)nprior + 100000u) engine_fail(PC_RC_SETTINGS, "could not generate live points (likelihood is logzero everywhere?)");
        }
        double *drows = dalloc<double>((size_t)nprior * nT);
        HIPCHK(hipMemcpy(drows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
        pc_launch_install_live(&S, drows, nprior, st);
        h_ctl->nlike = nlike;
        HIPCHK(hipMemcpyAsync(&S.ctl->nlike, &h_ctl->nlike, sizeof(long long), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        dfree(drows);
        ndiscarded = (long)attempt - nprior;
        cb_eval_seconds = attempt ? t_eval / attempt : -1.0;
        call_dumper(2);
        if (nprior > cfg.nlive) { path[PCHIP_PATH_CONSUME_GENERAL]++; pc_launch_consume(&S, 2, 0, st); read_ctl(); }
    }

    // one nursery batch in callback mode: tick the chains until all of them are done
    void slice_callback(unsigned batch)
    {
        const int D = S.D, nDer = S.nDer;
        int first = 1;
        while (true) {
            // answers of the host: [logL (B) | theta (B x D) | phi (B x nDerived)], device copy at d_ans (B = chains per nursery now)
            pc_launch_slice_tick(&S, batch, B, d_cs, d_x0s, d_decks, d_prop, d_ans, d_ans + B, d_ans + B + (size_t)B * D, first, hp_prop, hp_need, st);
            first = 0; cb_ticks++;
            HIPCHK(hipStreamSynchronize(st));
            if (stop.load(std::memory_order_relaxed)) return;
            double *evL = hp_ans, *evT = hp_ans + B, *evP = evT + (size_t)B * D;
            const int pd = std::max(1, nDer);
            int nneed = 0;
            if (batch_fn) {                                // the parked proposals of this round in one call
                cb_idx.clear();
                for (int c = 0; c < B; ++c) if (hp_need[c]) cb_idx.push_back(c);
                nneed = (int)cb_idx.size();
                if (nneed > 0) {
                    cb_c.resize((size_t)nneed * D); cb_t.resize((size_t)nneed * D); cb_p.assign((size_t)nneed * pd, 0.0); cb_l.resize(nneed);
                    for (int i = 0; i < nneed; ++i) std::copy(hp_prop + (size_t)cb_idx[i] * D, hp_prop + (size_t)(cb_idx[i] + 1) * D, cb_c.begin() + (size_t)i * D);
                    host_eval_batch(nneed, cb_c.data(), cb_t.data(), cb_p.data(), cb_l.data());
                    for (int i = 0; i < nneed; ++i) {
                        const int c = cb_idx[i];
                        evL[c] = cb_l[i];
                        std::copy(cb_t.begin() + (size_t)i * D, cb_t.begin() + (size_t)(i + 1) * D, evT + (size_t)c * D);
                        std::copy(cb_p.begin() + (size_t)i * pd, cb_p.begin() + (size_t)(i + 1) * pd, evP + (size_t)c * pd);
                    }
                    if (stop.load(std::memory_order_relaxed)) return;
                }
            } else
            for (int c = 0; c < B; ++c)
                if (hp_need[c]) { nneed++; evL[c] = host_eval(hp_prop + (size_t)c * D, evT + (size_t)c * D, evP + (size_t)c * pd); }
            if (nneed == 0) break;
            HIPCHK(hipMemcpyAsync(d_ans, hp_ans, sizeof(double) * (size_t)B * (1 + D + std::max(1, nDer)), hipMemcpyHostToDevice, st));
        }
    }

    void generate_live()
    {   // GenerateLivePoints (generate.F90:150-183): keep the first nprior valid prior samples
        const int nprior = cfg.nprior <= 0 ? cfg.nlive : cfg.nprior, nT = S.nT;
        double *rows = dalloc<double>((size_t)nprior * nT), *rl = dalloc<double>(nprior);
        std::vector<double> keep_rows; keep_rows.reserve((size_t)nprior * nT);
        std::vector<double> hrows, hl(nprior);        // (hrows: only when a prior sample was not valid -- 736 KB zeroed for nothing was a third of a run's set-up)
        int have = 0, attempt0 = 0, last_attempt = -1;
        long long nlike = 0;
        bool direct = true;
        while (have < nprior) {
            if (pc_launch_generate_live(&S, attempt0, nprior, rows, rl, st)) engine_fail(PC_RC_NDIMS, "nDims > 256 unsupported");
            HIPCHK(hipMemcpyAsync(hl.data(), rl, sizeof(double) * nprior, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            int nvalid = 0;
            for (int i = 0; i < nprior; ++i) nvalid += hl[i] > cfg.logzero;
            if (nvalid == nprior && have == 0) { have = nprior; nlike = nprior; last_attempt = nprior - 1; break; }   // common case: all valid
            direct = false;
            hrows.resize((size_t)nprior * nT);
            HIPCHK(hipMemcpy(hrows.data(), rows, sizeof(double) * (size_t)nprior * nT, hipMemcpyDeviceToHost));
            for (int i = 0; i < nprior && have < nprior; ++i)
                if (hl[i] > cfg.logzero) { keep_rows.insert(keep_rows.end(), hrows.begin() + (size_t)i * nT, hrows.begin() + (size_t)(i + 1) * nT); have++; nlike++; last_attempt = attempt0 + i; }
                else ndiscarded++;
            attempt0 += nprior;
        }
        if (!direct) HIPCHK(hipMemcpy(rows, keep_rows.data(), sizeof(double) * (size_t)nprior * nT, hipMemcpyHostToDevice));
        pc_launch_install_live(&S, rows, nprior, st);
        h_ctl->nlike = nlike; h_ctl->nlike_device = nlike;
        HIPCHK(hipMemcpyAsync(&S.ctl->nlike, &h_ctl->nlike, sizeof(long long), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (S.seq_mode) {
            // the reference draws and evaluates one more prior sample while it times the likelihood
            // (time_speeds, generate.F90:388-393); the stream position after it is where the sampling starts.
            // With explicit repeats for every grade it does not (generate.F90:285-287).
            int a = last_attempt + 1;
            for (; S.ngrade <= 1; ++a) {
                double l1 = 0.0;
                (void)pc_launch_generate_live(&S, a, 1, rows, rl, st);
                HIPCHK(hipMemcpyAsync(&l1, rl, sizeof(double), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (l1 > cfg.logzero) break;
            }
            h_ctl->seq = (unsigned long long)(a + (S.ngrade <= 1 ? 1 : 0)) * S.D;
            HIPCHK(hipMemcpy(&S.ctl->seq, &h_ctl->seq, sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
        dfree(rows); dfree(rl);
        call_dumper(2);                // write_prior_file, nested_sampling.F90:197
        if (nprior > cfg.nlive) {      // nested_sampling.F90:201-205
            path[PCHIP_PATH_CONSUME_GENERAL]++; pc_launch_consume(&S, 2, 0, st);
            read_ctl();
        }
    }

    // Dead rows are append-only: everything below ctl.ndead is final once the round's kernels have been
    // synchronised, so it can travel to the pinned result buffer on a second stream during the run.
    void stream_dead()
    {
        const size_t nd = (size_t)h_ctl->ndead;
        if (!h_dead || nd > h_dead_cap || nd <= h_dead_copied) return;
        if (ev_apply) HIPCHK(hipStreamWaitEvent(st_copy, ev_apply, 0));     // the rows are written by k_apply_dead_ph, which may still run
        HIPCHK(hipMemcpyAsync(h_dead + h_dead_copied * S.nT, S.dead + h_dead_copied * S.nT,
                              sizeof(double) * (nd - h_dead_copied) * S.nT, hipMemcpyDeviceToHost, st_copy));
        h_dead_copied = nd;
    }

    // ---- .resume (pc_resume.h): the sampler state at an update boundary, in the reference's own grammar
    void export_resume(PcResume &r)
    {
        HIPCHK(hipStreamSynchronize(st));
        const int D = S.D, nT = S.nT, nc = h_ctl->ncluster, ncd = std::min(h_ctl->ncluster_dead, S.maxc_dead), maxc = S.maxc;
        r = PcResume{};
        r.nDims = D; r.nDerived = S.nDer; r.ndead = h_ctl->ndead; r.ncluster = nc; r.ncluster_dead = ncd;
        {
            long ng[PC_MAX_GRADE];
            grade_counts(ng);
            r.grade_dims.clear(); r.num_repeats.clear(); r.nlike.clear();
            for (int g = 0; g < S.ngrade; ++g) {
                r.grade_dims.push_back((g + 1 < S.ngrade ? S.g_off[g + 1] : D) - S.g_off[g]);
                r.num_repeats.push_back(S.g_nr[g]); r.nlike.push_back(ng[g]);
            }
        }
        r.logZ = h_ctl->logZ; r.logZ2 = h_ctl->logZ2; r.logX_last_update = h_ctl->logX_last_update;
        r.thin_posterior = cfg.boost_posterior < 0.0 ? 1.0 : cfg.boost_posterior / (double)S.nr;      // generate.F90:311-316
        auto take = [&](const double *p) { auto v = dl(p, std::max(1, nc)); v.resize(nc); return v; };
        r.logLp = take(S.logLp); r.logXp = take(S.logXp); r.logZXp = take(S.logZXp); r.logZp = take(S.logZp);
        r.logZp2 = take(S.logZp2); r.logZpXp = take(S.logZpXp);
        auto xq = dl(S.XpXq, (size_t)maxc * maxc);
        r.logXpXq.assign((size_t)nc * nc, 0.0);
        for (int q = 0; q < nc; ++q) for (int p = 0; p < nc; ++p) r.logXpXq[(size_t)q * nc + p] = xq[(size_t)p * maxc + q];
        r.logZp_dead = dl(S.logZp_dead, std::max(1, ncd)); r.logZp_dead.resize(ncd);
        r.logZp2_dead = dl(S.logZp2_dead, std::max(1, ncd)); r.logZp2_dead.resize(ncd);
        auto cov = dl(S.cov, (size_t)std::max(1, nc) * D * D), ch = dl(S.chol, (size_t)std::max(1, nc) * D * D);
        r.covmat.assign((size_t)nc * D * D, 0.0); r.cholesky.assign((size_t)nc * D * D, 0.0);
        for (int c = 0; c < nc; ++c) for (int j = 0; j < D; ++j) for (int a = 0; a < D; ++a) {     // file line j = column j
            r.covmat[((size_t)c * D + j) * D + a] = cov[((size_t)c * D + a) * D + j];
            r.cholesky[((size_t)c * D + j) * D + a] = ch[((size_t)c * D + a) * D + j];
        }
        auto cn = dl(S.cl_n, std::max(1, nc)); auto cl = dl(S.cl_list, (size_t)maxc * S.Ncap);
        auto rows = dl(S.live, (size_t)S.Ncap * nT); auto uid = dl(S.cl_uid, std::max(1, nc));
        r.nlive.assign(nc, 0); r.imin.assign(nc, 1); r.live.assign(nc, {}); r.nphantom.assign(nc, 0); r.phantom.assign(nc, {});
        for (int c = 0; c < nc; ++c) {
            r.nlive[c] = cn[c];
            double lo = PC_HUGE;
            for (int k = 0; k < cn[c]; ++k) {
                const double *row = rows.data() + (size_t)cl[(size_t)c * S.Ncap + k] * nT;
                r.live[c].insert(r.live[c].end(), row, row + nT);
                if (row[S.l0] < lo) { lo = row[S.l0]; r.imin[c] = k + 1; }
            }
        }
        const int nph = h_ctl->nphantom;
        auto ph = dl(S.phantom, (size_t)std::max(1, nph) * nT); auto pc = dl(S.ph_cuid, std::max(1, nph));
        for (int j = 0; j < nph; ++j)
            for (int c = 0; c < nc; ++c)
                if (uid[c] == pc[j]) { r.phantom[c].insert(r.phantom[c].end(), ph.begin() + (size_t)j * nT, ph.begin() + (size_t)(j + 1) * nT); r.nphantom[c]++; break; }
        r.dead = dl(S.dead, (size_t)std::max(1, r.ndead) * nT); r.dead.resize((size_t)r.ndead * nT);
        r.logweights = dl(S.dead_logw, std::max(1, r.ndead)); r.logweights.resize(r.ndead);
    }

    // The reference's .resume grammar has no place for what this engine derives its posterior files from: the cluster
    // every dead point died in (a stable id), the genealogy of the splits with the evidence fractions, the phantoms kept
    // by boost_posterior.  They travel in a sidecar next to the file, <path>.hip (binary, this engine only); a run that
    // resumes from a .resume without it -- one the reference wrote -- has the evidences and the global posterior of
    // the dead points, and per-cluster posteriors from the resume point on.
    void write_sidecar(const std::string &path, int ndead, int nc, int ncd)
    {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) engine_fail(PC_RC_RESUME, "cannot write %s", path.c_str());
        auto put = [&](const void *p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) { std::fclose(f); engine_fail(PC_RC_RESUME, "short write to %s", path.c_str()); } };
        const unsigned magic = 0x50434831u;   // "PCH1"
        const int np = S.D + S.nDer + 2, nsplit = (int)split_child.size(), npp = (int)pp_logpost.size();
        const unsigned next_uid = h_ctl->next_cluster_uid;
        auto dc = dl(S.dead_cuid, (size_t)std::max(1, ndead)); auto ua = dl(S.cl_uid, (size_t)std::max(1, nc)); auto ud = dl(S.cl_uid_dead, (size_t)std::max(1, ncd));
        put(&magic, 4); put(&ndead, 4); put(&nc, 4); put(&ncd, 4); put(&next_uid, 4); put(&nsplit, 4); put(&npp, 4); put(&np, 4);
        put(dc.data(), sizeof(unsigned) * ndead); put(ua.data(), sizeof(unsigned) * nc); put(ud.data(), sizeof(unsigned) * ncd);
        put(split_child.data(), sizeof(unsigned) * nsplit); put(split_parent.data(), sizeof(unsigned) * nsplit); put(split_logfrac.data(), sizeof(double) * nsplit);
        put(pp_rows.data(), sizeof(double) * (size_t)npp * np); put(pp_logpost.data(), sizeof(double) * npp); put(pp_cuid.data(), sizeof(unsigned) * npp);
        std::fclose(f);
    }
    struct Sidecar { bool ok = false; unsigned next_uid = 1; std::vector<unsigned> dead_cuid, uid, uid_dead; };
    Sidecar read_sidecar(const std::string &path, int ndead, int nc, int ncd)
    {
        Sidecar sc;
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return sc;
        auto get = [&](void *p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; };
        unsigned magic = 0; int nd = 0, c = 0, cd = 0, nsplit = 0, npp = 0, np = 0;
        bool good = get(&magic, 4) && magic == 0x50434831u && get(&nd, 4) && get(&c, 4) && get(&cd, 4) && get(&sc.next_uid, 4) &&
                    get(&nsplit, 4) && get(&npp, 4) && get(&np, 4) && nd == ndead && c == nc && cd == ncd && np == S.D + S.nDer + 2 &&
                    nsplit >= 0 && npp >= 0;
        if (good) {
            sc.dead_cuid.resize(nd); sc.uid.resize(c); sc.uid_dead.resize(cd);
            split_child.resize(nsplit); split_parent.resize(nsplit); split_logfrac.resize(nsplit);
            pp_rows.resize((size_t)npp * np); pp_logpost.resize(npp); pp_cuid.resize(npp);
            good = get(sc.dead_cuid.data(), sizeof(unsigned) * nd) && get(sc.uid.data(), sizeof(unsigned) * c) && get(sc.uid_dead.data(), sizeof(unsigned) * cd) &&
                   get(split_child.data(), sizeof(unsigned) * nsplit) && get(split_parent.data(), sizeof(unsigned) * nsplit) &&
                   get(split_logfrac.data(), sizeof(double) * nsplit) && get(pp_rows.data(), sizeof(double) * (size_t)npp * np) &&
                   get(pp_logpost.data(), sizeof(double) * npp) && get(pp_cuid.data(), sizeof(unsigned) * npp);
        }
        std::fclose(f);
        if (!good) { split_child.clear(); split_parent.clear(); split_logfrac.clear(); pp_rows.clear(); pp_logpost.clear(); pp_cuid.clear(); }
        // (the kept phantoms' ids do not travel in the sidecar: after a resume the earlier ones are numbered -- ids only key the equal-weight trials)
        pp_uid.resize(pp_logpost.size()); for (size_t k = 0; k < pp_uid.size(); ++k) pp_uid[k] = (1ull << 62) | k;
        sc.ok = good;
        return sc;
    }

    void write_resume()
    {
        if (!cfg.resume_write) return;
        PcResume r;
        export_resume(r);
        std::string err;
        if (!pc_resume_write(cfg.resume_write, r, cfg.logzero, err)) engine_fail(PC_RC_RESUME, "%s", err.c_str());
        write_sidecar(std::string(cfg.resume_write) + ".hip", r.ndead, r.ncluster, r.ncluster_dead);
    }

    // upload a .resume state; the run continues with the counter RNG streams of batch `ndead` onwards
    // (the reference does not store its generator state either: a resumed run is a valid, different trajectory)
    bool import_resume(const PcResume &r, std::string &err)
    {
        const int D = S.D, nT = S.nT, nc = r.ncluster, Ncap = S.Ncap;
        if (r.nDims != D || r.nDerived != S.nDer) { err = "resume file has different nDims / nDerived"; return false; }
        if (nc > S.maxc) grow_clusters(nc);
        if (r.ncluster_dead + nc + 8 > S.maxc_dead) grow_dead_clusters(2 * (r.ncluster_dead + nc + 8));
        const int maxc = S.maxc;
        int ntot = 0, nph = 0;
        for (int c = 0; c < nc; ++c) { ntot += r.nlive[c]; nph += r.nphantom[c]; }
        // ncluster = 0: the file of a finished run (every live point killed): nothing left to sample, the run
        // returns what the file holds, as the reference does
        if (ntot > Ncap || (nc >= 1 && ntot < 1)) { err = "resume file: live point counts do not fit this run's settings"; return false; }
        if (nph + (long long)B * S.nr > S.Pcap) { h_ctl->nphantom = 0; grow_phantoms(nph + (long long)B * S.nr); }
        if ((long long)r.ndead + B + Ncap + 16 > S.Dcap) { h_ctl->ndead = 0; grow_dead(2 * (r.ndead + B + Ncap + 16)); }
        std::vector<double> rows((size_t)Ncap * nT, 0.0), lL(Ncap, PC_HUGE), entry(Ncap, cfg.logzero);
        std::vector<int> lc(Ncap, -1), lp(Ncap, 0), cl((size_t)maxc * Ncap, 0), cn(maxc, 0), imin(maxc, 0);
        std::vector<unsigned> uid(maxc, 0u);
        const Sidecar sc = cfg.resume_read ? read_sidecar(std::string(cfg.resume_read) + ".hip", r.ndead, nc, std::min(r.ncluster_dead, S.maxc_dead)) : Sidecar{};
        std::vector<double> logLp(maxc, cfg.logzero), lref(maxc, 0.0), lsum(maxc, 0.0);
        int slot = 0;
        for (int c = 0; c < nc; ++c) {
            cn[c] = r.nlive[c]; uid[c] = sc.ok ? sc.uid[c] : (unsigned)(c + 1);
            double lo = PC_HUGE, hi = -PC_HUGE;
            for (int k = 0; k < r.nlive[c]; ++k, ++slot) {
                const double *row = r.live[c].data() + (size_t)k * nT;
                std::memcpy(rows.data() + (size_t)slot * nT, row, sizeof(double) * nT);
                lL[slot] = row[S.l0]; lc[slot] = c; lp[slot] = k; entry[slot] = row[S.b0]; cl[(size_t)c * Ncap + k] = slot;
                if (row[S.l0] < lo) { lo = row[S.l0]; imin[c] = slot; }
                hi = std::max(hi, row[S.l0]);
            }
            logLp[c] = r.nlive[c] ? lo : cfg.logzero;
            lref[c] = r.nlive[c] ? hi : 0.0;
            for (int k = 0; k < r.nlive[c]; ++k) lsum[c] += std::exp(r.live[c][(size_t)k * nT + S.l0] - lref[c]);
        }
        ul(S.live, rows); ul(S.live_logL, lL); ul(S.live_entry, entry); ul(S.live_cluster, lc); ul(S.live_pos, lp);
        { std::vector<int> none(Ncap, -1); ul(S.slot_src, none); }     // every live row is current (k_install_live does this in a fresh run)
        ul(S.cl_list, cl); ul(S.cl_n, cn); ul(S.cl_uid, uid); ul(S.imin_slot, imin); ul(S.logLp, logLp); ul(S.lse_ref, lref); ul(S.lse_sum, lsum);
        auto put = [&](double *dst, const std::vector<double> &v) { std::vector<double> t(maxc, cfg.logzero); std::copy(v.begin(), v.end(), t.begin()); ul(dst, t); };
        put(S.logZp, r.logZp); put(S.logZXp, r.logZXp); put(S.logZp2, r.logZp2); put(S.logZpXp, r.logZpXp);
        { std::vector<double> t(maxc, 0.0); std::copy(r.logXp.begin(), r.logXp.end(), t.begin()); ul(S.logXp, t); }
        std::vector<double> xq((size_t)maxc * maxc, 0.0);
        for (int q = 0; q < nc; ++q) for (int p = 0; p < nc; ++p) xq[(size_t)p * maxc + q] = r.logXpXq[(size_t)q * nc + p];
        ul(S.XpXq, xq);
        std::vector<double> cov((size_t)maxc * D * D, 0.0), ch((size_t)maxc * D * D, 0.0);
        for (int c = 0; c < maxc; ++c) for (int a = 0; a < D; ++a) { cov[((size_t)c * D + a) * D + a] = 1.0; ch[((size_t)c * D + a) * D + a] = 1.0; }
        for (int c = 0; c < nc; ++c) for (int j = 0; j < D; ++j) for (int a = 0; a < D; ++a) {
            cov[((size_t)c * D + a) * D + j] = r.covmat[((size_t)c * D + j) * D + a];
            ch[((size_t)c * D + a) * D + j] = r.cholesky[((size_t)c * D + j) * D + a];
        }
        ul(S.cov, cov); ul(S.chol, ch);
        const int ncd = std::min(r.ncluster_dead, S.maxc_dead);
        { std::vector<double> a(r.logZp_dead.begin(), r.logZp_dead.begin() + ncd), b(r.logZp2_dead.begin(), r.logZp2_dead.begin() + ncd); ul(S.logZp_dead, a); ul(S.logZp2_dead, b); }
        if (sc.ok) ul(S.cl_uid_dead, sc.uid_dead);
        // phantoms, cluster by cluster
        std::vector<double> ph((size_t)std::max(1, nph) * nT), phL(std::max(1, nph));
        std::vector<unsigned> phC(std::max(1, nph)); std::vector<unsigned long long> phU(std::max(1, nph));
        int j = 0;
        for (int c = 0; c < nc; ++c)
            for (int k = 0; k < r.nphantom[c]; ++k, ++j) {
                std::memcpy(ph.data() + (size_t)j * nT, r.phantom[c].data() + (size_t)k * nT, sizeof(double) * nT);
                phL[j] = ph[(size_t)j * nT + S.l0]; phC[j] = uid[c]; phU[j] = 0xFFFFFFFF00000000ull | (unsigned)j;
            }
        if (nph) { ul(S.phantom, ph); ul(S.ph_logL, phL); ul(S.ph_cuid, phC); ul(S.ph_uid, phU); }
        // dead points
        if (r.ndead) {
            ul(S.dead, r.dead); ul(S.dead_logw, r.logweights);
            std::vector<double> z(r.ndead, 0.0), en(r.ndead);
            for (int i = 0; i < r.ndead; ++i) en[i] = r.dead[(size_t)i * nT + S.b0];
            ul(S.dead_postX, z); ul(S.dead_postZ, z); ul(S.dead_entry, en);
            std::vector<unsigned> du(r.ndead, 0u);
            if (sc.ok) du = sc.dead_cuid;
            ul(S.dead_cuid, du);
        }
        PcCtl c0 = *h_ctl;
        c0.status = nc >= 1 ? PC_ST_RUNNING : PC_ST_DONE; c0.ncluster = nc; c0.ncluster_dead = ncd; c0.ndead = r.ndead; c0.nphantom = nph;
        c0.logZ = r.logZ; c0.logZ2 = r.logZ2; c0.logX_last_update = r.logX_last_update; c0.next_cluster_uid = sc.ok ? sc.next_uid : (unsigned)nc + 1;
        c0.nlike = 0;                                  // the engine's counter is the total over the grades
        for (size_t g = 0; g < r.nlike.size(); ++g) { c0.nlike += r.nlike[g]; if (g >= 1 && g < PC_MAX_GRADE) nlike_g[g] = r.nlike[g]; }
        c0.nlike_device = c0.nlike; c0.i_nursery = 0; c0.failures = 0;
        HIPCHK(hipMemcpy(S.ctl, &c0, sizeof(PcCtl), hipMemcpyHostToDevice));
        *h_ctl = c0;
        return true;
    }

    // ---- a run in phases, so that ONE host thread can keep several runs of a device in flight (pchip_run_repeats): begin(),
    //      then round_enqueue() / round_ready() / round_finish() until it says stop, then end().  run() is the same sequence
    //      for a single run, spinning in between.  State that lives across the phases:
    struct ActiveRun {
        int d; std::atomic<int> *flag;
        ActiveRun(int dv, std::atomic<int> *f) : d(dv & 63), flag(f) { g_active_runs.fetch_add(1); g_active_dev[d].fetch_add(1); std::lock_guard<std::mutex> g(g_run_mutex); g_run_stop.push_back(flag); }
        ~ActiveRun() { g_active_runs.fetch_sub(1); g_active_dev[d].fetch_sub(1); std::lock_guard<std::mutex> g(g_run_mutex); g_run_stop.erase(std::find(g_run_stop.begin(), g_run_stop.end(), flag)); }
    };
    std::unique_ptr<ActiveRun> active_run;
    using clk = std::chrono::steady_clock;
    clk::time_point r_t0, r_t1, r_t2;
    unsigned r_batch = 0; bool r_sort_valid = false, r_fresh = false, r_par_ok = false, r_static_ok = false; int r_nursery_left = 0;
    int r_rc = 0;                                 // outcome of the loop: 0, or the code run() returns

    int run(pchip_result *out)
    {
        int rc = begin();
        if (rc >= 0) return rc;
        while (true) {
            if (!round_enqueue()) break;
            while (!round_ready()) __builtin_ia32_pause();
            if (!round_finish()) break;
        }
        if (r_rc != 0) return r_rc;
        return end(out);
    }

    // everything before the first round; -1: go on, else the code to return
    int begin()
    {
        active_run.reset(new ActiveRun(dev, &stop));
        r_t0 = clk::now();
        h_dead_cap = (size_t)S.Dcap; h_dead = halloc<double>(h_dead_cap * S.nT); h_dead_copied = 0;
        bool resumed = false;
        if (cfg.resume_read) {                         // read_write.F90:384-476; a missing file means a fresh start
            if (FILE *probe = std::fopen(cfg.resume_read, "r")) {
                std::fclose(probe);
                PcResume rs; std::string err;
                if (!pc_resume_read(cfg.resume_read, rs, err) || !import_resume(rs, err)) { std::fprintf(stderr, "polychord_hip: %s\n", err.c_str()); return 6; }
                resumed = true;
                int ntot = 0;
                for (int v : rs.nlive) ntot += v;
                if (ntot > cfg.nlive && rs.ncluster == 1) { path[PCHIP_PATH_CONSUME_GENERAL]++; pc_launch_consume(&S, 2, 0, st); read_ctl(); ntot = cfg.nlive; }   // nested_sampling.F90:201-205
                resume_static = (ntot == cfg.nlive);
                resume_batch0 = (unsigned)rs.ndead;
            }
        }
        if (!resumed) { if (callback_mode) generate_live_callback(); else generate_live(); }
        if (stop.load(std::memory_order_relaxed)) return 5;
        if (cb_auto_batch && !(cb_eval_seconds >= 0.0 && cb_eval_seconds < 2e-6)) { B = B_small; S.B = B; }   // expensive (or unmeasured) callback
        r_t1 = clk::now();
        r_batch = resume_batch0;                      // fresh counter-RNG streams after a resume
        r_sort_valid = false;
        r_nursery_left = 0;
        const int nprior0 = cfg.nprior <= 0 ? cfg.nlive : cfg.nprior;
        // the one-cluster kernels assume a static number of live points; each has its own LDS budget
        const bool static_ok = (cfg.n_nlives == 0) && (nprior0 >= cfg.nlive) && resume_static && cfg.force_general != 1;
        fast_ok = static_ok && pc_fast_fits(&S);
        const bool par_ok = static_ok && cfg.force_general == 0 && pc_par_fits(&S);
        r_static_ok = static_ok; r_par_ok = par_ok;
        // The parallel contraction may run past an update trigger and have the update made afterwards, for the state at
        // the trigger (pc_update.hip): a nursery is then consumed in ONE launch instead of being cut where the reference
        // updates.  Only when nothing on the host is tied to the moment of an update (files, dumper, resume) and the
        // fused update applies.
        static const bool defer_off = std::getenv("PC_DEFER_OFF") != nullptr;
        // (ablate bits 1, 2, 3: no pool mode, no deferred update, no fused update -- the same numbers by other kernels: tests/)
        S.defer_update = (!defer_off && !(cfg.ablate & 4) && par_ok && !cfg.do_clustering && cfg.boost_posterior == 0.0 && !dumper && !on_update && !cfg.resume_write &&
                          !S.seq_mode && pc_update_fused_ok(&S, 1) && !std::getenv("PC_UPDATE_FUSED_OFF") && !(cfg.ablate & 8)) ? 1 : 0;
        // Pool mode (same conditions, likelihood on the device): k_slice writes a nursery's babies into the phantom array itself,
        // updates invalidate phantoms where they lie, and the array is compacted only when it is full -- the rows of a run
        // are written once and read once (pc_state.h).  The host keeps the cursor: nothing it does not know moves it.
        static const bool pool_off = std::getenv("PC_POOL_OFF") != nullptr;
        S.pool = (S.defer_update && !callback_mode && !pool_off && !(cfg.ablate & 2)) ? 1 : 0;
        if (S.pool) {
            pool_cursor = h_ctl->nphantom;
            babies_own = S.babies;
            if (g_cap_phantoms.load() <= 0) {                                    // room for several updates between compactions,
                size_t free_b = 0, total_b = 0;                                  // where the device has it to spare
                (void)hipMemGetInfo(&free_b, &total_b);
                const double need = 2.0 * 2.0 * (double)S.Pcap * ((double)S.nT * 8.0 + 24.0);       // both buffers at twice the rows
                if (need < 0.4 * (double)free_b) grow_phantoms(2LL * S.Pcap);
            }
        }
        r_rc = 0;
        return -1;
    }

    // The sampling of one nursery: pool rows, the bases (drawn ahead on the side stream, or now), k_slice, the bases of the
    // nurseries to come.  spec: enqueued BEHIND the previous nursery's contraction before the host has seen its outcome; the
    // kernel starts by asking the device whether that nursery was consumed whole with neither an update nor the end of the
    // run in its way (PcCtl::spec_ok, left by k_consume_par) and returns at once if not.
    // what the cohort's launches for any device likelihood take (else the run launches for itself in between)
    bool cohort_general_ok() const
    {
        static const bool off = std::getenv("PC_COHORT_GENERAL") && std::atoi(std::getenv("PC_COHORT_GENERAL")) == 0;
        return !off && S.ngrade <= 1 && !S.seq_mode && S.like.kind != PC_LIKE_CORR_GAUSSIAN && S.like.kind != PC_LIKE_CALLBACK;
    }
    bool enqueue_nursery(bool spec)
    {
        unsigned &batch = r_batch; int &nursery_left = r_nursery_left;
        if (S.pool) {
            if (pool_cursor + (long long)B * S.nr > S.Pcap) pool_compact();
            S.pool_base = (int)pool_cursor; S.pool_rows = B * S.nr; S.babies = S.phantom + (size_t)pool_cursor * S.nT;
            pool_cursor += (long long)B * S.nr;
        }
        {
            // (between the stamp of the last round and the launch of k_slice the device idles: nothing that can wait
            //  is done in between -- capacity checks precede the contraction, not the sampling)
            hipEvent_t e0 = spec ? nullptr : kt.begin(KT_NHATS);
            // (a run that has the chip to itself: next to other runs the side stream takes from them what it gives)
            // (next to other runs of this device the bases are drawn in line, in front of the sampling kernel: their side streams
            //  would take from each other what they give -- but the split itself, and with it the fused sampling kernel, stays)
            const bool multi = co != nullptr || g_active_dev[dev & 63].load(std::memory_order_relaxed) > 1;
            // (nDims 25 ... 64: the halves for a run on its own; runs in step take the whole kernel with the run in the grid, CK_NHATS_G)
            const bool splittable = pc_nhats_splittable(&S) != 0 && raw_buf[1] && !(S.D > 24 && S.D <= 64 && multi);
            const bool split = splittable && !multi;
            bool fused_slice = false;
            int bases_seq = 0;                        // in step with other runs: the number of the launch that drew this nursery's bases (0: in line)
            if (splittable) {
                // the bases of this nursery were drawn on the side stream while earlier ones were sampled and consumed (or
                // are drawn now)
                RawSlot &rs = ring[batch % raw_depth];
                S.nhat_raw = raw_buf[batch % raw_depth];
                if (rs.valid && rs.batch == batch && rs.B == B) { if (!rs.waited) HIPCHK(hipStreamWaitEvent(st, rs.ready, 0)); bases_seq = rs.co_seq; }      // (in step with other runs: the wait for the launch that drew them, Cohort::flush)
                else {
                    if (rs.valid) HIPCHK(hipStreamWaitEvent(st, rs.ready, 0));       // (a stale job may still be writing there)
                    if (co && pc_bases_t_ok(&S)) co->rec(CK_BASES, S, {}, {(long long)B}, {(int)batch});
                    else (void)pc_launch_nhats_part(&S, batch, B, 1, st, (multi || (cfg.ablate & 128)) ? 1 : 0);
                }
                rs.valid = false;
                fused_slice = !callback_mode && pc_slice_fusable(&S) != 0;       // seeds + whitening inside k_slice
                if (!fused_slice) { if (co) { co->flush(); co->wait_next(); } (void)pc_launch_nhats_part(&S, batch, B, 2, st, 0); }
            }
            else if (co && !callback_mode && cohort_general_ok() && S.D >= 25 && S.D <= 64) co->rec(CK_NHATS_G, S, {}, {(long long)B}, {(int)batch});
            else if ((co ? (co->flush(), 0) : 0) || pc_launch_nhats(&S, batch, B, st)) { std::fprintf(stderr, "polychord_hip: nDims unsupported\n"); r_rc = 3; return false; }
            kt.end(KT_NHATS, e0);
            hipEvent_t e1 = spec ? nullptr : kt.begin(KT_SLICE);
            S.spec_guard = spec ? 1 : 0;                                         // (the kernel looks at the contraction's verdict first)
            if (callback_mode) { if (co) co->flush(); slice_callback(batch); if (stop.load(std::memory_order_relaxed)) { r_rc = 5; return false; } }
            // next to other runs of this device (or settings.ablate bit 6): the lane = chain kernel (pc_slice_t.hip), the same
            // numbers from 1/60 of the wavefronts
            else if (fused_slice && !spec && (multi || (cfg.ablate & 64)) && pc_slice_t_ok(&S, h_ctl->ncluster)) {
                path[PCHIP_PATH_SLICE_LANE]++;
                if (co) co->rec(CK_SLICE, S, {}, {(long long)B}, {(int)batch, 0, 0, bases_seq}); else (void)pc_launch_slice_t(&S, batch, B, st);
                // in step with other runs: the bases of the next nurseries on the runs' second stream, next to this round's kernels
                if (co && co->st2 && splittable && raw_depth >= 2 && pc_bases_t_ok(&S)) bases_ahead(batch);
            }
            else if (co && !callback_mode && !spec && cohort_general_ok() && (fused_slice || !splittable)) {
                // in step with other runs, any device likelihood / several clusters: the one-run kernel with the run in the grid
                path[PCHIP_PATH_SLICE_WAVE]++;
                co->rec(CK_SLICE_G, S, {}, {(long long)B, fused_slice ? 1LL : 0LL}, {(int)batch, 0, 0, fused_slice ? bases_seq : 0});
                if (fused_slice && co->st2 && raw_depth >= 2 && pc_bases_t_ok(&S)) bases_ahead(batch);
            }
            else if ((path[PCHIP_PATH_SLICE_WAVE]++, co ? (co->flush(), co->wait_next(), 0) : 0) || (fused_slice ? pc_launch_slice_fused(&S, batch, B, st) : pc_launch_slice(&S, batch, B, st))) { std::fprintf(stderr, "polychord_hip: nDims unsupported\n"); r_rc = 3; return false; }
            S.spec_guard = 0;
            kt.end(KT_SLICE, e1);
            if (split && !spec) side_prefetch(batch);      // (speculative: only once the device is known to have taken the nursery)
            if (S.ngrade > 1) HIPCHK(hipMemcpyAsync(h_nlike_g.data(), S.ch_nlike_g, sizeof(int) * h_nlike_g.size(), hipMemcpyDeviceToHost, st));
            batch++; tm.batches++;
            S.nn_valid = 0; nursery_left = B;
        }
        return true;
    }

    // in step with other runs: the bases of the nurseries after `cur` that are not drawn yet (two ahead with a ring of three) written down for
    // the cohort's second stream
    void bases_ahead(unsigned cur)
    {
        // (two ahead -- PC_COHORT_AHEAD=2 -- takes the wait for the bases out of a round without an update, but the bases kernels then run
        //  next to k_slice_t: in a process of the engine's own sixteen runs take as long as with one ahead and 32 / 64 runs 2 % less, in
        //  bench.py's process (the HIP runtime PyTorch brings) 8-10 % MORE: one ahead)
        static const unsigned ahead_env = std::getenv("PC_COHORT_AHEAD") ? (unsigned)std::max(1, std::min(2, std::atoi(std::getenv("PC_COHORT_AHEAD")))) : 1u;
        const unsigned ahead = raw_depth >= 3 ? ahead_env : 1u;
        for (unsigned x = cur + 1; x <= cur + ahead; ++x) {
            RawSlot &rn = ring[x % raw_depth];
            if (rn.valid && rn.batch == x && rn.B == B) continue;
            PcState S1 = S; S1.nhat_raw = raw_buf[x % raw_depth];
            co->rec(CK_BASES_NEXT, S1, {}, {(long long)B}, {(int)x});
            rn.valid = true; rn.batch = x; rn.B = B; rn.waited = true; rn.co_seq = co->seq_for_next();
        }
    }

    void ensure_side()
    {
        if (st_side) return;
        st_side = stream_beside({st, st_copy}); ev_main = hpool().get_sync_event();
        for (int r = 0; r < raw_depth; ++r) { ring[r].ready = hpool().get_sync_event(); ring[r].consumed = hpool().get_sync_event(); }
    }
    // Several clusters, a run on its own: the sorted order of the live set (k_sort_live: one workgroup, 17-26 us + a launch) is what the
    // candidate lists and the one-wave contraction start from, and it depends on the live set only -- which the sampling of a fresh
    // nursery does not touch.  It is made on the side stream while k_slice runs; the main stream waits for it in front of the lists.
    hipEvent_t ev_presort_a = nullptr, ev_presort_b = nullptr;
    bool presorted = false;
    void presort_live()
    {
        static const bool off = std::getenv("PC_PRESORT_OFF") != nullptr;
        presorted = false;
        if (off || co || callback_mode || S.seq_mode || !S.nn_list || h_ctl->ncluster < 2 || g_active_dev[dev & 63].load(std::memory_order_relaxed) != 1) return;
        ensure_side();
        if (!ev_presort_a) { ev_presort_a = hpool().get_sync_event(); ev_presort_b = hpool().get_sync_event(); }
        HIPCHK(hipEventRecord(ev_presort_a, st)); HIPCHK(hipStreamWaitEvent(st_side, ev_presort_a, 0));      // (behind the row copies of the round before)
        if (pc_launch_sort_live(&S, st_side) != 0) return;
        HIPCHK(hipEventRecord(ev_presort_b, st_side));
        presorted = true;
    }
    // the bases of the nurseries after `cur`, on the side stream (behind cur's sampling kernel on the main stream)
    void side_prefetch(unsigned cur)
    {
        const unsigned batch = cur;
        {
                // drawn while the one-CU contraction of this nursery runs: next to k_slice (one wave per SIMD) the
        // 2000 workgroups of the bases kernel cost it 10 us, next to the contraction nothing
        ensure_side();
        // (nDims > 64: the bases take longer than the contraction and the slice kernel is one wave per SIMD for
        //  half a millisecond: there they run next to it from the start)
        static const bool side_free_env = std::getenv("PC_SIDE_FREE") != nullptr, side_ord_env = std::getenv("PC_SIDE_ORDERED") != nullptr;
        const bool side_free = side_free_env || (S.D > 64 && !side_ord_env);
        // the buffer of this nursery is free again once its bases have been whitened (fused: once sampled)
        // (ordered: the side stream follows k_slice anyway -- one event between k_slice and the contraction, not two:
        //  every record on the main stream is a few microseconds before the next kernel starts)
        RawSlot &cur = ring[batch % raw_depth];
        if (side_free) { HIPCHK(hipEventRecord(cur.consumed, st)); cur.used = true; }
        if (!side_free) { HIPCHK(hipEventRecord(ev_main, st)); HIPCHK(hipStreamWaitEvent(st_side, ev_main, 0)); }
        // ordered: the next nursery only; free: as far ahead as there are buffers (the last one is this nursery's own)
        const unsigned xmax = batch + (unsigned)raw_depth - (side_free ? 0u : 1u);
        for (unsigned x = batch + 1; x <= xmax; ++x) {
            RawSlot &rs = ring[x % raw_depth];
            if (rs.valid && rs.batch == x && rs.B == B) continue;
            if (rs.valid) continue;                                          // (a job for another nursery size: used or replaced when its turn comes)
            if (rs.used) HIPCHK(hipStreamWaitEvent(st_side, rs.consumed, 0));
            PcState S1 = S; S1.nhat_raw = raw_buf[x % raw_depth];
            hipEvent_t es = kt.begin_on(KT_SIDE, st_side);
            (void)pc_launch_nhats_part(&S1, x, B, 1, st_side, (cfg.ablate & 128) ? 1 : 0);
            kt.end_on(KT_SIDE, es, st_side);
            HIPCHK(hipEventRecord(rs.ready, st_side));
            rs.valid = true; rs.batch = x; rs.B = B; rs.waited = false;
        }
    }
    }

    // a speculative nursery the device declined: the host's bookkeeping of it is taken back (its bases stay where they are,
    // drawn and waited for: the real launch finds them)
    bool spec_pending = false, spec_hit = false;
    long spec_tried = 0, spec_declined = 0;
    int spec_upd_in = 0; double spec_lived = 0.0;      // lived deaths until the next update after the round just seen; lived deaths a nursery yields (estimate)
    void spec_undo()
    {
        r_batch--; tm.batches--;
        if (S.pool) pool_cursor -= (long long)B * S.nr;
        RawSlot &rs = ring[r_batch % raw_depth];
        rs.valid = true; rs.batch = r_batch; rs.B = B; rs.waited = true;
    }

    // enqueue one round (sampling when the nursery is empty, contraction, row copies); false: the loop is over (r_rc says how)
    bool round_enqueue()
    {
        unsigned &batch = r_batch; bool &sort_valid = r_sort_valid; int &nursery_left = r_nursery_left;
        const bool par_ok = r_par_ok, static_ok = r_static_ok; const int wide = 0;
        {
            if (h_ctl->status == PC_ST_DONE) { if (spec_pending) { spec_pending = false; spec_undo(); } return false; }
            if (h_ctl->status == PC_ST_ERROR) { std::fprintf(stderr, "polychord_hip: device error %d\n", h_ctl->error); r_rc = 2; return false; }
            bool fresh_nursery = false;
            if (h_ctl->i_nursery == 0) {
                fresh_nursery = true;
                const bool have = spec_pending;                 // (still pending here = the device took it: round_finish undid the others)
                spec_pending = false;
                if (have) side_prefetch(batch - 1);
                else { const auto n0 = std::chrono::steady_clock::now(); if (r_static_ok && cfg.force_general == 0 && !(cfg.ablate & 32)) presort_live(); const bool okn = enqueue_nursery(false); g_dbg_nursery_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - n0).count(); if (!okn) return false; }
            }
            if (fresh_nursery) { const auto n0 = std::chrono::steady_clock::now(); ensure_capacity(); g_dbg_capacity_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - n0).count(); }
            hipEvent_t e2 = kt.begin(KT_CONSUME);
            int rc2;
            const bool use_fast = fast_ok && h_ctl->ncluster == 1;
            launch_stamp();
            if (presorted && (co || !(h_ctl->ncluster > 1) || use_fast)) { HIPCHK(hipStreamWaitEvent(st, ev_presort_b, 0)); presorted = false; }
            if (par_ok && h_ctl->ncluster == 1) {
                // the parallel contraction keeps the sorted order of the live set up to date itself
                rc2 = 0; S.nn_valid = 0;                     // (the one-cluster kernels do not keep the list bookkeeping)
                path[PCHIP_PATH_CONSUME_PAR]++;
                if (co) { if (!sort_valid) { co->rec(CK_SORT, S, {}, {}, {}); sort_valid = true; } co->rec(CK_CONSUME, S, {}, {}, {}); }
                else {
                if (!sort_valid) { rc2 = pc_launch_sort_live(&S, st); sort_valid = true; }
                rc2 = rc2 || pc_launch_consume_par(&S, st);      // also lays out the phantoms
                }
            }
            else if (use_fast) { if (co) co->flush(); sort_valid = false; S.nn_valid = 0; path[PCHIP_PATH_CONSUME_FAST]++; rc2 = pc_launch_consume_fast(&S, 0, st); pc_launch_ph_prepare(&S, st); }
            else {
                sort_valid = false;
                // several clusters: rank the possible nearest neighbours of every baby still in the nursery once, on
                // the whole chip; the serial contraction then walks short lists instead of searching the live set
                static const bool nn_off = std::getenv("PC_NN_LISTS_OFF") != nullptr;
                const bool want_nn = h_ctl->ncluster > 1 && S.nn_list && !S.nn_valid && !nn_off && !S.seq_mode && nursery_left > 1;
                // several clusters, static number of live points, lists in place: the one-wave contraction (pc_clus.hip); everything
                // else -- and every launch when settings.ablate bit 5 is set -- goes to the general kernel, which is its arbiter
                static const bool cl_off = std::getenv("PC_CONSUME_CL_OFF") != nullptr;
                const bool use_cl = static_ok && cfg.force_general == 0 && !cl_off && !(cfg.ablate & 32) && (S.nn_valid || want_nn) && !S.seq_mode && h_ctl->ncluster > 1 &&
                                    pc_consume_cl_fits(&S, h_ctl->ncluster);
                if (co && use_cl && cohort_general_ok()) {
                    // in step with other runs: lists, sort and the one-wave contraction once for all runs with several clusters
                    if (want_nn) { co->rec(CK_NN, S, {}, {}, {0, nursery_left}); S.nn_valid = 1; path[PCHIP_PATH_NN_LISTS]++; }
                    path[pc_consume_clp_fits(&S, h_ctl->ncluster) ? PCHIP_PATH_CONSUME_CL : PCHIP_PATH_CONSUME_CL_SERIAL]++;
                    co->rec(CK_SORT, S, {}, {}, {});
                    co->rec(CK_CONSUME_CL, S, {}, {h_ctl->ncluster > 64 ? 1LL : 0LL}, {});
                    rc2 = 0;
                } else {
                if (co) co->flush();
                bool sorted_now = false;
                if (presorted) { HIPCHK(hipStreamWaitEvent(st, ev_presort_b, 0)); sorted_now = fresh_nursery; presorted = false; }      // (made beside k_slice: presort_live)
                if (want_nn) {
                    // (the sorted order first: its ranks tell the lists' kernel which candidates cannot die before a chain is looked at)
                    if (!sorted_now) sorted_now = pc_launch_sort_live(&S, st) == 0;
                    pc_launch_nn_lists(&S, nursery_left, sorted_now ? 1 : 0, st);
                    S.nn_valid = 1; path[PCHIP_PATH_NN_LISTS]++;
                }
                path[use_cl ? (pc_consume_clp_fits(&S, h_ctl->ncluster) ? PCHIP_PATH_CONSUME_CL : PCHIP_PATH_CONSUME_CL_SERIAL) : PCHIP_PATH_CONSUME_GENERAL]++;
                if (use_cl) {
                    rc2 = (sorted_now ? 0 : pc_launch_sort_live(&S, st)) || pc_launch_consume_cl(&S, h_ctl->ncluster, st);
                } else
                rc2 = pc_launch_consume(&S, 0, (h_ctl->ncluster > 1) ? 1 : wide, st);
                }
            }
            if (rc2) { std::fprintf(stderr, "polychord_hip: nlive too large for the LDS-resident contraction\n"); r_rc = 4; return false; }
            kt.end(KT_CONSUME, e2);
            hipEvent_t e3 = kt.begin(KT_APPLY);
            if (co) co->rec(CK_APPLY, S, {}, {(long long)B}, {(int)(batch - 1)}); else pc_launch_apply(&S, batch - 1, B, st);
            kt.end(KT_APPLY, e3);
            // the main stream's wait for the next nursery's bases is enqueued now, behind this round's kernels (long
            // satisfied when the next k_slice gets there), not between the stamp and the next launch
            if (st_side) { RawSlot &rs = ring[batch % raw_depth]; if (rs.valid && rs.batch == batch && !rs.waited) { HIPCHK(hipStreamWaitEvent(st, rs.ready, 0)); rs.waited = true; } }
            r_fresh = fresh_nursery;
            ready_spins = 0;
            // PC_SPEC=1 (experiment, off by default): the next nursery's sampling enqueued behind this round's kernels before the
            // host knows how the round ends; k_slice asks the device first (PcCtl::spec_ok) and returns at once when an update or
            // the end of the run is in the way.  48 of the 79 rounds of the metric configuration end with an empty nursery and no
            // update, and in those the device goes from the row copies straight into k_slice -- but the run is no shorter for it
            // (14.75 ms against 14.65, A/B in one call, identical results): the 31 declined launches and the second trip through
            // the launch path cost what the 48 saved host round trips give.
            // Round 4: the launch is enqueued ahead only when the update is far enough away.  The contraction reports how many deaths
            // that enter the live set are left until the next trigger (PcCtl::upd_in, as the state stood after the LAST launch); a nursery
            // yields at most spec_lived of them (a running estimate from the rounds seen so far, B before any was seen), so the sampling
            // of the nursery after this one is enqueued now if this nursery AND a tenth more cannot reach the trigger.  A wrong guess
            // costs a declined launch, never a result: the device's guard decides.  PC_SPEC=0: never; PC_SPEC=1: always (the experiment).
            // Measured (round 4, metric configuration): 40 of 79 nurseries enqueued ahead, none declined -- and the run is no shorter
            // (12.81 against 12.83 ms, three A/B pairs): the 13-18 us between the row copies and the next k_slice are the queue's, not the
            // host's.  So: off unless asked for (PC_SPEC=2 = by the estimate).
            static const int spec_mode = std::getenv("PC_SPEC") ? std::atoi(std::getenv("PC_SPEC")) : 0;
            const bool spec_off = spec_mode == 0 || (spec_mode == 2 && !(spec_upd_in > 0 && (double)spec_upd_in > 1.1 * spec_lived + 8.0));
            if (!spec_off && S.pool && par_ok && h_ctl->ncluster == 1 && !callback_mode && st_side && pc_slice_fusable(&S) != 0 &&
                g_active_dev[dev & 63].load(std::memory_order_relaxed) == 1 &&
                pool_cursor + (long long)B * S.nr <= S.Pcap && (long long)h_ctl->ndead + 2LL * B + S.Ncap + 16 <= S.Dcap) {
                RawSlot &rs = ring[batch % raw_depth];
                if (rs.valid && rs.batch == batch && rs.B == B && rs.waited) {
                    if (!enqueue_nursery(true)) return false;
                    spec_pending = true; spec_hit = false; spec_tried++;
                }
            }
        }
        return true;
    }

    // will round_finish wait for the device (an update that is not merely written down)?
    bool finish_may_wait() const
    {
        const bool upd = h_ctl->status == PC_ST_UPDATE || (h_ctl->upd_pending && h_ctl->status == PC_ST_RUNNING);
        return upd && (cfg.do_clustering || dumper || on_update || cfg.resume_write || cfg.boost_posterior != 0.0 || S.seq_mode || h_ctl->ncluster > 1 ||
                       !pc_update_fused_ok(&S, h_ctl->ncluster));
    }
    // what the round did, once its stamp is in; false: the loop is over
    bool round_finish()
    {
        unsigned &batch = r_batch; int &nursery_left = r_nursery_left; const bool fresh_nursery = r_fresh;
        (void)batch;
        if (spec_pending) {
            spec_hit = h_ctl->status == PC_ST_RUNNING && !h_ctl->upd_pending && h_ctl->i_nursery == 0 && h_ctl->error == 0;   // what the device's guard saw
            if (!spec_hit) { spec_pending = false; spec_undo(); spec_declined++; }      // (before the update looks at the pool's cursor)
        }
        {
            // A run whose last death exhausts a nursery AND triggers an update learns that it is over only from the next
            // launch (the kernels test more_samples_needed before a death, nested_sampling.F90:237): the nursery
            // generated in between was never touched and does not count.
            if (fresh_nursery && h_ctl->status == PC_ST_DONE && h_ctl->i_nursery == B) tm.batches--;
            nursery_left = h_ctl->i_nursery;
            {   // what the contraction said about the next update (round_enqueue decides by it whether to enqueue ahead)
                const int now = h_ctl->upd_in;
                const bool updated = h_ctl->status == PC_ST_UPDATE || h_ctl->upd_pending;
                if (spec_lived <= 0.0) spec_lived = (double)B;
                if (!updated && fresh_nursery && h_ctl->i_nursery == 0 && spec_upd_in > now && now > 0) {
                    const double lived = (double)(spec_upd_in - now);
                    spec_lived = spec_lived >= (double)B ? lived : 0.75 * spec_lived + 0.25 * lived;
                }
                spec_upd_in = now;
            }
            tally_grades();
            tm.rounds++;
            // dead rows leave for the host at every update (a copy per round, ~350 KB, next to the one-CU contraction cost it
            // 6 us per launch: 72 against 66 us)
            if (h_ctl->status == PC_ST_UPDATE || (h_ctl->upd_pending && h_ctl->status == PC_ST_RUNNING)) {
                do_update(h_ctl->status != PC_ST_UPDATE); h_ctl->status = PC_ST_RUNNING; h_ctl->upd_pending = 0;
                // (every few updates: in step with other runs a copy request costs the one thread that drives them all ~10 us, and a run on its own
                //  finds the k_slice next to a copy as much longer as the copy lasts -- four copies of a run's thirty-one: 11.40 -> 11.33 ms, three A/B pairs)
                if ((size_t)h_ctl->ndead >= h_dead_copied + 4 * (size_t)cfg.nlive) {
                if (!ev_apply) ev_apply = hpool().get_sync_event();
                HIPCHK(hipEventRecord(ev_apply, st));       // the dead rows of the rounds so far are in place behind this point
                stream_dead();
                }
            }
        }
        return true;
    }

    // kill-off, results; the code pchip_run returns
    // ---- the end of a run in two halves, so that runs in step can end together: end_a asks the device for everything (kill-off,
    //      moments, copies), end_b -- once the stream has been waited for -- makes the results.  end() = both, for a run on its own.
    struct EndState {
        double *hlive = nullptr; int *hcl = nullptr; double *d_pmax = nullptr, *h_part = nullptr, *h_zp = nullptr;
        int nc_end = 0, ncd_max = 0, pmD = 0, pm_nb = 0, pm_pw = 0; clk::time_point t2;
    } es;
    // A run in step that ends with several clusters: its kill-off is the general contraction kernel, one workgroup for ~4 ms
    // (BASELINE configs[2]: a thousand deaths one after the other), and on the cohort's stream the other runs' next round waited
    // behind it -- sixteen endings 70 ms of a 670 ms call.  It goes to the run's copy stream, behind everything the run has had.
    hipStream_t st_fin = nullptr; hipEvent_t ev_fin = nullptr;
    void end_a(bool fused_final = false)
    {
        if (co && !fused_final) co->flush();      // (fused: the caller has launched what was pending, and launches the kill-offs together)
        const bool par_ok = r_par_ok; bool &sort_valid = r_sort_valid;
        es.t2 = clk::now();
        st_fin = st;
        static const bool fin_off = std::getenv("PC_COHORT_FINAL_ASIDE") && std::atoi(std::getenv("PC_COHORT_FINAL_ASIDE")) == 0;
        if (co && fused_final && !fin_off && st_copy && st_copy != st && h_ctl->ncluster > 1) {
            ev_fin = hpool().get_sync_event();
            HIPCHK(hipEventRecord(ev_fin, st));
            HIPCHK(hipStreamWaitEvent(st_copy, ev_fin, 0));
            st_fin = st_copy;
        }
        hipStream_t st = st_fin;                  // (what follows is this run's alone)
        // snapshot of the live set at termination, then nested_sampling.F90:381-384
        const int nT = S.nT;
        // (pinned buffers, copies in stream order in front of the kill-off: the host does not stop here)
        es.hlive = halloc<double>((size_t)S.Ncap * nT); es.hcl = halloc<int>(S.Ncap);
        HIPCHK(hipMemcpyAsync(es.hlive, S.live, sizeof(double) * (size_t)S.Ncap * nT, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(es.hcl, S.live_cluster, sizeof(int) * S.Ncap, hipMemcpyDeviceToHost, st));
        es.nc_end = h_ctl->ncluster;
        if (h_ctl->ncluster == 0) {
            // a finished run read back from its .resume file: nothing to kill
        } else if (par_ok && h_ctl->ncluster == 1) {
            path[PCHIP_PATH_KILLOFF_PAR]++;
            if (co && sort_valid) { co->rec(CK_FINAL, S, {}, {}, {}); if (!fused_final) co->flush(); }      // (fused: the caller launches the kill-off of all runs that end now, then calls end_a2)
            else {
            if (!sort_valid) (void)pc_launch_sort_live(&S, st);
            (void)pc_launch_final_par(&S, st);
            }
        } else if (h_ctl->ncluster > 1 && pc_launch_killoff_cl(&S, h_ctl->ncluster, st) == 0) {      // (several clusters: the deaths in sorted order by one wavefront, pc_clus.hip)
            path[PCHIP_PATH_KILLOFF_CL]++;
        } else if (fast_ok && h_ctl->ncluster == 1 && pc_launch_consume_fast(&S, 1, st) == 0) path[PCHIP_PATH_KILLOFF_FAST]++;
        else { path[PCHIP_PATH_KILLOFF_GENERAL]++; pc_launch_consume(&S, 1, (h_ctl->ncluster > 1 && !S.seq_mode) ? 1 : 0, st); }   // (the general kernel: four waves, a death's jobs side by side)
        if (!fused_final) end_a2();
    }
    void end_a2()
    {
        hipStream_t st = st_fin ? st_fin : this->st;
        // what the results need from the device is requested here, behind the kill-off and before the host waits for it: the
        // posterior moments of theta over the dead points (device reduction, fixed order; the kernels take the count from the
        // control block) and the evidences of the retired clusters -- one wait instead of four
        es.pmD = S.D + S.nDer; es.pm_nb = pc_post_blocks(); es.pm_pw = 2 * es.pmD + 1;   // theta and phi columns are contiguous in a row
        es.ncd_max = std::min(h_ctl->ncluster_dead + h_ctl->ncluster, S.maxc_dead);
        // (the partial sums go straight into pinned host memory, which the device addresses like its own: 180 KB over the
        //  link instead of a D2H copy request behind the kernel -- that request stalled its caller for 5-6 ms once per process,
        //  in the second or third run)
        es.d_pmax = dalloc<double>(es.pm_nb); es.h_part = halloc<double>((size_t)es.pm_nb * es.pm_pw);
        es.h_zp = halloc<double>(2 * (size_t)std::max(1, es.ncd_max));
        pc_launch_post_moments(&S, -1, es.d_pmax, es.h_part, st);
        if (es.ncd_max > 0) {
            HIPCHK(hipMemcpyAsync(es.h_zp, S.logZp_dead, sizeof(double) * es.ncd_max, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(es.h_zp + es.ncd_max, S.logZp2_dead, sizeof(double) * es.ncd_max, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(h_ctl, S.ctl, sizeof(PcCtl), hipMemcpyDeviceToHost, st));       // (read_ctl without its wait: end_b's caller waits)
        if (ev_fin) HIPCHK(hipEventRecord(ev_fin, st));      // (a kill-off beside the cohort's stream: end_b's caller waits for this one too)
    }
    void end_wait_aside() { if (ev_fin) { HIPCHK(hipEventSynchronize(ev_fin)); hpool().put_sync_event(ev_fin); ev_fin = nullptr; } }
    int end(pchip_result *out) { end_a(); HIPCHK(hipStreamSynchronize(st)); end_wait_aside(); return end_b(out); }
    int end_b(pchip_result *out)
    {
        struct HostBuf { EndState &e; ~HostBuf() { if (e.hlive) hfree(e.hlive); if (e.hcl) hfree(e.hcl); e.hlive = nullptr; e.hcl = nullptr; } } hb{es};
        double *hlive = es.hlive; int *hcl = es.hcl; double *d_pmax = es.d_pmax, *h_part = es.h_part, *h_zp = es.h_zp;
        const int nc_end = es.nc_end, ncd_max = es.ncd_max, pmD = es.pmD, pm_nb = es.pm_nb, pm_pw = es.pm_pw, nT = S.nT;
        const auto t0 = r_t0, t1 = r_t1; auto t2 = es.t2;
        HIPCHK(hipGetLastError());                    // a kernel that could not be launched must not go unnoticed
        nph_stale = false;
        kt.collect();
        if (cfg.boost_posterior != 0.0 && (cfg.posteriors || cfg.equals) && h_ctl->nphantom > 0) {
            // the last update_posteriors (nested_sampling.F90:386-390): every remaining phantom is below the last death
            pc_launch_clean(&S, h_ctl->nphantom, keep, blk, d_total, ph2, phL2, phC2, phU2, nullptr, st);
            collect_phantom_posteriors(h_ctl->nphantom);
        }
        call_dumper(1);
        write_resume();
        auto t3 = clk::now();
        tm.t_gen = std::chrono::duration<double>(t1 - t0).count();
        tm.t_loop = std::chrono::duration<double>(t2 - t1).count();
        tm.t_final = std::chrono::duration<double>(t3 - t2).count();
        tm.t_total = std::chrono::duration<double>(t3 - t0).count();
        // ---- results (calculate_logZ_estimate, run_time_info.f90:652-678)
        std::memset(out, 0, sizeof(*out));
        out->logZ = std::max(-PC_HUGE, 2 * h_ctl->logZ - 0.5 * h_ctl->logZ2);
        out->varlogZ = h_ctl->logZ2 - 2 * h_ctl->logZ;
        out->ndead = h_ctl->ndead; out->nlike = h_ctl->nlike; out->niter = h_ctl->niter;
        out->nlike_failed = h_ctl->nlike_failed; out->ncluster_peak = ncluster_peak; out->epoch_discard = S.epoch_discard;
        path[PCHIP_PATH_NN_FALLBACKS] = (long)h_ctl->nn_fallbacks; path[PCHIP_PATH_POOL_MODE] = S.pool; path[PCHIP_PATH_DEFER_UPDATE] = S.defer_update;
        for (int k = 0; k < PCHIP_PATH_COUNT; ++k) out->path[k] = path[k];
        grade_counts(out->nlike_grade);
        out->ncluster = nc_end; out->ncluster_dead = h_ctl->ncluster_dead; out->nbatches = tm.batches;
        out->nrounds = tm.rounds; out->nupdates = tm.updates; out->nTotal = nT; out->batch = B;
        out->t_generate = tm.t_gen; out->t_loop = tm.t_loop; out->t_final = tm.t_final; out->t_total = tm.t_total;
        for (int k = 0; k < KT_N; ++k) { out->k_time_s[k] = kt.total_ms[k] * 1e-3; out->k_launches[k] = kt.launches[k]; }
        // developer counters (PC_DEBUG=2|3|4); the feedback setting keeps the reference's meaning (feedback.f90)
        static const int dbg_lvl = std::getenv("PC_DEBUG") ? std::atoi(std::getenv("PC_DEBUG")) : 0;
        if (dbg_lvl == 6) std::fprintf(stderr, "polychord_hip dbg spec: %ld nurseries enqueued ahead, %ld of them declined by the device (%ld rounds); a nursery yields ~%.0f lived deaths\n", spec_tried, spec_declined, tm.rounds, spec_lived);
        if (dbg_lvl >= 3) std::fprintf(stderr, "polychord_hip dbg general: term %lld identify %lld kill+add %lld tail %lld cycles; %lld chains identified from the candidate lists, %lld of them fell back to the full search\n", h_ctl->gen_cyc[0], h_ctl->gen_cyc[1], h_ctl->gen_cyc[2], h_ctl->gen_cyc[3], h_ctl->nn_walks, h_ctl->nn_fallbacks);
        if (dbg_lvl == 4) std::fprintf(stderr, "polychord_hip dbg par: stage+search %lld rank-sort %lld accept %lld merge+slots %lld evidence %lld triggers %lld publish %lld cycles\n", h_ctl->dbg[0], h_ctl->dbg[1], h_ctl->dbg[2], h_ctl->dbg[3], h_ctl->dbg[4], h_ctl->dbg[5], h_ctl->dbg[6]);
        if (dbg_lvl == 4) std::fprintf(stderr, "polychord_hip dbg par: %lld evidence scans as pairs (terms beyond one scale)\n", h_ctl->dbg[7]);
        if (dbg_lvl == 4 && h_ctl->wave_cyc[0]) std::fprintf(stderr, "polychord_hip dbg clp (several clusters; the line above then reads: decisions | order of deaths + slots | counts + volumes | volume sum + step 0's phantoms | prefix sums + commit | waves 1-3 | write back | passes): wave 1 %lld wave 2 %lld wave 3 %lld phantom wave %lld cycles\n", h_ctl->wave_cyc[0], h_ctl->wave_cyc[1], h_ctl->wave_cyc[2], h_ctl->wave_cyc[3]);
        if (dbg_lvl == 2) std::fprintf(stderr, "polychord_hip dbg: loop cycles %lld passB %lld (%lld flushes) accept-steps %lld (%lld) ins-rescan %lld (%lld) reject-steps cycles %lld\n", h_ctl->dbg[0], h_ctl->dbg[1], h_ctl->dbg[4], h_ctl->dbg[2], h_ctl->dbg[3], h_ctl->dbg[5], h_ctl->dbg[6], h_ctl->dbg[7]);
        if ((size_t)h_ctl->ndead > h_dead_cap) {          // the dead array grew beyond the first estimate
            HIPCHK(hipStreamSynchronize(st_copy));
            hfree(h_dead); h_dead_cap = (size_t)h_ctl->ndead; h_dead = halloc<double>(h_dead_cap * nT); h_dead_copied = 0;
        }
        stream_dead();
        out->dead = h_dead; h_dead = nullptr;
        out->logweights = halloc<double>(std::max(1, h_ctl->ndead));
        HIPCHK(hipMemcpyAsync(out->logweights, S.dead_logw, sizeof(double) * h_ctl->ndead, hipMemcpyDeviceToHost, st_copy));
        out->entry = halloc<double>(std::max(1, h_ctl->ndead));
        HIPCHK(hipMemcpyAsync(out->entry, S.dead_entry, sizeof(double) * h_ctl->ndead, hipMemcpyDeviceToHost, st_copy));
        int nl = 0;
        for (int s = 0; s < S.Ncap; ++s) nl += hcl[s] >= 0;
        out->nlive_final = nl;
        out->live = (double *)std::malloc(sizeof(double) * (size_t)std::max(1, nl) * nT);
        out->live_cluster = (int *)std::malloc(sizeof(int) * (size_t)std::max(1, nl));
        for (int s = 0, k = 0; s < S.Ncap; ++s) if (hcl[s] >= 0) { std::memcpy(out->live + (size_t)k * nT, hlive + (size_t)s * nT, sizeof(double) * nT); out->live_cluster[k] = hcl[s]; k++; }
        const int ncd = std::min(h_ctl->ncluster_dead, S.maxc_dead);
        out->nZp = ncd;
        out->logZp = (double *)std::malloc(sizeof(double) * std::max(1, ncd));
        out->varlogZp = (double *)std::malloc(sizeof(double) * std::max(1, ncd));
        if (ncd > ncd_max) engine_fail(PC_RC_DEVICE, "more retired clusters (%d) than the final stage could have left (%d)", ncd, ncd_max);
        for (int i = 0; i < ncd; ++i) { const double zp = h_zp[i], zp2 = h_zp[ncd_max + i]; out->logZp[i] = 2 * zp - 0.5 * zp2; out->varlogZp[i] = zp2 - 2 * zp; }
        hfree(h_zp);
        // posterior moments of theta from the dead points (requested above)
        const int D = pmD, nb = pm_nb, pw = pm_pw;
        out->post_mean = (double *)std::calloc(D, sizeof(double)); out->post_var = (double *)std::calloc(D, sizeof(double));
        double sw = 0.0;
        for (int b = 0; b < nb; ++b) {
            sw += h_part[(size_t)b * pw + 2 * D];
            for (int d = 0; d < D; ++d) { out->post_mean[d] += h_part[(size_t)b * pw + d]; out->post_var[d] += h_part[(size_t)b * pw + D + d]; }
        }
        for (int d = 0; d < D; ++d) { out->post_mean[d] /= sw; out->post_var[d] = out->post_var[d] / sw - out->post_mean[d] * out->post_mean[d]; }
        dfree(d_pmax); hfree(h_part);
        // settings.device_records: the lived records picked here, on the device that made them (what the exchange step of repeat-sharded
        // runs sends); their count comes back with the wait below
        long long *h_nrec = nullptr;
        if (cfg.device_records && h_ctl->ndead > 0) {
            const long long cap = h_ctl->ndead;
            double *blk_rec = (double *)pc_cache_dev_alloc(pc_records_block_bytes(cap, nT));
            if (!blk_rec) engine_fail(PC_RC_MEMORY, "no device memory for the run's records (%lld rows)", cap);
            h_nrec = halloc<long long>(1);
            if (pc_pack_lived_device(S.dead, S.dead_logw, S.dead_entry, cap, nT, cfg.logzero, blk_rec, cap, h_nrec, st_copy) != 0) { pc_cache_dev_free(blk_rec); hfree(h_nrec); engine_fail(PC_RC_DEVICE, "packing the run's records"); }
            out->d_records = blk_rec; out->records_cap = (long)cap; out->records_device = dev;
        }
        HIPCHK(hipStreamSynchronize(st_copy));
        if (h_nrec) { out->n_records = (long)*h_nrec; hfree(h_nrec); }
        active_run.reset();
        return 0;
    }

    // (streams_idle: the caller has waited for everything the run's streams were given -- a run in step whose ending was waited
    //  for by event; six stream waits from each of eight threads at once were half of the teardown's time)
    void destroy(bool streams_idle = false)
    {
        active_run.reset();
        // work may still be in flight on any of the run's streams (early returns, the prefetched bases of a nursery
        // that was never consumed): the blocks below go back to a process-wide cache and may be handed to another
        // run's thread at once
        if (!streams_idle) {
        if (st) (void)hipStreamSynchronize(st);
        if (st_copy) (void)hipStreamSynchronize(st_copy);
        if (st_side) (void)hipStreamSynchronize(st_side);
        }
        const auto dq0 = std::chrono::steady_clock::now();
        std::vector<void *> blocks; blocks.reserve(160);
        struct Batch { std::vector<void *> &b; Batch(std::vector<void *> &v) : b(v) { tl_dfree_batch = &b; } ~Batch() { tl_dfree_batch = nullptr; dcache().put_many(b); b.clear(); } };
        {
        Batch batch_guard(blocks);
        if (raw_buf[0]) { S.nhat_raw = raw_buf[0]; for (int r = 1; r < RAW_RING; ++r) dfree(raw_buf[r]); raw_buf[0] = nullptr; }     // S.nhat_raw pointed at one of them
        if (babies_own) { S.babies = babies_own; babies_own = nullptr; }      // (pool mode pointed it into the phantom array)
        double **dd[] = { &S.live, &S.live_logL, &S.logZp, &S.logXp, &S.logZXp, &S.logZp2, &S.logZpXp, &S.logLp, &S.XpXq,
                          &S.lse_ref, &S.lse_sum, &S.death_thr, &S.chol, &S.cov, &S.logZp_dead, &S.logZp2_dead, &S.phantom,
                          &S.ph_logL, &S.dead, &S.dead_logw, &S.dead_postX, &S.dead_postZ, &S.babies, &S.baby_logL, &S.baby_logL_T,
                          &S.ch_contour, &S.nhat, &S.nhat_w, &S.nhat_raw, &S.nhat_Ms, &S.ch_My, &S.live_entry, &S.dead_entry, &ph2, &phL2, &psum, &mean,
                          &pcov, &d_lo, &d_hi, &d_invcovT, &d_mean, &d_dynL };
        for (auto p : dd) dfree(*p);
        int **ii[] = { &S.live_cluster, &S.live_pos, &S.cl_list, &S.cl_n, &S.imin_slot, &S.ch_cluster, &S.ch_epoch, &S.ch_nlike,
                       &S.ch_seed_slot, &S.slot_src, &S.slot_step, &S.slot_dead, &S.sort_slot, &blk, &d_total, &pcnt, &count, &d_dynN };
        for (auto p : ii) dfree(*p);
        { char *cs = (char *)d_cs; dfree(cs); d_cs = nullptr; }
        dfree(d_x0s); dfree(d_prop); dfree(d_ans); dfree(d_decks);
        if (hp_prop) { hfree(hp_prop); hp_prop = nullptr; } if (hp_ans) { hfree(hp_ans); hp_ans = nullptr; } if (hp_need) { hfree(hp_need); hp_need = nullptr; }
        dfree(upd_part); dfree(upd_shift); upd_part_cap = 0;
        dfree(c_gdesc); dfree(c_gpool); dfree(c_glab); dfree(c_gout); c_g_cap = 0; dfree(c_map); c_map_cap = 0; dfree(c_desc); dfree(c_bout); dfree(c_Sm); dfree(c_pts); dfree(c_gidx); dfree(c_knn); dfree(c_lab); dfree(c_out); dfree(c_cnt); dfree(c_olduid); c_cap = 0; c_desc_cap = 0;   // (the clustering scratch used to stay behind: 12 MB per clustered run)
        for (void *h : staged_up) hfree(h);
        staged_up.clear();
        for (const Fetch &f : fetching) hfree(f.h);
        fetching.clear();
        dfree(d_logn); dfree(S.ch_nlike_g); dfree(S.nn_list); dfree(S.nn_slot_owner); dfree(S.nn_chain_slot); dfree(S.nn_pts); dfree(S.nn_code);
        unsigned **uu[] = { &S.cl_uid, &S.ph_cuid, &S.dead_cuid, &phC2, &S.cl_uid_dead };
        for (auto p : uu) dfree(*p);
        dfree(S.ph_uid); dfree(S.sort_key); dfree(S.plan); dfree(phU2); dfree(keep); dfree(S.ctl);
        }
        const auto dq1 = std::chrono::steady_clock::now();
        kt.destroy();
        if (h_dead) { hfree(h_dead); h_dead = nullptr; }
        for (void *h : setup_staged) hfree(h);
        setup_staged.clear();
        if (h_ctl) hfree(h_ctl); h_ctl = nullptr;
        if (h_note) hfree((void *)h_note); h_note = nullptr;
        if (ev_apply) { hpool().put_sync_event(ev_apply); ev_apply = nullptr; }

        if (st) { if (!streams_idle) (void)hipStreamSynchronize(st); if (!co) hpool().put_stream(st); } st = nullptr;
        if (st_copy) { if (!streams_idle) (void)hipStreamSynchronize(st_copy); if (!st_copy_shared) hpool().put_stream(st_copy); } st_copy = nullptr;
        if (st_side) {
            if (!streams_idle) (void)hipStreamSynchronize(st_side); hpool().put_stream(st_side); hpool().put_sync_event(ev_main);
            if (ev_presort_a) { hpool().put_sync_event(ev_presort_a); hpool().put_sync_event(ev_presort_b); ev_presort_a = ev_presort_b = nullptr; }
            for (int r = 0; r < RAW_RING; ++r) { if (ring[r].ready) hpool().put_sync_event(ring[r].ready); if (ring[r].consumed) hpool().put_sync_event(ring[r].consumed); ring[r] = RawSlot(); }
        }
        st_side = nullptr; ev_main = nullptr;
        const auto dq2 = std::chrono::steady_clock::now();
        g_dbg_d1 += std::chrono::duration_cast<std::chrono::nanoseconds>(dq1 - dq0).count();
        g_dbg_d2 += std::chrono::duration_cast<std::chrono::nanoseconds>(dq2 - dq1).count();
    }
};

}  // namespace

extern "C" {

void polychord_hip_request_stop(void) { std::lock_guard<std::mutex> g(g_run_mutex); for (auto *f : g_run_stop) f->store(1); }
// tests of the error paths: the next run fails once at the chosen point (1: a device allocation, 2: the cluster
// capacity at the next split, 3: the phantom array at its next growth); 0 disarms
void pchip_inject_fault(int kind) { g_inject_fault = kind; }
// the block caches for the library's other files (the merge's scratch and its rows): null when there is no memory.  The merge
// took its scratch from the driver and gave it back at its end: the driver then works the frees off for 10 ... 30 ms, and the
// next call's first wait for the device -- if it came at once -- sat behind that
void *pc_cache_dev_alloc(size_t bytes) { try { return dcache().get(bytes); } catch (const EngineError &) { (void)hipGetLastError(); return nullptr; } }
void pc_cache_dev_free(void *p) { if (p && !dcache().put(p)) (void)hipFree(p); }
void *pc_cache_host_alloc(size_t bytes) { try { return hcache().get(bytes); } catch (const EngineError &) { (void)hipGetLastError(); return nullptr; } }
void pc_cache_host_free(void *p) { if (p && !hcache().put(p)) (void)hipHostFree(p); }
void pchip_trim_cache(void) { (void)hipDeviceSynchronize(); dcache().trim(); hcache().trim(); }      // the blocks finished runs left for the next ones go back to the driver
// initial capacity of the per-cluster arrays (default 128) and of the phantom array in rows (0 = the engine's estimate);
// both grow on demand, so these only matter to tests of the growth paths
void pchip_set_capacity(int clusters, int phantom_rows) { if (clusters > 0) g_cap_clusters = clusters; if (phantom_rows >= 0) g_cap_phantoms = phantom_rows; }   // negative: leave as is
void polychord_hip_set_batch_callback(polychord_batch_fn fn, void *user) { std::lock_guard<std::mutex> g(g_cb_mutex); g_batch_fn = fn; g_batch_user = user; }

double polychord_hip_keyed_uniform(unsigned seed, unsigned dom, unsigned shi, unsigned slo, unsigned idx)
{   // the engine's counter RNG on the host (same numbers as pc_dev.h pc_uniform)
    return h_uniform(seed, 0x504F4C59u, dom, shi, slo, idx);
}

void pchip_settings_default(pchip_settings *s, int nDims, int nDerived)
{   // defaults of the reference's C++ Settings (src/polychord/c_interface.cpp:6-39)
    std::memset(s, 0, sizeof(*s));
    s->nDims = nDims; s->nDerived = nDerived; s->nlive = 500; s->num_repeats = 5 * nDims; s->nprior = -1; s->nfail = -1;
    s->precision_criterion = 0.001; s->logzero = -1e30; s->max_ndead = -1; s->compression_factor = 0.36787944117144233;
    s->seed = -1; s->batch = 0; s->device = -1;
}

int pchip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pchip_run(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, pchip_result *out)
{
    return pchip_run_hooks(s, like, prior, nullptr, out);
}

int pchip_run_hooks(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, const pchip_hooks *hooks, pchip_result *out)
{
    if (s->num_repeats < 1) { std::fprintf(stderr, "polychord_hip: You need to set num_repeats. Suggestion: 5*nDims\n"); return 1; } // settings.f90:216
    if (s->num_repeats > 64 * PC_MASK_WORDS) { std::fprintf(stderr, "polychord_hip: num_repeats > %d unsupported\n", 64 * PC_MASK_WORDS); return 1; }
    if (like->kind == PC_LIKE_CALLBACK && !like->fn) { std::fprintf(stderr, "polychord_hip: callback likelihood without a function pointer\n"); return 1; }
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    std::memset(out, 0, sizeof(*out));
    Engine E;
    if (hooks) { E.dumper = hooks->dumper; E.on_update = hooks->on_update; E.hook_user = hooks->user; }
    int rc;
    auto t1 = t0, t2 = t0;
    try {
        E.setup(*s, *like, *prior);
        t1 = clk::now();
        rc = E.run(out);
        t2 = clk::now();
    } catch (const EngineError &e) {
        std::fprintf(stderr, "polychord_hip: %s\n", e.msg.c_str());
        (void)hipGetLastError();
        rc = e.code;
        pchip_result_free(out);
        t2 = clk::now();
    } catch (const std::bad_alloc &) {
        std::fprintf(stderr, "polychord_hip: out of host memory\n");
        rc = PC_RC_MEMORY;
        pchip_result_free(out);
        t2 = clk::now();
    } catch (...) {                       // thrown by a host hook (a binding's halt request): pass it on, resources released
        pchip_result_free(out);
        E.destroy();
        throw;
    }
    E.destroy();
    auto t3 = clk::now();
    if (rc == 0) {
        out->t_setup = std::chrono::duration<double>(t1 - t0).count();
        out->t_teardown = std::chrono::duration<double>(t3 - t2).count();
        out->t_results = std::chrono::duration<double>(t2 - t1).count() - out->t_total;
    }
    return rc;
}

// Several runs of one problem on ONE device, driven by the calling thread: every run is an engine of its own (own stream, own
// state), and the thread goes round them -- enqueue a round, look whether a stamp has arrived, finish that round, enqueue the
// next -- so that the kernels of up to `max_in_flight` runs are in the device's queues at any time.  (One host thread per
// run, as before, spent its time in the HIP runtime's locks: 533 launches per run from sixteen threads at once gave 1.5 x the
// throughput of one run; a single run keeps one wavefront per SIMD busy and the contraction kernel one CU.)
// Built-in device likelihoods only: a host callback belongs to its caller's thread.  Returns 0 or the first failing run's code.
// The runs of `seeds` on `device`, up to max_in_flight of them in step on one stream (Cohort): every phase of the round is gone
// through for all of them before anything is launched, then each kernel of the phase once for all.  PC_COHORT=0: the older
// scheduler below (one stream per run, kernels launched run by run).
static int pc_run_cohort(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds, int device,
                         int max_in_flight, pchip_result *results)
{
    static const bool prof = std::getenv("PC_DEBUG") && std::atoi(std::getenv("PC_DEBUG")) == 5;
    for (int k = 0; k < nseeds; ++k) std::memset(&results[k], 0, sizeof(pchip_result));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "polychord_hip: no HIP device available -- this engine has no CPU path\n"); return PC_RC_DEVICE; }
    (void)hipSetDevice(device >= 0 ? device % ndev : 0);
    const int W = std::max(1, std::min(std::min(max_in_flight, nseeds), 64));
    int worst = 0;
    int done_here = 0;
    for (int base = 0; base < nseeds && !worst; base += done_here) {
        int n = std::min(W, nseeds - base);
        const auto Tpre = std::chrono::steady_clock::now();
        Cohort co; bool own_streams = false;
        static const bool side_off = std::getenv("PC_COHORT_SIDE") && std::atoi(std::getenv("PC_COHORT_SIDE")) == 0;
        static const bool prio_off = !(std::getenv("PC_COHORT_PRIO") && std::atoi(std::getenv("PC_COHORT_PRIO")) == 1);      // (tried: 79 ms against 70 for sixteen runs -- off)
        // (the round's own kernels first, the bases of the next round in what they leave: stream priorities)
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);      // (least, greatest: numerically lower = more urgent)
        const auto Tp1 = std::chrono::steady_clock::now();
        // (a main stream whose hardware queue is known already, if the pool has one: the side stream is then picked without a test --
        //  a test is a millisecond, several once PyTorch lives in the process, and the pool's first stream was a different one of
        //  the engines' copy streams at every call)
        // (and the pair of the call before, if the pool still has it: whatever the runtime sets up for a stream at its first copy or
        //  launch is then there -- a call's first wait was 10 ... 24 ms now and then while the pair changed from call to call)
        static std::mutex last_m; static hipStream_t last_st[64] = {nullptr}, last_st2[64] = {nullptr};
        int devq = 0; (void)hipGetDevice(&devq); devq &= 63;
        int cls_main = -1, cls_side = -1;
        bool cohort_loaded = false;
        CohortLease lease(devq);
        if (prio_off || plo == phi) {
            hipStream_t want, want2;
            { std::lock_guard<std::mutex> g(last_m); want = last_st[devq]; want2 = last_st2[devq]; }
            std::lock_guard<std::mutex> gq(cstreams().m);             // (one group at a time picks: what it takes the next one avoids)
            std::vector<int> busy = cstreams().busy(devq);
            const bool loaded = !busy.empty();                    // another group is at work on this device: no class tests now
            // (more groups than hardware queues can keep apart: at least not on another group's MAIN stream's queue)
            if (busy.size() >= 4) busy = cstreams().busy(devq, true);
            auto free_cls = [&](hipStream_t x) { const int c = sclasses().known(x); return c >= 0 && std::find(busy.begin(), busy.end(), c) == busy.end(); };
            if (want) co.st = hpool().take_stream_if([&](hipStream_t x) { return x == want && (busy.empty() || free_cls(x)); });
            if (!co.st) co.st = hpool().take_stream_if([&](hipStream_t x) { return busy.empty() ? sclasses().known(x) >= 0 : free_cls(x); });
            if (!co.st) co.st = busy.empty() ? hpool().get_stream() : stream_avoiding(busy, loaded);
            const auto Tp2 = std::chrono::steady_clock::now();
            if (!busy.empty() || !side_off) cls_main = loaded ? sclasses().known(co.st) : sclasses().classify(co.st);
            if (!side_off) {
                if (cls_main >= 0) busy.push_back(cls_main);
                if (co.st == want && want2) co.st2 = hpool().take_stream_if([&](hipStream_t x) { return x == want2 && free_cls(x); });
                if (!co.st2) co.st2 = stream_avoiding(busy, loaded);
                cls_side = loaded ? sclasses().known(co.st2) : sclasses().classify(co.st2);
            }
            cohort_loaded = loaded;
            cstreams().take(devq, cls_main, true); cstreams().take(devq, cls_side); lease.hold(cls_main, cls_side);
            { std::lock_guard<std::mutex> g(last_m); last_st[devq] = co.st; last_st2[devq] = co.st2; }
            if (prof) std::fprintf(stderr, "polychord_hip dbg cohort: priority range %.2f ms, main stream %.2f ms, side stream %.2f ms\n", std::chrono::duration<double>(Tp1 - Tpre).count() * 1e3, std::chrono::duration<double>(Tp2 - Tp1).count() * 1e3, std::chrono::duration<double>(std::chrono::steady_clock::now() - Tp2).count() * 1e3); }
        else { HIPCHK(hipStreamCreateWithPriority(&co.st, hipStreamNonBlocking, phi)); if (!side_off) HIPCHK(hipStreamCreateWithPriority(&co.st2, hipStreamNonBlocking, plo)); own_streams = true; }
        if (co.st2) { co.ev_up = hpool().get_sync_event(); co.ev_next = hpool().get_sync_event(); }
        static const bool stc_off = std::getenv("PC_COHORT_COPY_STREAMS") && std::atoi(std::getenv("PC_COHORT_COPY_STREAMS")) == 0;
        if (!stc_off && !own_streams) { co.stc[0] = stream_beside({co.st, co.st2}, cohort_loaded); co.stc[1] = stream_beside({co.st, co.st2, co.stc[0]}, cohort_loaded); }
        std::vector<Engine *> E((size_t)n, nullptr);
        std::vector<char> live((size_t)n, 0), enq((size_t)n, 0);
        const auto T0 = std::chrono::steady_clock::now();
        long rounds = 0; double t_begin = 0, t_end = 0, t_wait = 0, t_enq = 0, t_fin = 0, t_fl = 0, t_comp = 0, t_end_dev = 0, t_fwait = 0; long n_fwait = 0;
        static const bool fibers_on = !(std::getenv("PC_COHORT_FIBERS") && std::atoi(std::getenv("PC_COHORT_FIBERS")) == 0);
        std::vector<Fiber> fibs;
        int *h_totals = nullptr; size_t totals_cap = 0;
        double t_setup_max = 0; int n_comp_pass = 0;
        struct EndBatch { std::vector<int> fin, rcs; hipEvent_t ev = nullptr, ev2 = nullptr; int dev = 0; std::thread th; };
        std::vector<std::unique_ptr<EndBatch>> endings;
        auto nowc = [] { return std::chrono::steady_clock::now(); };
        auto secc = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        auto close = [&](int k, int rc) {
            if (rc != 0) { pchip_result_free(&results[base + k]); if (!worst) worst = rc; }
            E[k]->destroy(); delete E[k]; E[k] = nullptr; live[k] = 0;
        };
        try {
            {
                // set-up and live points of the runs: the first one by itself (a run that does not fit, or fails, alone is the
                // call's failure), then the others -- a tenth of a millisecond of host work each, 7 ms of sixty-four runs' 175.
                // (Shared out among four threads it was TWICE as long -- 13 ms -- the runtime's calls queue for one another:
                //  PC_COHORT_SETUP_THREADS, one by default)
                std::vector<int> rc_begin((size_t)n, -1), err((size_t)n, 0);
                std::vector<std::string> errmsg((size_t)n);
                std::atomic<bool> failed{false};
                int devnow = 0; (void)hipGetDevice(&devnow);
                auto setup_one = [&](int k) {
                    E[k] = new Engine; E[k]->co = &co;
                    pchip_settings c = *s; c.seed = seeds[base + k]; c.device = device;
                    const auto q0 = nowc();
                    try { E[k]->setup(c, *like, *prior); rc_begin[k] = E[k]->begin(); }
                    catch (const EngineError &e) { err[k] = e.code ? e.code : PC_RC_DEVICE; errmsg[k] = e.msg; failed = true; }
                    catch (const std::bad_alloc &) { err[k] = PC_RC_MEMORY; errmsg[k] = "out of host memory"; failed = true; }
                    if (k == 0) t_setup_max = secc(q0, nowc());
                };
                const auto b0 = nowc();
                setup_one(0);
                static const int setup_threads = std::getenv("PC_COHORT_SETUP_THREADS") ? std::max(1, std::atoi(std::getenv("PC_COHORT_SETUP_THREADS"))) : 1;
                if (n > 1 && !failed) {
                    std::atomic<int> nextk{1};
                    auto worker = [&] { (void)hipSetDevice(devnow); for (int k; !failed && (k = nextk.fetch_add(1)) < n;) setup_one(k); };
                    std::vector<std::thread> th;
                    for (int t = 1; t < std::min(setup_threads, n - 1) && n >= 8; ++t) th.emplace_back(worker);
                    worker();
                    for (auto &t : th) t.join();
                }
                t_begin += secc(b0, nowc());
                for (int k = 0; k < n; ++k) {
                    if (!err[k] && E[k]) continue;
                    if (err[k] && (err[k] != PC_RC_MEMORY || k == 0)) throw EngineError{err[k], errmsg[k]};
                    // no memory for one more run of this size next to the k that are set up: those go in step, the others after them
                    (void)hipGetLastError();
                    for (int j = k; j < n; ++j) if (E[j]) { try { E[j]->destroy(); } catch (...) {} delete E[j]; E[j] = nullptr; }
                    n = k;
                    break;
                }
                for (int k = 0; k < n; ++k) {
                    const int rc = rc_begin[k];
                    if (rc >= 0) { close(k, rc ? rc : PC_RC_DEVICE); continue; }
                    live[k] = 1;
                }
            }
            int nlive = 0;
            for (int k = 0; k < n; ++k) nlive += live[k];
            while (nlive > 0 && !worst) {
                {   // phantom arrays that are full: compacted together, one wait for all.  (Only the full ones: taking the arrays
                    // that are more than half full along -- the runs fill theirs at slightly different rates, and a pass a round
                    // later is another wait of the whole cohort -- kept the passes at three a call, and changed the last bits of
                    // some runs: the update's partial sums are grouped by the array's extent, so a run must compact exactly
                    // when it would alone.  tools/dev/fuzz_in_step.py found it; PC_COHORT_COMPACT_ALIGN=1 brings it back)
                    int nc = 0;
                    for (int k = 0; k < n; ++k) if (live[k] && E[k]->compact_wanted()) nc++;
                    if (nc) {
                        const auto c0 = nowc();
                        static const bool align_off = !(std::getenv("PC_COHORT_COMPACT_ALIGN") && std::atoi(std::getenv("PC_COHORT_COMPACT_ALIGN")) == 1);
                        std::vector<char> cmp((size_t)n, 0);
                        for (int k = 0; k < n; ++k) if (live[k] && (E[k]->compact_wanted() || (!align_off && E[k]->compact_worth_it()))) { cmp[k] = 1; E[k]->compact_record(); }
                        co.flush();
                        if ((size_t)n > totals_cap) { if (h_totals) hfree(h_totals); h_totals = halloc<int>((size_t)n); totals_cap = (size_t)n; }
                        for (int k = 0; k < n; ++k) if (cmp[k]) HIPCHK(hipMemcpyAsync(&h_totals[k], E[k]->d_total, sizeof(int), hipMemcpyDeviceToHost, co.st));
                        HIPCHK(hipStreamSynchronize(co.st));
                        for (int k = 0; k < n; ++k) if (cmp[k]) E[k]->compact_finish(h_totals[k]);
                        t_comp += secc(c0, nowc()); n_comp_pass++;
                    }
                }
                { const auto a0 = nowc(); for (int k = 0; k < n; ++k) enq[k] = (live[k] && E[k]->round_enqueue()) ? 1 : 0; const auto a1 = nowc(); t_enq += secc(a0, a1);
                  co.flush(); t_fl += secc(a1, nowc()); }
                { const auto w0 = nowc(); for (int k = 0; k < n; ++k) if (enq[k]) while (!E[k]->round_ready()) __builtin_ia32_pause(); t_wait += secc(w0, nowc()); }
                { const auto a0 = nowc();
                  // what the round did: a run whose update has to wait for the device in the middle (clustering, files, hooks) makes it as a
                  // fiber of this thread; the others at once
                  std::vector<int> wk;
                  for (int k = 0; k < n; ++k) {
                      if (!enq[k]) continue;
                      if (fibers_on && E[k]->finish_may_wait()) wk.push_back(k);
                      else if (!E[k]->round_finish()) enq[k] = 0;
                  }
                  if (!wk.empty()) {
                      if (fibs.size() < wk.size()) fibs.resize(wk.size());
                      std::vector<char> ok(wk.size(), 1);
                      for (size_t a = 0; a < wk.size(); ++a) {
                          Fiber &f = fibs[a];
                          f.make_stack();
                          f.started = false; f.done = false; f.cancel = false; f.err = nullptr;
                          Engine *e = E[wk[a]]; char *okp = &ok[a];
                          f.fn = [e, okp] { *okp = e->round_finish() ? 1 : 0; };
                          e->fib = &f;
                      }
                      std::exception_ptr first_err;
                      for (;;) {
                          bool waiting = false;
                          for (size_t a = 0; a < wk.size(); ++a) {
                              Fiber &f = fibs[a];
                              if (f.done) continue;
                              f.resume();
                              if (f.err && !first_err) first_err = f.err;
                              if (!f.done) waiting = true;
                          }
                          if (first_err || !waiting) break;
                          // one launch of what they all wrote down, one wait for all of them
                          const auto w0 = nowc();
                          co.flush();
                          for (int spins = 0; spins < 200000; ++spins) { const hipError_t q = hipStreamQuery(co.st); if (q != hipErrorNotReady) { HIPCHK(q); break; } __builtin_ia32_pause(); }
                          HIPCHK(hipStreamSynchronize(co.st));
                          t_fwait += secc(w0, nowc()); n_fwait++;
                      }
                      if (first_err) {
                          // the fibers still suspended hold locals, pinned blocks and copies written down for a flush that will not come: what
                          // they wrote down is dropped, and each is resumed once more with the cancel flag -- its wait throws, its frames unwind
                          co.pre.clear(); co.post.clear(); co.pend.clear(); co.pre_copies.clear(); co.post_copies.clear();
                          for (size_t a = 0; a < wk.size(); ++a) {
                              Fiber &f = fibs[a];
                              if (f.started && !f.done) { f.cancel = true; f.resume(); }
                          }
                          co.pre.clear(); co.post.clear(); co.pend.clear(); co.pre_copies.clear(); co.post_copies.clear();
                      }
                      for (size_t a = 0; a < wk.size(); ++a) { E[wk[a]]->fib = nullptr; if (!ok[a]) enq[wk[a]] = 0; }
                      if (first_err) std::rethrow_exception(first_err);
                  }
                  const auto a1 = nowc(); t_fin += secc(a0, a1);
                  co.flush(); t_fl += secc(a1, nowc()); }
                rounds++;
                bool any_done = false;
                for (int k = 0; k < n; ++k) any_done = any_done || (live[k] && !enq[k]);
                if (any_done) {
                    // the runs that are over end together: their kill-off in one launch, and what their results need asked of the
                    // device behind it.  Nobody waits here: a thread takes the batch from there (two events, then the host's half
                    // of the endings -- results, buffers given back, a third of a millisecond per run, shared out among a few
                    // threads) while the runs that are left go on with their rounds
                    const auto e0 = nowc();
                    co.flush();
                    for (int k = 0; k < n; ++k) if (live[k] && !enq[k] && !E[k]->r_rc) E[k]->end_a(true);
                    co.flush();
                    for (int k = 0; k < n; ++k) if (live[k] && !enq[k] && !E[k]->r_rc) E[k]->end_a2();
                    endings.emplace_back(new EndBatch);
                    EndBatch *eb = endings.back().get();
                    for (int k = 0; k < n; ++k) if (live[k] && !enq[k]) eb->fin.push_back(k);
                    eb->rcs.assign(eb->fin.size(), 0);
                    eb->dev = E[eb->fin[0]]->dev;
                    eb->ev = hpool().get_sync_event(); HIPCHK(hipEventRecord(eb->ev, co.st));
                    if (co.st2) { eb->ev2 = hpool().get_sync_event(); HIPCHK(hipEventRecord(eb->ev2, co.st2)); }      // (bases drawn ahead for a run that is over: not into freed memory)
                    for (int k : eb->fin) { live[k] = 0; nlive--; if (E[k]->r_rc && !worst) worst = E[k]->r_rc; }      // (a run that failed stops the others at once)
                    eb->th = std::thread([&E, &results, base, eb] {
                        (void)hipSetDevice(eb->dev);
                        const auto w0 = std::chrono::steady_clock::now();
                        const hipError_t w1 = hipEventSynchronize(eb->ev), w2 = eb->ev2 ? hipEventSynchronize(eb->ev2) : hipSuccess;
                        g_dbg_evwait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
                        auto finish_one = [&](size_t a) {
                            const int k = eb->fin[a];
                            int r = E[k]->r_rc;
                            if (!r && (w1 != hipSuccess || w2 != hipSuccess)) r = PC_RC_DEVICE;
                            const auto q0 = std::chrono::steady_clock::now();
                            if (!r) {
                                try { E[k]->end_wait_aside(); r = E[k]->end_b(&results[base + k]); }
                                catch (const EngineError &e) { std::fprintf(stderr, "polychord_hip: %s\n", e.msg.c_str()); r = e.code; }
                                catch (const std::bad_alloc &) { r = PC_RC_MEMORY; }
                            }
                            const auto q1 = std::chrono::steady_clock::now();
                            if (r != 0) pchip_result_free(&results[base + k]);
                            try { E[k]->destroy(r == 0 && E[k]->st_side == nullptr); } catch (...) {}      // (end_b has waited for the copy stream, this thread for the cohort's two)
                            delete E[k]; E[k] = nullptr; eb->rcs[a] = r;
                            g_dbg_endb_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(q1 - q0).count();
                            g_dbg_destroy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - q1).count();
                        };
                        const size_t nth = std::min<size_t>(eb->fin.size(), 8);
                        if (nth <= 1) { for (size_t a = 0; a < eb->fin.size(); ++a) finish_one(a); }
                        else {
                            std::atomic<size_t> nexta{0};
                            auto worker = [&] { (void)hipSetDevice(eb->dev); for (size_t a; (a = nexta.fetch_add(1)) < eb->fin.size();) finish_one(a); };
                            std::vector<std::thread> th;
                            for (size_t t = 1; t < nth; ++t) th.emplace_back(worker);
                            worker();
                            for (auto &t : th) t.join();
                        }
                    });
                    t_end_dev += secc(e0, nowc());
                }
            }
        }
        catch (const EngineError &e) { std::fprintf(stderr, "polychord_hip: %s\n", e.msg.c_str()); (void)hipGetLastError(); if (!worst) worst = e.code; }
        catch (const std::bad_alloc &) { std::fprintf(stderr, "polychord_hip: out of host memory\n"); if (!worst) worst = PC_RC_MEMORY; }
        catch (const std::exception &e) { std::fprintf(stderr, "polychord_hip: %s\n", e.what()); if (!worst) worst = PC_RC_DEVICE; }      // (a thread that could not be started: nothing leaves through the C interface)
        {   // the endings under way
            const auto e0 = nowc();
            for (auto &eb : endings) {
                if (eb->th.joinable()) eb->th.join();
                for (int r : eb->rcs) if (r != 0 && !worst) worst = r;
                if (eb->ev) hpool().put_sync_event(eb->ev); if (eb->ev2) hpool().put_sync_event(eb->ev2);
            }
            t_end += secc(e0, nowc());
        }
        if (co.st2) (void)hipStreamSynchronize(co.st2);
        for (int k = 0; k < n; ++k) if (E[k]) { pchip_result_free(&results[base + k]); try { E[k]->destroy(); } catch (...) {} delete E[k]; E[k] = nullptr; }
        if (prof) { std::fprintf(stderr, "polychord_hip dbg cohort: of enqueue: nursery %.2f ms (compaction %.2f), capacity %.2f\n", g_dbg_nursery_ns.exchange(0) * 1e-6, g_dbg_compact_ns.exchange(0) * 1e-6, g_dbg_capacity_ns.exchange(0) * 1e-6);
                    std::fprintf(stderr, "polychord_hip dbg cohort: %zu ending batches: events %.2f ms, results %.2f ms, teardown %.2f ms (summed over threads); the block caches hold %.2f GB of device and %.2f GB of pinned memory; teardown: device blocks %.2f, the rest %.2f ms\n", endings.size(), g_dbg_evwait_ns.exchange(0) * 1e-6, g_dbg_endb_ns.exchange(0) * 1e-6, g_dbg_destroy_ns.exchange(0) * 1e-6, dcache().cached / 1073741824.0, hcache().cached / 1073741824.0, g_dbg_d1.exchange(0) * 1e-6, g_dbg_d2.exchange(0) * 1e-6); }
        if (prof) std::fprintf(stderr, "polychord_hip dbg cohort: the first run's set-up and live points %.2f ms\n", t_setup_max * 1e3);
        if (prof) std::fprintf(stderr, "polychord_hip dbg cohort: trips to the driver: %lld device blocks (%.2f ms), %lld pinned blocks (%.2f ms), %lld streams (%.2f ms)\n", g_dbg_miss_n[0].exchange(0), g_dbg_miss_ns[0].exchange(0) * 1e-6, g_dbg_miss_n[1].exchange(0), g_dbg_miss_ns[1].exchange(0) * 1e-6, g_dbg_mk_stream_n.exchange(0), g_dbg_mk_stream_ns.exchange(0) * 1e-6);
        if (prof) std::fprintf(stderr, "polychord_hip dbg cohort: %d runs, %ld rounds, streams %.2f ms, wall %.2f ms (setup + begin %.2f, compactions %.2f in %d passes, enqueue %.2f, finish %.2f, launches %.2f, waiting for the device %.2f, the endings' requests %.2f, waiting for the endings %.2f); %ld records launched together, %ld one by one\n", n, rounds, std::chrono::duration<double>(T0 - Tpre).count() * 1e3,
                               std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count() * 1e3, t_begin * 1e3, t_comp * 1e3, n_comp_pass, t_enq * 1e3, t_fin * 1e3, t_fl * 1e3, t_wait * 1e3, t_end_dev * 1e3, t_end * 1e3, co.n_fused, co.n_single);
        for (Fiber &f : fibs) f.free_stack();
        if (prof) std::fprintf(stderr, "polychord_hip dbg cohort: of finish: %ld shared waits, %.2f ms\n", n_fwait, t_fwait * 1e3);
        co.destroy();
        if (h_totals) hfree(h_totals);
        (void)hipStreamSynchronize(co.st);
        lease.release();
        if (own_streams) (void)hipStreamDestroy(co.st); else hpool().put_stream(co.st);
        if (co.st2) { (void)hipStreamSynchronize(co.st2); if (own_streams) (void)hipStreamDestroy(co.st2); else hpool().put_stream(co.st2); hpool().put_sync_event(co.ev_up); hpool().put_sync_event(co.ev_next); }
        for (int q = 0; q < 2; ++q) if (co.stc[q]) { (void)hipStreamSynchronize(co.stc[q]); hpool().put_stream(co.stc[q]); co.stc[q] = nullptr; }
        done_here = n;
    }
    return worst;
}

int pc_run_many(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, int nseeds, const int *seeds, int device,
                int max_in_flight, pchip_result *results)
{
    {
        static const bool cohort_off = std::getenv("PC_COHORT") && std::atoi(std::getenv("PC_COHORT")) == 0;
        if (!cohort_off) return pc_run_cohort(s, like, prior, nseeds, seeds, device, max_in_flight, results);
    }
    struct Job { Engine *E = nullptr; int k = -1; bool waiting = false; };
    std::vector<Job> jobs((size_t)std::max(1, std::min(max_in_flight, nseeds)));
    int next = 0, active = 0, worst = 0;
    for (int k = 0; k < nseeds; ++k) std::memset(&results[k], 0, sizeof(pchip_result));
    auto finish = [&](Job &j, int rc) {
        if (rc != 0) { pchip_result_free(&results[j.k]); if (!worst) worst = rc; }
        j.E->destroy(); delete j.E; j.E = nullptr; j.waiting = false; active--;
    };
    // one step of a job; exceptions of the engine end that job only
    auto guarded = [&](Job &j, auto &&fn) {
        try { fn(); }
        catch (const EngineError &e) { std::fprintf(stderr, "polychord_hip: %s\n", e.msg.c_str()); (void)hipGetLastError(); finish(j, e.code); }
        catch (const std::bad_alloc &) { std::fprintf(stderr, "polychord_hip: out of host memory\n"); finish(j, PC_RC_MEMORY); }
    };
    static const bool prof = std::getenv("PC_DEBUG") && std::atoi(std::getenv("PC_DEBUG")) == 5;
    double t_begin = 0, t_enq = 0, t_fin = 0, t_end = 0; long n_enq = 0, n_poll = 0;
    const auto T0 = std::chrono::steady_clock::now();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    while (next < nseeds || active > 0) {
        for (Job &j : jobs) {
            if (!j.E) {
                if (next >= nseeds || worst) continue;
                j.k = next++; j.E = new Engine; active++;
                guarded(j, [&] {
                    pchip_settings c = *s; c.seed = seeds[j.k]; c.device = device;
                    j.E->setup(c, *like, *prior);
                    const auto a0 = now();
                    const int rc = j.E->begin();
                    t_begin += secs(a0, now());
                    if (rc >= 0) { finish(j, rc ? rc : PC_RC_DEVICE); return; }
                    if (!j.E->round_enqueue()) { const int r = j.E->r_rc ? j.E->r_rc : j.E->end(&results[j.k]); finish(j, r); return; }
                    j.waiting = true;
                });
            } else if (j.waiting) {
                guarded(j, [&] {
                    n_poll++;
                    if (!j.E->round_ready()) return;
                    const auto a0 = now();
                    const bool go = j.E->round_finish();
                    const auto a1 = now(); t_fin += secs(a0, a1);
                    const bool go2 = go && j.E->round_enqueue();
                    const auto a2 = now(); t_enq += secs(a1, a2); n_enq++;
                    if (!go2) { const int r = j.E->r_rc ? j.E->r_rc : j.E->end(&results[j.k]); t_end += secs(a2, now()); finish(j, r); }
                });
            }
        }
        if (worst && active == 0) break;
    }
    if (prof) std::fprintf(stderr, "polychord_hip dbg many: %d runs, wall %.2f ms; begin %.2f, finish(+updates) %.2f, enqueue %.2f (%ld rounds), end %.2f ms; %ld polls\n",
                           nseeds, secs(T0, now()) * 1e3, t_begin * 1e3, t_fin * 1e3, t_enq * 1e3, n_enq, t_end * 1e3, n_poll);
    return worst;
}

void pchip_result_free(pchip_result *r)
{
    hfree(r->dead); hfree(r->logweights); hfree(r->entry); std::free(r->live); std::free(r->logZp); std::free(r->varlogZp);
    if (r->d_records) pc_cache_dev_free(r->d_records);
    std::free(r->post_mean); std::free(r->post_var); std::free(r->live_cluster);
    std::memset(r, 0, sizeof(*r));
}

// kernel-level entry for the parity tests: run K0 + K1 for `nchains` chains, chain c seeded from
// seeds[c] (row of nTotal doubles), all under the same Cholesky factor and contour.
int pchip_slice_chains(const pchip_settings *s, const pchip_like *like, const pchip_prior *prior, unsigned batch,
                       int nchains, const double *seeds, const double *chol, double contour, double *babies_out,
                       double *nhats_out, int *nlike_out)
{
    pchip_settings c = *s;
    c.nlive = nchains; c.nprior = nchains; c.batch = nchains; c.do_clustering = 0;
    Engine E;
    int rc;
    try {
        E.setup(c, *like, *prior);
        PcState &S = E.S;
        const int nT = S.nT, D = S.D, nr = S.nr;
        HIPCHK(hipMemcpy(S.live, seeds, sizeof(double) * (size_t)nchains * nT, hipMemcpyHostToDevice));
        std::vector<double> ll(nchains); std::vector<int> idn(nchains), zero(nchains, 0);
        for (int i = 0; i < nchains; ++i) { ll[i] = seeds[(size_t)i * nT + S.l0]; idn[i] = i; }
        HIPCHK(hipMemcpy(S.live_logL, ll.data(), sizeof(double) * nchains, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.live_cluster, zero.data(), sizeof(int) * nchains, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.live_pos, idn.data(), sizeof(int) * nchains, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.cl_list, idn.data(), sizeof(int) * nchains, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.cl_n, &nchains, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.logLp, &contour, sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(S.chol, chol, sizeof(double) * D * D, hipMemcpyHostToDevice));
        S.seed_override = 1;
        rc = pc_launch_nhats(&S, batch, nchains, E.st) || pc_launch_slice(&S, batch, nchains, E.st);
        HIPCHK(hipStreamSynchronize(E.st));
        HIPCHK(hipMemcpy(babies_out, S.babies, sizeof(double) * (size_t)nchains * nr * nT, hipMemcpyDeviceToHost));
        if (nhats_out) HIPCHK(hipMemcpy(nhats_out, S.nhat, sizeof(double) * (size_t)nchains * nr * D, hipMemcpyDeviceToHost));
        if (nlike_out) HIPCHK(hipMemcpy(nlike_out, S.ch_nlike, sizeof(int) * nchains, hipMemcpyDeviceToHost));
    } catch (const EngineError &e) {
        std::fprintf(stderr, "polychord_hip: %s\n", e.msg.c_str());
        (void)hipGetLastError();
        rc = e.code;
    }
    E.destroy();
    return rc;
}

}  // extern "C"
