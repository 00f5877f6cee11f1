// pc_dev.h -- device-side building blocks of the MI355X nested-sampling engine (gfx950 only).
//
// Counter-based RNG (Philox4x32-10), AS241 inverse normal, wave64 DPP reductions,
// log-space arithmetic and the built-in likelihood / prior functors.
// Reference behaviour restated (file:line under the reference tree):
//   random_utils.F90:119-131 (uniforms), :251-263 (gaussian via inverse CDF),
//   utils.F90:806-966 (AS241), utils.F90:362-439 (log-space sums),
//   likelihoods/examples/{gaussian,rastrigin,twin_gaussian,random_gaussian}.f90, priors.f90:40-55.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PC_WAVE 64
#define PC_SLICE_STRIDE 128
enum { PC_DOM_LIVEGEN = 0, PC_DOM_SEED = 1, PC_DOM_NHAT = 2, PC_DOM_SHUFFLE = 3, PC_DOM_SLICE = 4,
       PC_DOM_PHANTOM = 5, PC_DOM_POST = 6, PC_DOM_SEQ = 0xFFFF };
enum { PC_LIKE_CALLBACK = 0, PC_LIKE_GAUSSIAN = 1, PC_LIKE_RASTRIGIN = 2, PC_LIKE_TWIN_GAUSSIAN = 3,
       PC_LIKE_CORR_GAUSSIAN = 4 };

#define PC_HUGE 1.7976931348623157e308
#define PC_LOG_TWO_PI 1.8378770664093453
#define PC_TWO_PI 6.283185307179586

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{   // 53 random bits, centred: never exactly 0 or 1
    const uint64_t w = ((uint64_t)hi << 32) | lo;
    return ((double)(w >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// uniform #idx of stream (shi,slo) in domain dom; idx and idx^1 share one Philox call
__device__ __forceinline__ double pc_uniform(uint32_t k0, uint32_t k1, uint32_t dom, uint32_t shi,
                                             uint32_t slo, uint32_t idx)
{
    const u32x4 o = philox4x32_10(idx >> 1, slo, shi, dom, k0, k1);
    return (idx & 1u) ? u53(o.z, o.w) : u53(o.x, o.y);
}
// both uniforms of call #call (indices 2*call, 2*call+1)
__device__ __forceinline__ void pc_uniform2(uint32_t k0, uint32_t k1, uint32_t dom, uint32_t shi,
                                            uint32_t slo, uint32_t call, double &u0, double &u1)
{
    const u32x4 o = philox4x32_10(call, slo, shi, dom, k0, k1);
    u0 = u53(o.x, o.y); u1 = u53(o.z, o.w);
}

// ---------------------------------------------------------------- AS241 (PPND16)
__device__ __forceinline__ double poly8(const double *a, double x)
{
    double v = 0.0;
#pragma unroll
    for (int i = 7; i >= 0; --i) v = v * x + a[i];
    return v;
}

__device__ inline double pc_inv_normal_cdf(double p)
{
    const double a[8] = { 3.3871328727963666080e+00, 1.3314166789178437745e+02,
        1.9715909503065514427e+03, 1.3731693765509461125e+04, 4.5921953931549871457e+04,
        6.7265770927008700853e+04, 3.3430575583588128105e+04, 2.5090809287301226727e+03 };
    const double b[8] = { 1.0, 4.2313330701600911252e+01, 6.8718700749205790830e+02,
        5.3941960214247511077e+03, 2.1213794301586595867e+04, 3.9307895800092710610e+04,
        2.8729085735721942674e+04, 5.2264952788528545610e+03 };
    const double c[8] = { 1.42343711074968357734e+00, 4.63033784615654529590e+00,
        5.76949722146069140550e+00, 3.64784832476320460504e+00, 1.27045825245236838258e+00,
        2.41780725177450611770e-01, 2.27238449892691845833e-02, 7.74545014278341407640e-04 };
    const double d[8] = { 1.0, 2.05319162663775882187e+00, 1.67638483018380384940e+00,
        6.89767334985100004550e-01, 1.48103976427480074590e-01, 1.51986665636164571966e-02,
        5.47593808499534494600e-04, 1.05075007164441684324e-09 };
    const double e[8] = { 6.65790464350110377720e+00, 5.46378491116411436990e+00,
        1.78482653991729133580e+00, 2.96560571828504891230e-01, 2.65321895265761230930e-02,
        1.24266094738807843860e-03, 2.71155556874348757815e-05, 2.01033439929228813265e-07 };
    const double f[8] = { 1.0, 5.99832206555887937690e-01, 1.36929880922735805310e-01,
        1.48753612908506148525e-02, 7.86869131145613259100e-04, 1.84631831751005468180e-05,
        1.42151175831644588870e-07, 2.04426310338993978564e-15 };
    if (p <= 0.0) return -PC_HUGE;
    if (p >= 1.0) return PC_HUGE;
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * poly8(a, r) / poly8(b, r);
    }
    double r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    double v;
    if (r <= 5.0) { r -= 1.6; v = poly8(c, r) / poly8(d, r); }
    else          { r -= 5.0; v = poly8(e, r) / poly8(f, r); }
    return (q < 0.0) ? -v : v;
}

// The same function in two halves, for callers that have many deviates to make: the central branch (85 % of the arguments)
// costs two polynomials and a division, the tails a log and a square root on top -- a wavefront that evaluates the whole
// function pays for both branches on every call, because some lane is always in a tail.  pc_inv_normal_central returns the
// value, or sets `tail` and leaves the argument to be finished by pc_inv_normal_tail (collected and done together).
__device__ inline double pc_inv_normal_central(double p, bool &tail)
{
    const double a[8] = { 3.3871328727963666080e+00, 1.3314166789178437745e+02,
        1.9715909503065514427e+03, 1.3731693765509461125e+04, 4.5921953931549871457e+04,
        6.7265770927008700853e+04, 3.3430575583588128105e+04, 2.5090809287301226727e+03 };
    const double b[8] = { 1.0, 4.2313330701600911252e+01, 6.8718700749205790830e+02,
        5.3941960214247511077e+03, 2.1213794301586595867e+04, 3.9307895800092710610e+04,
        2.8729085735721942674e+04, 5.2264952788528545610e+03 };
    tail = false;
    if (p <= 0.0) return -PC_HUGE;
    if (p >= 1.0) return PC_HUGE;
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * poly8(a, r) / poly8(b, r);
    }
    tail = true;
    return 0.0;
}
__device__ inline double pc_inv_normal_tail(double p)
{
    const double c[8] = { 1.42343711074968357734e+00, 4.63033784615654529590e+00,
        5.76949722146069140550e+00, 3.64784832476320460504e+00, 1.27045825245236838258e+00,
        2.41780725177450611770e-01, 2.27238449892691845833e-02, 7.74545014278341407640e-04 };
    const double d[8] = { 1.0, 2.05319162663775882187e+00, 1.67638483018380384940e+00,
        6.89767334985100004550e-01, 1.48103976427480074590e-01, 1.51986665636164571966e-02,
        5.47593808499534494600e-04, 1.05075007164441684324e-09 };
    const double e[8] = { 6.65790464350110377720e+00, 5.46378491116411436990e+00,
        1.78482653991729133580e+00, 2.96560571828504891230e-01, 2.65321895265761230930e-02,
        1.24266094738807843860e-03, 2.71155556874348757815e-05, 2.01033439929228813265e-07 };
    const double f[8] = { 1.0, 5.99832206555887937690e-01, 1.36929880922735805310e-01,
        1.48753612908506148525e-02, 7.86869131145613259100e-04, 1.84631831751005468180e-05,
        1.42151175831644588870e-07, 2.04426310338993978564e-15 };
    const double q = p - 0.5;
    double r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    double v;
    if (r <= 5.0) { r -= 1.6; v = poly8(c, r) / poly8(d, r); }
    else          { r -= 5.0; v = poly8(e, r) / poly8(f, r); }
    return (q < 0.0) ? -v : v;
}

// ---------------------------------------------------------------- log-space sums
__device__ __forceinline__ double pc_logaddexp(double a, double b)
{   // utils.F90:377-389:  (a > b) ? a + log(exp(b - a) + 1) : b + log(exp(a - b) + 1)
    // More than 37 nats apart the smaller term does not reach the larger one's last bit: exp(-37) = 8.5e-17 < 2^-53, so exp(d) + 1 IS 1,
    // its logarithm 0 and the sum the larger term -- the same bits without the exponential and the logarithm (140 + 480 cycles of a
    // wavefront: with the twin Gaussian's two modes a hundred nats apart that was 40 % of a slice, five evaluations each).
    const bool ab = a > b;
    const double hi = ab ? a : b, d = ab ? b - a : a - b;
    if (d < -37.0) return hi;
    return hi + log(exp(d) + 1.0);
}

// workgroup barrier for data that lives in LDS only: waits for this wave's LDS traffic, not for its stores to HBM (a
// __syncthreads() after global stores holds every wave until they are acknowledged -- microseconds on a one-CU kernel)
__device__ __forceinline__ void pc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- wave64 DPP reductions
// Butterfly inside each row of 16 lanes with DPP (no LDS traffic), then the four row
// results are combined through v_readlane.  Every lane ends with the bit-identical total
// (IEEE add/min are commutative and each level combines the same two partial results).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

#define PC_DPP_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define PC_DPP_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define PC_DPP_HALF_MIRROR 0x141
#define PC_DPP_MIRROR 0x140

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row_sum(double v)
{
    v += dpp_f64<PC_DPP_XOR1>(v);
    v += dpp_f64<PC_DPP_XOR2>(v);
    v += dpp_f64<PC_DPP_HALF_MIRROR>(v);
    v += dpp_f64<PC_DPP_MIRROR>(v);
    return v;
}

// sum over the whole wave; NROWS = number of 16-lane rows that can hold non-zero data
template <int NROWS>
__device__ __forceinline__ double wave_sum(double v)
{
    v = row_sum(v);
    if (NROWS == 1) return readlane_f64(v, 0);
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    if (NROWS == 2) return r0 + r1;
    const double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_f64<PC_DPP_XOR1>(v));
    v = fmax(v, dpp_f64<PC_DPP_XOR2>(v));
    v = fmax(v, dpp_f64<PC_DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<PC_DPP_MIRROR>(v));
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

// lexicographic (value, key) minimum over the wave: smallest value, ties -> smallest key
struct vk_t { double v; int k; };
__device__ __forceinline__ vk_t vk_min(vk_t a, vk_t b)
{
    const bool take_b = (b.v < a.v) || (b.v == a.v && b.k < a.k);
    return take_b ? b : a;
}
template <int CTRL>
__device__ __forceinline__ vk_t vk_dpp(vk_t a) { return vk_t{ dpp_f64<CTRL>(a.v), dpp_i32<CTRL>(a.k) }; }

__device__ __forceinline__ vk_t wave_argmin(vk_t a)
{
    a = vk_min(a, vk_dpp<PC_DPP_XOR1>(a));
    a = vk_min(a, vk_dpp<PC_DPP_XOR2>(a));
    a = vk_min(a, vk_dpp<PC_DPP_HALF_MIRROR>(a));
    a = vk_min(a, vk_dpp<PC_DPP_MIRROR>(a));
    vk_t r0{ readlane_f64(a.v, 0), __builtin_amdgcn_readlane(a.k, 0) };
    vk_t r1{ readlane_f64(a.v, 16), __builtin_amdgcn_readlane(a.k, 16) };
    vk_t r2{ readlane_f64(a.v, 32), __builtin_amdgcn_readlane(a.k, 32) };
    vk_t r3{ readlane_f64(a.v, 48), __builtin_amdgcn_readlane(a.k, 48) };
    return vk_min(vk_min(r0, r1), vk_min(r2, r3));
}

// ---------------------------------------------------------------- likelihood / prior specs
struct PcLike {
    int kind;
    double mu, sigma;        // gaussian mean/width; twin gaussian width
    double norm;             // host-precomputed -D (log sigma + log(2 pi)/2)
    double inv_sigma;        // 1 / sigma
    double log_vn;           // log volume of the unit D-ball (utils.F90:754-760)
    const double *invcov;    // corr gaussian: device pointer, row-major D x D
    const double *mean;      // corr gaussian: device pointer (D)
    double logdetcov;
};
struct PcPrior {
    int kind;                // 0 callback (host), 1 uniform box
    const double *lo, *hi;   // device pointers or nullptr => [0,1]
};

// volume of the unit D-ball times r^D, in logs (gaussian.f90:36-37, utils.F90:754-760)
__device__ __forceinline__ double pc_log_ball(double r, int D, double log_vn)
{
    return (double)D * log(r) + log_vn;      // log( r^D V_D )
}
