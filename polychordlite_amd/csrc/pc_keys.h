// pc_keys.h -- order-preserving integer keys for logL and the log-space helpers shared by the
// single-cluster contraction kernels (pc_fast.hip, pc_par.hip).
#pragma once
#include "pc_state.h"

#define NEGBIG (-1e300)

// order-preserving map double -> uint64 (no NaNs in the live set): the serial pass compares integers
__device__ __forceinline__ unsigned long long d2key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key2d(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{   // the value is identical in every lane: move it to scalar registers
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
#define KEY_HUGE 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ double lae2(double a, double b)
{   // logaddexp that tolerates the NEGBIG neutral element
    const double m = fmax(a, b), d = fmin(a, b) - m;
    return m + log(1.0 + exp(d));
}

