// Shared helpers for the gfx950 kernels: storage <-> compute type traits, 16-byte
// vector access, error reporting across the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include <cstdarg>
#include <cstdio>

#include "sg_ops.h"

namespace sg {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error text, one slot per host thread ---------------------------------
char* error_slot();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
// Opt a kernel in to 160 KB of dynamic LDS, once per kernel and device (not a stream operation: kept out of the
// launch path proper so that launches can be captured into graphs).  Returns hipSuccess or the error.
hipError_t allow_large_lds(const void* kernel);

#define SG_REQUIRE(cond, code, ...)                      \
    do {                                                 \
        if (!(cond)) return ::sg::fail((code), __VA_ARGS__); \
    } while (0)

// ---- element types ----------------------------------------------------------
// Storage type T -> arithmetic type acc_t (float for f32/f16/bf16, double for f64).
template <class T> struct Elem;
template <> struct Elem<float> {
    using acc_t = float;
    static __device__ __forceinline__ float load(float v) { return v; }
    static __device__ __forceinline__ float store(float v) { return v; }
};
template <> struct Elem<double> {
    using acc_t = double;
    static __device__ __forceinline__ double load(double v) { return v; }
    static __device__ __forceinline__ double store(double v) { return v; }
};
template <> struct Elem<__half> {
    using acc_t = float;
    static __device__ __forceinline__ float load(__half v) { return __half2float(v); }
    static __device__ __forceinline__ __half store(float v) { return __float2half(v); }
};
template <> struct Elem<__hip_bfloat16> {
    using acc_t = float;
    static __device__ __forceinline__ float load(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ __hip_bfloat16 store(float v) { return __float2bfloat16(v); }
};

// Loads that every lane issues: a piece outside the tensor is read from a packet of zeros instead of being skipped.
// No branch then surrounds the load, the value's first use decides where the wave waits, and that wait is a counted
// one which leaves the loads issued after it in flight.  (A load inside a branch is waited for by `s_waitcnt vmcnt(0)`
// wherever the compiler cannot count what came after it - typically at once, when a conversion of the loaded value
// was moved into the branch - and that wait covers everything the wave has outstanding, prefetches included.)
static __device__ __attribute__((aligned(16))) const unsigned kZeroPacket[8] = {0, 0, 0, 0, 0, 0, 0, 0};
typedef unsigned sg_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned sg_u32x2_t __attribute__((ext_vector_type(2)));
// (both sources are addresses in global memory; said so - and loaded as a plain vector, not through uint4's copy
// constructor - or the choice between them makes the load a flat one)
template <class V> __device__ __forceinline__ V load_global_or_zero(bool ok, const void* src) {
    typedef const V __attribute__((address_space(1)))* Global;
    return *(ok ? (Global)src : (Global)(const void*)kZeroPacket);
}
__device__ __forceinline__ uint4 load16_or_zero(bool ok, const void* src) {
    const sg_u32x4_t v = load_global_or_zero<sg_u32x4_t>(ok, src); return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 load8_or_zero(bool ok, const void* src) {
    const sg_u32x2_t v = load_global_or_zero<sg_u32x2_t>(ok, src); return make_uint2(v.x, v.y);
}
__device__ __forceinline__ unsigned load4_or_zero(bool ok, const void* src) { return load_global_or_zero<unsigned>(ok, src); }
__device__ __forceinline__ unsigned short load2_or_zero(bool ok, const void* src) { return load_global_or_zero<unsigned short>(ok, src); }

// A 16-byte packet of T, moved with one dwordx4 access.
template <class T> struct alignas(16) Pack {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

template <class T> __device__ __forceinline__ Pack<T> load_pack(const T* p) {
    return *reinterpret_cast<const Pack<T>*>(p);
}
template <class T> __device__ __forceinline__ void store_pack(T* p, const Pack<T>& v) {
    *reinterpret_cast<Pack<T>*>(p) = v;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Dispatch a callable templated on the storage type.
#define SG_DISPATCH_DTYPE(dtype, ALLOW_F64, ...)                                   \
    switch (dtype) {                                                               \
        case SG_F32: { using T = float; __VA_ARGS__; break; }                      \
        case SG_F16: { using T = __half; __VA_ARGS__; break; }                     \
        case SG_BF16: { using T = __hip_bfloat16; __VA_ARGS__; break; }            \
        case SG_F64:                                                               \
            if (ALLOW_F64) { using T = double; __VA_ARGS__; break; }               \
            return ::sg::fail(SG_ERR_UNSUPPORTED, "float64 is not supported here"); \
        default: return ::sg::fail(SG_ERR_INVALID_ARGUMENT, "unknown dtype %d", (int)(dtype)); \
    }

}  // namespace sg
