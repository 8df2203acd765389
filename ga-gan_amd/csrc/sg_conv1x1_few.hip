// 1x1 convolutions with at most 4 channels on one side, 16-bit tensors: to-RGB (C -> 3), from-RGB (3 -> C), their
// data gradients and their weight gradients.  At 3 flop per byte these are pure streaming problems - the matrix
// pipeline of the general kernel (LDS staging, 16-wide channel tiles that are 13/16 padding) only gets in the way -
// so the forward / data-gradient forms run on the vector units with every lane owning 8 consecutive pixels
// (16-byte accesses, a wave reads or writes 1 KB contiguous per channel), and the weight gradient - a contraction
// over ALL pixels - feeds NCHW packets straight from global memory into v_mfma_f32_32x32x16, whose K index can be
// any 8 consecutive pixels of a channel row: no LDS, no transposition, the 32 x 32 sums live in the accumulators.
//
//   few_out :  y[n, co <= 4, p] = epilogue(sum_ci w[co, ci] s[n, ci] x[n, ci, p])                      (to-RGB forward)
//   few_in  :  y[n, co, p] = epilogue(out_scale[n, co] sum_{ci <= 4} w[co, ci] s[n, ci] x[n, ci, p] m)  (from-RGB forward;
//              to-RGB data gradient, with the clamp / activation mask m of the saved output applied to x as it is
//              loaded, and with the style-gradient reduction sum_p acc * reduce_x as a by-product)
//   wgrad   :  dw[cg, cx] = sum_{n, p} g[n, cg, p] m gs[n, cg] x[n, cx, p] xs[n, cx], min(cg, cx) <= 4
//
// sg_conv2d / sg_conv2d_wgrad dispatch here.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <class T> struct Pair;
template <> struct Pair<__hip_bfloat16> {
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const bf16x2_t v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Pair<__half> {
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const f16x2_t v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

template <class T> __device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = Pair<T>::lo(v.x); f[1] = Pair<T>::hi(v.x); f[2] = Pair<T>::lo(v.y); f[3] = Pair<T>::hi(v.y);
    f[4] = Pair<T>::lo(v.z); f[5] = Pair<T>::hi(v.z); f[6] = Pair<T>::lo(v.w); f[7] = Pair<T>::hi(v.w);
}
template <class T> __device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(Pair<T>::pack(f[0], f[1]), Pair<T>::pack(f[2], f[3]), Pair<T>::pack(f[4], f[5]), Pair<T>::pack(f[6], f[7]));
}
// x * m for 8 elements: m = 1 where the saved output is positive, else slope; 0 where it reached the clamp bound
template <class T> __device__ __forceinline__ void mask8(float (&x)[8], const uint4& saved, float slope, float bound) {
    float y[8];
    unpack8<T>(saved, y);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float f = (y[j] > 0.f) ? 1.f : slope;
        f = (fabsf(y[j]) < bound) ? f : 0.f;
        x[j] *= f;
    }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Four per-lane values summed over the wave and added to dst[0..3] in LDS: the first two exchange steps halve the number
// of values a lane carries (a lane keeps the pair, then the one, that its bits 5 and 4 name and hands the rest to its
// partner), so seven exchanges do what four separate wave sums take twenty-four for.  All 64 lanes must be active.
__device__ __forceinline__ void wave_sum4_to_lds(float r0, float r1, float r2, float r3, float* dst, int lane) {
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0;
    float k0 = b5 ? r2 : r0, k1 = b5 ? r3 : r1;
    k0 += __shfl_xor(b5 ? r0 : r2, 32, 64);
    k1 += __shfl_xor(b5 ? r1 : r3, 32, 64);
    float k = b4 ? k1 : k0;
    k += __shfl_xor(b4 ? k0 : k1, 16, 64);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) k += __shfl_xor(k, off, 64);
    if ((lane & 15) == 0) atomicAdd(dst + 2 * (lane >> 5) + ((lane >> 4) & 1), k);
}

struct FewArgs {
    const void* x; const void* mask_y; const void* w; void* y;
    const float* in_scale; const float* out_scale; const void* bias; const void* reduce_x; float* reduce_out;
    const void* addend;           // few_in only: a tensor of y's shape added to the rounded result (or null)
    int n, ci, co;
    int packets;                  // 8-pixel packets per plane
    float slope, gain, hi, m_slope, m_hi;
};

constexpr int kFewThreads = 256;
constexpr int kFewOutMaxCi = 512;
constexpr int kReduceRows = 8;    // partial rows of the reduction: waves add into row (wave index mod 8)

// ---- few_out: every lane 8 pixels, loop over the input channels ---------------------------------------------------
template <class T, int CO>
__global__ void __launch_bounds__(kFewThreads) conv1x1_few_out_kernel(FewArgs g) {
    __shared__ float s_w[kFewOutMaxCi][CO];
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int e = tid; e < g.ci * CO; e += kFewThreads) {
        const int c = e / CO, o = e - c * CO;
        const float s = g.in_scale ? g.in_scale[(int64_t)n * g.ci + c] : 1.f;
        s_w[c][o] = Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.w)[o * g.ci + c])) * s;
    }
    __syncthreads();
    const int pk = blockIdx.x * kFewThreads + tid;
    if (pk >= g.packets) return;
    const int64_t plane = (int64_t)g.packets * 8;
    const T* __restrict__ x = (const T*)g.x + (int64_t)n * g.ci * plane + (int64_t)pk * 8;
    float acc[CO][8];
#pragma unroll
    for (int o = 0; o < CO; o++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[o][j] = 0.f;
    int c = 0;
    for (; c + 8 <= g.ci; c += 8) {           // 8 packets in flight per lane
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const uint4*>(x + (c + u) * plane);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float f[8];
            unpack8<T>(v[u], f);
#pragma unroll
            for (int o = 0; o < CO; o++) {
                const float wv = s_w[c + u][o];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[o][j] = __builtin_fmaf(f[j], wv, acc[o][j]);
            }
        }
    }
    for (; c < g.ci; c++) {
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + c * plane), f);
#pragma unroll
        for (int o = 0; o < CO; o++) {
            const float wv = s_w[c][o];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[o][j] = __builtin_fmaf(f[j], wv, acc[o][j]);
        }
    }
    T* __restrict__ y = (T*)g.y + (int64_t)n * CO * plane + (int64_t)pk * 8;
#pragma unroll
    for (int o = 0; o < CO; o++) {
        const float sc = (g.out_scale ? g.out_scale[(int64_t)n * CO + o] : 1.f) * g.gain;
        const float bs = (g.bias ? Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.bias)[o])) : 0.f) * g.gain;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float z = __builtin_fmaf(acc[o][j], sc, bs);
            z = __builtin_fmaxf(z, z * g.slope);
            acc[o][j] = __builtin_amdgcn_fmed3f(z, -g.hi, g.hi);
        }
        *reinterpret_cast<uint4*>(y + o * plane) = pack8<T>(acc[o]);
    }
}

// ---- few_in: every lane 8 pixels (times several packets), loop over the output channels ---------------------------------
// CO_R: compile-time bound of the output channels when the reduction is wanted (its partial sums live in registers)
// ADD: y = round(round(result) + addend) - the arithmetic of a separate sum of two 16-bit tensors (the data gradient of
// a to-RGB layer meets the other gradient of its input here instead of in a pass of its own)
template <class T, int CI, bool MASK, int CO_R, bool ADD>
__global__ void __launch_bounds__(kFewThreads) conv1x1_few_in_kernel(FewArgs g) {
    extern __shared__ float s_few[];            // [co][CI] weights x in_scale, then [co] out_scale x gain, [co] bias x gain, [CO_R] sums
    float* s_w = s_few;
    float* s_sc = s_few + g.co * CI;
    float* s_bs = s_sc + g.co;
    float* s_red = s_bs + g.co;
    const int n = blockIdx.y, tid = threadIdx.x;
    if (tid < CO_R) s_red[tid] = 0.f;
    for (int e = tid; e < g.co * CI; e += kFewThreads) {
        const int o = e / CI, c = e - o * CI;
        const float s = g.in_scale ? g.in_scale[(int64_t)n * CI + c] : 1.f;
        s_w[e] = Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.w)[o * CI + c])) * s;
    }
    for (int o = tid; o < g.co; o += kFewThreads) {
        s_sc[o] = (g.out_scale ? g.out_scale[(int64_t)n * g.co + o] : 1.f) * g.gain;
        s_bs[o] = (g.bias ? Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.bias)[o])) : 0.f) * g.gain;
    }
    __syncthreads();
    const int64_t plane = (int64_t)g.packets * 8;
    const T* __restrict__ xn = (const T*)g.x + (int64_t)n * CI * plane;
    const T* __restrict__ mn = MASK ? (const T*)g.mask_y + (int64_t)n * CI * plane : nullptr;
    const T* __restrict__ rn = (CO_R > 0) ? (const T*)g.reduce_x + (int64_t)n * g.co * plane : nullptr;
    T* __restrict__ yn = (T*)g.y + (int64_t)n * g.co * plane;
    const T* __restrict__ an = ADD ? (const T*)g.addend + (int64_t)n * g.co * plane : nullptr;
    // (The reduction's sums live in LDS, not in CO_R registers of every lane: with them the kernel took all 256 registers
    // and ran one wave per SIMD - 2.8 TB/s on a kernel that only streams.)
    for (int base = blockIdx.x * kFewThreads; base < g.packets; base += gridDim.x * kFewThreads) {   // (uniform: the sums exchange across the wave)
        const bool valid = base + tid < g.packets;
        const int pk = valid ? base + tid : g.packets - 1;      // a lane beyond the plane repeats the last packet and drops the result
        float xin[CI][8];
#pragma unroll
        for (int c = 0; c < CI; c++) {
            unpack8<T>(*reinterpret_cast<const uint4*>(xn + c * plane + (int64_t)pk * 8), xin[c]);
            if constexpr (MASK) mask8<T>(xin[c], *reinterpret_cast<const uint4*>(mn + c * plane + (int64_t)pk * 8), g.m_slope, g.m_hi);
        }
        auto one = [&](int o, const uint4& rxv, const uint4& addv) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll
            for (int c = 0; c < CI; c++) {
                const float wv = s_w[o * CI + c];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = __builtin_fmaf(xin[c][j], wv, acc[j]);
            }
            float r = 0.f;
            if constexpr (CO_R > 0) {
                float f[8];
                unpack8<T>(rxv, f);
#pragma unroll
                for (int j = 0; j < 8; j++) r = __builtin_fmaf(acc[j], f[j], r);
            }
            const float sc = s_sc[o], bs = s_bs[o];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float z = __builtin_fmaf(acc[j], sc, bs);
                z = __builtin_fmaxf(z, z * g.slope);
                acc[j] = __builtin_amdgcn_fmed3f(z, -g.hi, g.hi);
            }
            uint4 out = pack8<T>(acc);
            if constexpr (ADD) {
                float a[8], b[8];
                unpack8<T>(out, a);
                unpack8<T>(addv, b);
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] += b[j];
                out = pack8<T>(a);
            }
            if (valid) *reinterpret_cast<uint4*>(yn + o * plane + (int64_t)pk * 8) = out;
            return valid ? r : 0.f;
        };
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        if constexpr (CO_R > 0) {
            // the reduction (and addend) packets of the next four channels are requested before this group's arithmetic and stores
            uint4 rxv[2][4], adv[2][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                rxv[0][u] = *reinterpret_cast<const uint4*>(rn + u * plane + (int64_t)pk * 8);
                adv[0][u] = ADD ? *reinterpret_cast<const uint4*>(an + u * plane + (int64_t)pk * 8) : zero4;
            }
#pragma unroll
            for (int o0 = 0; o0 < CO_R; o0 += 4) {
                if (o0 + 4 < CO_R) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        rxv[((o0 >> 2) + 1) & 1][u] = *reinterpret_cast<const uint4*>(rn + (o0 + 4 + u) * plane + (int64_t)pk * 8);
                        adv[((o0 >> 2) + 1) & 1][u] = ADD ? *reinterpret_cast<const uint4*>(an + (o0 + 4 + u) * plane + (int64_t)pk * 8) : zero4;
                    }
                }
                float r[4];
#pragma unroll
                for (int u = 0; u < 4; u++) r[u] = one(o0 + u, rxv[(o0 >> 2) & 1][u], adv[(o0 >> 2) & 1][u]);
                wave_sum4_to_lds(r[0], r[1], r[2], r[3], s_red + o0, tid & 63);
            }
        } else if constexpr (ADD) {
            // groups of four output channels (co is a multiple of 4 with an addend, checked by the dispatcher), the next
            // group's addend packets requested before this group's arithmetic and stores
            uint4 adv[2][4];
#pragma unroll
            for (int u = 0; u < 4; u++) adv[0][u] = *reinterpret_cast<const uint4*>(an + u * plane + (int64_t)pk * 8);
            for (int o0 = 0; o0 < g.co; o0 += 8) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int ob = o0 + 4 * h;
                    if (ob >= g.co) break;
                    const int nb = min(ob + 4, g.co - 4);      // (the last group requests itself again: no branch around the loads)
#pragma unroll
                    for (int u = 0; u < 4; u++) adv[(h + 1) & 1][u] = *reinterpret_cast<const uint4*>(an + (nb + u) * plane + (int64_t)pk * 8);
#pragma unroll
                    for (int u = 0; u < 4; u++) one(ob + u, zero4, adv[h][u]);
                }
            }
        } else {
            for (int o = 0; o < g.co; o++) one(o, zero4, zero4);
        }
    }
    if constexpr (CO_R > 0) {
        __syncthreads();
        if (tid < CO_R) atomicAdd(g.reduce_out + ((int64_t)n * kReduceRows + blockIdx.x % kReduceRows) * CO_R + tid, s_red[tid]);
    }
}

// ---- weight gradient: 32 "many" channels x up to 4 "few" channels, contraction over the pixels on the matrix cores -------
struct FewWgradArgs {
    const void* many; const void* few; const void* mask_y; float* dw;
    const float* many_scale; const float* few_scale;
    int n, cm, cf;                // channels of the many / few side
    int packets;                  // 8-pixel packets per plane
    int chunk;                    // packets per wave
    int chunks;                   // chunks per plane
    int many_is_g;                // the many side is the gradient g (dw[cg = many][cx = few]) or x (dw[cg = few][cx = many])
    int mask_on_many;             // which operand the mask applies to (the gradient side)
    float m_slope, m_hi;
};

// Each wave owns (sample, chunk of pixels, 32 many-side channels) and walks the chunk 128 pixels at a time.  The
// packets of a step are requested row by row - 16 lanes read 256 contiguous bytes of one channel plane, four planes
// per instruction - masked in registers, and parked in a wave-private LDS image [channel][128 pixels] (272-byte rows),
// from which every lane takes the fragments of its own channel: 8 consecutive pixels are 16 contiguous bytes.  (Read
// straight from memory, a lane per channel plane, the same fragments cost 32 scattered 32-byte requests per
// instruction and ran at 1.3 - 2.5 TB/s.)  The next step's packets are in flight while this one is multiplied.
constexpr int kWgSP = 16;                         // packets (of 8 pixels) per step
constexpr int kWgPitch = kWgSP * 16 + 16;         // bytes per LDS row
constexpr int kWgRows = 32 + 4;                   // many-side rows, few-side rows
constexpr int kWgWaveBytes = kWgRows * kWgPitch;  // 9792

template <class T, bool MASK>
__global__ void __launch_bounds__(256) wgrad1x1_few_kernel(FewWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_wg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lc = lane & 31, hh = lane >> 5;
    int unit = blockIdx.x * 4 + wave;                      // (sample, chunk) of this wave
    const int mblk = blockIdx.y;                           // group of 32 many-side channels
    if (unit >= g.n * g.chunks) return;
    const int n = unit / g.chunks, chunk = unit - n * g.chunks;
    const int64_t plane = (int64_t)g.packets * 8;
    unsigned char* s_many = s_wg + wave * kWgWaveBytes;
    unsigned char* s_few = s_many + 32 * kWgPitch;
    const int pk0 = chunk * g.chunk, pk1 = min(g.packets, pk0 + g.chunk);

    // staging plan: item k of the many side is channel 4 k + (lane >> 4), packet lane & 15; the few side is one item
    const int srow = lane >> 4, spk = lane & 15;
    const T* __restrict__ pm = (const T*)g.many + ((int64_t)n * g.cm + mblk * 32) * plane;
    const T* __restrict__ pf = (const T*)g.few + (int64_t)n * g.cf * plane;
    const T* __restrict__ pmask_m = nullptr;
    const T* __restrict__ pmask_f = nullptr;
    if constexpr (MASK) {
        if (g.mask_on_many) pmask_m = (const T*)g.mask_y + ((int64_t)n * g.cm + mblk * 32) * plane;
        else pmask_f = (const T*)g.mask_y + (int64_t)n * g.cf * plane;
    }
    const int rows_m = min(32, g.cm - mblk * 32);
    uint4 va[8], vm[MASK ? 8 : 1], vf, vfm;
    auto load = [&](int pp) __attribute__((always_inline)) {
        const int p = pp + spk;
        const bool p_ok = p < pk1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = 4 * k + srow;
            const bool ok = p_ok && row < rows_m;             // (unconditional loads: a packet outside is read from zeros)
            va[k] = load16_or_zero(ok, pm + row * plane + (int64_t)p * 8);
            if constexpr (MASK) vm[k] = load16_or_zero(ok && pmask_m != nullptr, pmask_m + row * plane + (int64_t)p * 8);
        }
        const bool f_row = p_ok && srow < g.cf;
        vf = load16_or_zero(f_row, pf + srow * plane + (int64_t)p * 8);
        vfm = make_uint4(0, 0, 0, 0);
        if constexpr (MASK) vfm = load16_or_zero(f_row && pmask_f != nullptr, pmask_f + srow * plane + (int64_t)p * 8);
    };
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if constexpr (MASK) {
                if (pmask_m) {
                    float f[8];
                    unpack8<T>(va[k], f);
                    mask8<T>(f, vm[k], g.m_slope, g.m_hi);
                    va[k] = pack8<T>(f);
                }
            }
            *reinterpret_cast<uint4*>(s_many + (4 * k + srow) * kWgPitch + spk * 16) = va[k];
        }
        if constexpr (MASK) {
            if (pmask_f) {
                float f[8];
                unpack8<T>(vf, f);
                mask8<T>(f, vfm, g.m_slope, g.m_hi);
                vf = pack8<T>(f);
            }
        }
        *reinterpret_cast<uint4*>(s_few + srow * kWgPitch + spk * 16) = vf;
    };

    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const unsigned char* a_src = s_many + lc * kWgPitch + hh * 16;
    const unsigned char* b_src = s_few + (lc & 3) * kWgPitch + hh * 16;
    const bool f_ok = lc < g.cf;
    load(pk0);
    for (int pp = pk0; pp < pk1; pp += kWgSP) {      // uniform trip count: the matrix instruction runs on all lanes
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the previous step are done
        park();
        if (pp + kWgSP < pk1) load(pp + kWgSP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its writes have landed (the image is the wave's own)
        // K step = 16 pixels: lane half hh takes packet 2 s + hh of both operands (the K order is free as long as both agree)
#pragma unroll
        for (int s2 = 0; s2 < kWgSP / 2; s2++) {
            const uint4 a = *reinterpret_cast<const uint4*>(a_src + s2 * 32);
            uint4 b = *reinterpret_cast<const uint4*>(b_src + s2 * 32);
            if (!f_ok) b = make_uint4(0, 0, 0, 0);
            Pair<T>::mma(a, b, acc);      // rows: many-side channel, columns: few-side channel
        }
    }
    // accumulator register e of lane (column lc, half hh): row (e & 3) + 8 (e >> 2) + 4 hh
    if (!f_ok) return;
    const float fs = g.few_scale ? g.few_scale[(int64_t)n * g.cf + lc] : 1.f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int row = mblk * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (row >= g.cm) continue;
        const float ms = g.many_scale ? g.many_scale[(int64_t)n * g.cm + row] : 1.f;
        const float v = acc[e] * fs * ms;
        float* dst = g.many_is_g ? g.dw + (int64_t)row * g.cf + lc : g.dw + (int64_t)lc * g.cm + row;
        atomicAdd(dst, v);
    }
}

bool common_ok(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 1 || p.in_step != 1 || p.out_step != 1) return false;
    const sg_conv2d_phase& ph = p.phase[0];
    if (ph.nky != 1 || ph.nkx != 1 || ph.oy0 != 0 || ph.ox0 != 0 || ph.dy0 != 0 || ph.dx0 != 0 || ph.w_offset != 0) return false;
    if (p.oh != p.h || p.ow != p.w_in) return false;
    if (((int64_t)p.h * p.w_in) % 8 != 0 || (int64_t)p.h * p.w_in < 4096) return false;   // small planes: launch-bound either way
    if (p.noise) return false;
    if (!(p.gain > 0.f) || (p.act == 3 && !(p.alpha >= 0.f && p.alpha <= 1.f))) return false;
    if (!aligned16(p.x) || !aligned16(p.y) || (p.mask_y && !aligned16(p.mask_y)) || !aligned16(p.reduce_x)) return false;
    return (int64_t)p.h * p.w_in / 8 <= INT32_MAX;
}

FewArgs few_args(const sg_conv2d_params& p) {
    FewArgs g;
    g.x = p.x; g.mask_y = p.mask_y; g.w = p.w; g.y = p.y;
    g.in_scale = p.in_scale; g.out_scale = p.out_scale; g.bias = p.bias;
    g.reduce_x = p.reduce_out ? p.reduce_x : nullptr; g.reduce_out = p.reduce_x ? p.reduce_out : nullptr;
    g.addend = p.addend;
    g.n = p.n; g.ci = p.ci; g.co = p.co;
    g.packets = (int)((int64_t)p.h * p.w_in / 8);
    g.slope = (p.act == 3) ? p.alpha : 1.f;
    g.gain = p.gain;
    g.hi = (p.clamp >= 0.f) ? p.clamp : __builtin_inff();
    g.m_slope = p.mask_alpha;
    g.m_hi = (p.mask_clamp >= 0.f) ? p.mask_clamp : __builtin_inff();
    return g;
}

template <class T, int CO>
int launch_few_out(const FewArgs& g, hipStream_t stream) {
    hipLaunchKernelGGL((conv1x1_few_out_kernel<T, CO>), dim3((unsigned)((g.packets + kFewThreads - 1) / kFewThreads), (unsigned)g.n),
                       dim3(kFewThreads), 0, stream, g);
    return check_launch("sg_conv2d");
}

template <class T, int CI, bool MASK>
int launch_few_in(const FewArgs& g, hipStream_t stream) {
    const size_t lds = ((size_t)g.co * (CI + 2) + 64) * sizeof(float);
    // a lane keeps its reduction sums over several packets; without a reduction one packet per lane
    const int per_lane = g.reduce_out ? 8 : 1;
    const dim3 grid((unsigned)std::max(1, (g.packets + kFewThreads * per_lane - 1) / (kFewThreads * per_lane)), (unsigned)g.n);
    if constexpr (CI == 3) {      // (the addend form exists for three channels: the to-RGB layers)
        if (g.addend) {
            if (!g.reduce_out) hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 0, true>), grid, dim3(kFewThreads), lds, stream, g);
            else if (g.co == 32) hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 32, true>), grid, dim3(kFewThreads), lds, stream, g);
            else hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 64, true>), grid, dim3(kFewThreads), lds, stream, g);
            return check_launch("sg_conv2d");
        }
    }
    if (g.reduce_out) {
        if (g.co == 32) hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 32, false>), grid, dim3(kFewThreads), lds, stream, g);
        else hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 64, false>), grid, dim3(kFewThreads), lds, stream, g);
    } else {
        hipLaunchKernelGGL((conv1x1_few_in_kernel<T, CI, MASK, 0, false>), grid, dim3(kFewThreads), lds, stream, g);
    }
    return check_launch("sg_conv2d");
}

template <class T>
int launch_few(const sg_conv2d_params& p, hipStream_t stream) {
    const FewArgs g = few_args(p);
    if (p.co <= 4 && p.ci > 4) {
        switch (p.co) {
            case 1: return launch_few_out<T, 1>(g, stream);
            case 2: return launch_few_out<T, 2>(g, stream);
            case 3: return launch_few_out<T, 3>(g, stream);
            default: return launch_few_out<T, 4>(g, stream);
        }
    }
    const bool mask = p.mask_y != nullptr;
    switch (p.ci) {
        case 1: return mask ? launch_few_in<T, 1, true>(g, stream) : launch_few_in<T, 1, false>(g, stream);
        case 2: return mask ? launch_few_in<T, 2, true>(g, stream) : launch_few_in<T, 2, false>(g, stream);
        case 3: return mask ? launch_few_in<T, 3, true>(g, stream) : launch_few_in<T, 3, false>(g, stream);
        default: return mask ? launch_few_in<T, 4, true>(g, stream) : launch_few_in<T, 4, false>(g, stream);
    }
}

}  // namespace

bool conv1x1_few_eligible(const sg_conv2d_params& p) {
    if (!common_ok(p)) return false;
    if (p.co <= 4 && p.ci > 4) {                                   // few_out
        return p.ci <= kFewOutMaxCi && !p.mask_y && !p.reduce_x && !p.addend;
    }
    if (p.ci <= 4 && p.co > 4) {                                   // few_in
        if (p.co > 1024) return false;
        // (decided by reduce_x, which callers set before they size reduce_out with sg_conv2d_reduce_rows)
        if (p.reduce_x && p.co != 32 && p.co != 64) return false;
        if (p.addend && (p.ci != 3 || p.co % 4 != 0 || !aligned16(p.addend))) return false;
        return true;
    }
    return false;
}

int conv1x1_few_reduce_rows(const sg_conv2d_params&) { return kReduceRows; }

int conv1x1_few_launch(const sg_conv2d_params& p, hipStream_t stream) {
    return (p.dtype == SG_BF16) ? launch_few<__hip_bfloat16>(p, stream) : launch_few<__half>(p, stream);
}

bool wgrad1x1_few_eligible(const sg_conv2d_wgrad_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.nky != 1 || p.nkx != 1 || p.step != 1 || p.dy0 != 0 || p.dx0 != 0) return false;
    if (p.gh != p.h || p.gw != p.w_in) return false;
    if (std::min(p.cx, p.cg) > 4 || std::max(p.cx, p.cg) < 8) return false;
    if (((int64_t)p.h * p.w_in) % 8 != 0 || (int64_t)p.h * p.w_in < 4096 || (int64_t)p.h * p.w_in / 8 > INT32_MAX) return false;
    return aligned16(p.x) && aligned16(p.g) && (!p.mask_y || aligned16(p.mask_y));
}

int wgrad1x1_few_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    FewWgradArgs g;
    const bool many_is_g = p.cg > p.cx;
    g.many = many_is_g ? p.g : p.x;
    g.few = many_is_g ? p.x : p.g;
    g.many_scale = many_is_g ? p.g_scale : p.x_scale;
    g.few_scale = many_is_g ? p.x_scale : p.g_scale;
    g.mask_y = p.mask_y;
    g.dw = p.dw;
    g.n = p.n;
    g.cm = std::max(p.cx, p.cg);
    g.cf = std::min(p.cx, p.cg);
    g.packets = (int)((int64_t)p.h * p.w_in / 8);
    g.chunk = 2048;                                   // 16384 pixels per wave
    // enough waves to fill the chip, no more: every wave ends with 16 x cf atomic adds to the same few addresses
    const int64_t mblocks = (std::max(p.cx, p.cg) + 31) / 32;
    while (g.chunk > 64 && (int64_t)p.n * ((g.packets + g.chunk - 1) / g.chunk) * mblocks < 2 * 1024) g.chunk /= 2;
    g.chunks = (g.packets + g.chunk - 1) / g.chunk;
    g.many_is_g = many_is_g ? 1 : 0;
    g.mask_on_many = many_is_g ? 1 : 0;               // the mask belongs to the gradient operand
    g.m_slope = p.mask_alpha;
    g.m_hi = (p.mask_clamp >= 0.f) ? p.mask_clamp : __builtin_inff();
    const int64_t units = (int64_t)p.n * g.chunks;
    const dim3 grid((unsigned)((units + 3) / 4), (unsigned)((g.cm + 31) / 32));
    if (p.dtype == SG_BF16) {
        if (p.mask_y) hipLaunchKernelGGL((wgrad1x1_few_kernel<__hip_bfloat16, true>), grid, dim3(256), 4 * kWgWaveBytes, stream, g);
        else hipLaunchKernelGGL((wgrad1x1_few_kernel<__hip_bfloat16, false>), grid, dim3(256), 4 * kWgWaveBytes, stream, g);
    } else {
        if (p.mask_y) hipLaunchKernelGGL((wgrad1x1_few_kernel<__half, true>), grid, dim3(256), 4 * kWgWaveBytes, stream, g);
        else hipLaunchKernelGGL((wgrad1x1_few_kernel<__half, false>), grid, dim3(256), 4 * kWgWaveBytes, stream, g);
    }
    return check_launch("sg_conv2d_wgrad");
}

}  // namespace sg
