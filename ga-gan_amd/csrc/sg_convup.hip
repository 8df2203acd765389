// Transposed 3x3 convolution with stride 2 of 16-bit tensors (the up-sampling layers of the
// generator and the data gradient of the discriminator's down-sampling layers) as ONE unit-stride
// contraction: output pixel (2a + py, 2b + px) only depends on the 2 x 2 input window
// {a-1, a} x {b-1, b}, through 4 / 2 / 2 / 1 of the 9 weight taps for the phases (py, px) =
// (0,0) / (0,1) / (1,0) / (1,1).  GEMM view: M = input-grid pixels (a, b), N = phase x output
// channel, K = input channels x window taps - the 9 non-zero (phase, tap) combinations only.
//
// Against the general kernel's polyphase form this stages the input tile once for all four phases
// (not once per phase), and a lane ends up with 4 consecutive b of both px phases of a channel,
// i.e. 8 consecutive output pixels: one 16-byte store instead of 8 strided 2-byte stores.
//
// Workgroup: 4 waves, tile of 8 x 32 input-grid pixels (16 x 64 outputs) x 32 output channels x 4
// phases; K in chunks of 32 channels with the next chunk's packets (input and weights) loaded into
// registers during the multiply phase; swizzled 64-byte LDS rows as in the 3x3 kernel.  Input
// modulation (in_scale) is applied while staging; there is no output epilogue (the layer's epilogue
// runs in the FIR that follows) - sg_conv2d keeps the general kernel for calls that ask for one.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <type_traits>
#include <utility>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TW = 32, TH = 8;            // input-grid tile
constexpr int KCH = 32;                   // input channels per chunk
constexpr int BCO = 32;                   // output channels per workgroup (x 4 phases = 128 GEMM columns)
constexpr int kRow = 64;
constexpr int LH = TH + 1;                // footprint: rows a0-1 .. a0+7, columns b0-1 .. b0+31
constexpr int NV = 5;                     // packets b0-8 .. b0+31
constexpr int LWP = NV * 8;
constexpr int PX_COL = 7;                 // column of input b0-1 inside the staged row
constexpr int kInBytes = LH * LWP * kRow;         // 23040
constexpr int kCombos = 9;
constexpr int kWBytes = kCombos * BCO * kRow;     // 18432
constexpr int ITEMS = LH * NV;                    // 45 (row, packet) items per channel pair
constexpr int IN_STEPS = (ITEMS + 15) / 16;       // 3

// (phase, tap) combinations: phase = 2 py + px; (ty, tx) index the phase's own tap grid in the packed weights;
// (dy, dx) in {-1, 0} is the input offset of the tap.
struct Combo { int phase, ty, tx, dy, dx; };
template <class T> struct Mma16;
template <> struct Mma16<__hip_bfloat16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<__half> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ int swz(int row, int byte) {
    return row * kRow + ((((byte >> 4) ^ (row >> 1)) & 3) << 4) + (byte & 15);
}
// Two 16-bit elements in a dword <-> two floats (one conversion instruction for the pair).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <class T> struct Pair;
template <> struct Pair<__hip_bfloat16> {
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const bf16x2_t v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    }
};
template <> struct Pair<__half> {
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const f16x2_t v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
};
template <class T> __device__ __forceinline__ unsigned pack16(float a, float b) { return Pair<T>::pack(a, b); }

// Buffer addressing: a byte offset at or beyond the descriptor's bound reads zeros (packets outside the image, channels
// beyond the last): decided once per workgroup, no branch in the K loop.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOutside = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 load16(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant in every
// iteration's own code, so that register arrays indexed by it are split into registers.
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct ConvUpArgs {
    const void* x; const void* w; void* y;
    const float* in_scale;
    int n, ci, co, h, wid, oh, ow;
    int w_offset0, w_offset1, w_offset2, w_offset3;   // start of each phase's taps in the packed weights (elements)
    int tiles_a, tiles_b, units;
    unsigned w_bytes;                                  // bytes of the packed weights
};

// WAVES = 8 (layers with more than 32 output channels): 64 output channels per workgroup - waves 4..7 multiply the
// same staged pixels with the second 32 channels; one such workgroup per CU, two waves per SIMD.  MOD: in_scale given.
template <class T, int WAVES, bool MOD>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) conv_up2_kernel(ConvUpArgs g) {
    constexpr int BCOT = BCO * (WAVES / 4);                                    // output channels per workgroup
    constexpr int SLOTS = 4 * WAVES, STEPS = (ITEMS + SLOTS - 1) / SLOTS;      // staging: item slots, items per lane
    constexpr bool LAST_PARTIAL = (ITEMS % SLOTS) != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_in = smem;
    unsigned char* s_w = smem + kInBytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3, half = tid >> 8;   // tile rows 2 wave, 2 wave + 1; channels 32 half ..
    const int lrow = lane & 15, lgrp = lane >> 4;

    // XCD k (workgroups k, k + 8, ...) takes the k-th contiguous eighth of the units: neighbouring tiles share an L2
    const int per_xcd = (g.units + 7) >> 3;
    int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= g.units) return;
    const int tb = unit % g.tiles_b; unit /= g.tiles_b;
    const int ta = unit % g.tiles_a; unit /= g.tiles_a;
    const int n = unit % g.n;
    const int m0 = (unit / g.n) * BCOT;
    const int a0 = ta * TH, b0 = tb * TW;
    const int plane = g.h * g.wid;
    const unsigned plane_bytes = (unsigned)plane * 2u;
    const T* __restrict__ xs = (const T*)g.x + (unsigned)(n * g.ci * plane);
    const T* __restrict__ w = (const T*)g.w;

    // ---- staging plan, decided once: which packets a lane moves (kOutside: outside the image), where they land -----
    const int cg = tid & 15, slot0 = tid >> 4;
    unsigned voff[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; q++) {
        const int item = slot0 + SLOTS * q;
        const int ry = item / NV, v = item - ry * NV;
        const int iy = a0 - 1 + ry, px0 = b0 - 8 + v * 8;
        const bool ok = item < ITEMS && iy >= 0 && iy < g.h && px0 >= 0 && px0 < g.wid;
        voff[q] = ok ? (unsigned)(2 * cg * plane + iy * g.wid + px0) * 2u : kOutside;
    }
    const bool last_item = !LAST_PARTIAL || slot0 + SLOTS * (STEPS - 1) < ITEMS;
    int st_addr[4];
#pragma unroll
    for (int jp = 0; jp < 4; jp++) st_addr[jp] = slot0 * 8 * kRow + swz(2 * jp, cg * 4);
    // weights of a chunk: waves 0-1 (of each half) take the even (phase, tap) combinations, waves 2-3 the odd ones; a lane
    // moves one 16-byte packet of one output channel per combination
    constexpr int WQ = 5;
    unsigned w_voff[WQ];
    int w_st[WQ];
    const int w_slot = tid & 3;
    {
        constexpr Combo kc[kCombos + 1] = {{0, 0, 0, -1, -1}, {0, 0, 1, -1, 0}, {0, 1, 0, 0, -1}, {0, 1, 1, 0, 0}, {1, 0, 0, -1, 0},
                                           {1, 1, 0, 0, 0}, {2, 0, 0, 0, -1}, {2, 0, 1, 0, 0}, {3, 0, 0, 0, 0}, {3, 0, 0, 0, 0}};
        constexpr int nkx[4] = {2, 1, 2, 1};
        const bool odd = (tid & 255) >= 128;
        const int o_local = ((tid >> 2) & 31) + 32 * half;
        const int o = m0 + o_local;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int combo = 2 * q + (odd ? 1 : 0);
            const int phase = odd ? kc[2 * q + 1].phase : kc[2 * q].phase;
            const int tap = odd ? kc[2 * q + 1].ty * nkx[kc[2 * q + 1].phase] + kc[2 * q + 1].tx
                                : kc[2 * q].ty * nkx[kc[2 * q].phase] + kc[2 * q].tx;
            const int w_off = phase == 0 ? g.w_offset0 : phase == 1 ? g.w_offset1 : phase == 2 ? g.w_offset2 : g.w_offset3;
            w_voff[q] = (combo < kCombos && o < g.co) ? (unsigned)(w_off + (tap * g.co + o) * g.ci + w_slot * 8) * 2u : kOutside;
            w_st[q] = (combo < kCombos) ? swz(combo * BCOT + o_local, w_slot * 16) : -1;
        }
    }

    uint4 va[STEPS], vb[STEPS], vw[WQ];
    float s0 = 1.f, s1 = 1.f;

    // The loads of a chunk (2 STEPS input packets, two modulation scales, WQ weight packets per lane): the first chunk's
    // as one batch, every other chunk's dealt over the four window offsets of the multiply phase before it - a batch in
    // front of the phase queues in the texture path (16 cycles per 16-byte load of a wave) and staggers the waves.
    rsrc_t d_x = make_rsrc(xs, 0u), d_s = make_rsrc(xs, 0u), d_w = make_rsrc(xs, 0u);
    bool slot_ok = false;
    auto fetch_begin = [&](int c0) {
        const int left = min(KCH, g.ci - c0);
        d_x = make_rsrc(xs + (size_t)c0 * plane, (unsigned)left * plane_bytes);
        if constexpr (MOD) d_s = make_rsrc(g.in_scale + (size_t)n * g.ci + c0, (unsigned)left * 4u);
        d_w = make_rsrc(w + c0, (unsigned)g.w_bytes);
        slot_ok = w_slot * 8 < left;
    };
    constexpr int kLoads = 2 * STEPS + WQ;
    auto fetch_part = [&](auto offset) __attribute__((always_inline)) {      // offset o = 0 .. 3: loads l with l * 4 / kLoads == o
        constexpr int o = decltype(offset)::value;
        static_for<kLoads>([&](auto load) __attribute__((always_inline)) {
            constexpr int l = decltype(load)::value;
            if constexpr ((l * 4) / kLoads == o) {
                if constexpr (l < 2 * STEPS) {
                    if constexpr (l & 1) vb[l >> 1] = load16(d_x, voff[l >> 1], plane_bytes);
                    else va[l >> 1] = load16(d_x, voff[l >> 1], 0);
                } else {
                    vw[l - 2 * STEPS] = load16(d_w, slot_ok ? w_voff[l - 2 * STEPS] : kOutside, 0);
                }
            }
        });
        if constexpr (MOD) {
            if constexpr (o == 1) s0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s, cg * 8, 0, 0));
            if constexpr (o == 2) s1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s, cg * 8, 4, 0));
        }
    };
    auto fetch = [&](int c0) {
        fetch_begin(c0);
        static_for<4>([&](auto o) __attribute__((always_inline)) { fetch_part(o); });
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            if (LAST_PARTIAL && q == STEPS - 1 && !last_item) continue;
            const unsigned wa[4] = {va[q].x, va[q].y, va[q].z, va[q].w};
            const unsigned wb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                unsigned e0, e1;
                if constexpr (MOD) {
                    e0 = Pair<T>::pack(Pair<T>::lo(wa[jp]) * s0, Pair<T>::lo(wb[jp]) * s1);
                    e1 = Pair<T>::pack(Pair<T>::hi(wa[jp]) * s0, Pair<T>::hi(wb[jp]) * s1);
                } else {
                    e0 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x05040100u);
                    e1 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x07060302u);
                }
                unsigned char* dst = s_in + st_addr[jp] + q * (SLOTS * 8 * kRow);
                *reinterpret_cast<unsigned*>(dst) = e0;
                *reinterpret_cast<unsigned*>(dst + kRow) = e1;
            }
        }
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (w_st[q] >= 0) *reinterpret_cast<uint4*>(s_w + w_st[q]) = vw[q];
    };

    // n-tile j of wave w: tile row 2w + (j >> 1), columns 16 (j & 1) .. +15; LDS row of input (a + dy, b + dx):
    // (a - a0 + 1 + dy) * LWP + PX_COL + 1 + (b - b0) + dx
    int brow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) brow[j] = (2 * wave + (j >> 1) + 1) * LWP + PX_COL + 1 + (j & 1) * 16 + lrow;
    const int a_lane = lrow * kRow + (((lgrp ^ (lrow >> 1)) & 3) << 4);

    f32x4_t acc[4][2][4];   // [phase][channel tile][pixel tile]: rows = pixels 4 lgrp + r, column = channel lrow
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[p][ct][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](auto with_fetch) __attribute__((always_inline)) {
        // window offset outermost: its 4 input fragments serve every (phase, tap) combination that reads it
        static_for<4>([&](auto offset) __attribute__((always_inline)) {   // o = 2 (dy + 1) + (dx + 1)
            constexpr int o = decltype(offset)::value;
            constexpr Combo kc[kCombos] = {{0, 0, 0, -1, -1}, {0, 0, 1, -1, 0}, {0, 1, 0, 0, -1}, {0, 1, 1, 0, 0}, {1, 0, 0, -1, 0},
                                           {1, 1, 0, 0, 0}, {2, 0, 0, 0, -1}, {2, 0, 1, 0, 0}, {3, 0, 0, 0, 0}};
            uint4 fb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = brow[j] + ((o >> 1) - 1) * LWP + (o & 1) - 1;
                fb[j] = *reinterpret_cast<const uint4*>(s_in + row * kRow + (((lgrp ^ (row >> 1)) & 3) << 4));
            }
            if constexpr (decltype(with_fetch)::value) fetch_part(offset);
#pragma unroll
            for (int combo = 0; combo < kCombos; combo++) {
                if (2 * (kc[combo].dy + 1) + kc[combo].dx + 1 != o) continue;
                const int p = kc[combo].phase;
                uint4 fa[2];
#pragma unroll
                for (int ct = 0; ct < 2; ct++) fa[ct] = *reinterpret_cast<const uint4*>(s_w + (combo * BCOT + 32 * half + ct * 16) * kRow + a_lane);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) Mma16<T>::run(fb[j], fa[ct], acc[p][ct][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    fetch(0);
    const int c_last = ((g.ci - 1) / KCH) * KCH;
    for (int c0 = 0; c0 < c_last; c0 += KCH) {
        __syncthreads();
        commit();
        __syncthreads();
        fetch_begin(c0 + KCH);
        multiply(std::true_type{});     // (the next chunk's loads go out between its window offsets)
    }
    __syncthreads();
    commit();
    __syncthreads();
    multiply(std::false_type{});

    // ---- store: both px phases of a channel interleave into 8 consecutive output pixels -----------------------
    T* __restrict__ y = (T*)g.y;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = a0 + 2 * wave + (j >> 1), b = b0 + (j & 1) * 16 + 4 * lgrp;
        const int ox = 2 * b;
        if (ox >= g.ow) continue;
#pragma unroll
        for (int py = 0; py < 2; py++) {
            const int oy = 2 * a + py;
            if (oy >= g.oh) continue;
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int ch = m0 + 32 * half + ct * 16 + lrow;
                if (ch >= g.co) continue;
                const f32x4_t e = acc[2 * py][ct][j], o = acc[2 * py + 1][ct][j];
                const uint4 out = make_uint4(pack16<T>(e[0], o[0]), pack16<T>(e[1], o[1]), pack16<T>(e[2], o[2]), pack16<T>(e[3], o[3]));
                T* dst = y + (unsigned)(((n * g.co + ch) * g.oh + oy) * g.ow + ox);
                if (ox + 8 <= g.ow) *reinterpret_cast<uint4*>(dst) = out;
                else {   // ragged last packet of a row
                    const unsigned words[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (ox + i < g.ow) dst[i] = __builtin_bit_cast(T, (unsigned short)(words[i >> 1] >> (16 * (i & 1))));
                }
            }
        }
    }
}

}  // namespace

bool conv_up2_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 4 || p.in_step != 1 || p.out_step != 2) return false;
    // the four phases of a 3x3, stride-2, unpadded transposed convolution, in (py, px) order
    static const int want[4][6] = {{0, 0, -1, -1, 2, 2}, {0, 1, -1, 0, 2, 1}, {1, 0, 0, -1, 1, 2}, {1, 1, 0, 0, 1, 1}};
    for (int i = 0; i < 4; i++) {
        const sg_conv2d_phase& ph = p.phase[i];
        if (ph.oy0 != want[i][0] || ph.ox0 != want[i][1] || ph.dy0 != want[i][2] || ph.dx0 != want[i][3] ||
            ph.nky != want[i][4] || ph.nkx != want[i][5] || ph.w_offset < 0 || ph.w_offset > INT32_MAX / 2)
            return false;
    }
    if (p.out_scale || p.bias || p.noise || p.reduce_out || p.act != 1 || p.gain != 1.f || p.clamp >= 0.f) return false;
    if (p.w_in % 8 != 0 || p.ow % 8 != 0 || p.ci % 8 != 0 || p.w_in < 16 || p.ci < KCH) return false;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.y)) return false;
    if ((int64_t)p.ci * p.h * p.w_in >= (int64_t)1 << 30 || (int64_t)p.co * p.ci >= (1 << 26)) return false;   // buffer offsets < 2^31 bytes
    return true;
}

int conv_up2_launch(const sg_conv2d_params& p, hipStream_t stream) {
    ConvUpArgs g;
    g.x = p.x; g.w = p.w; g.y = p.y; g.in_scale = p.in_scale;
    g.n = p.n; g.ci = p.ci; g.co = p.co; g.h = p.h; g.wid = p.w_in; g.oh = p.oh; g.ow = p.ow;
    g.w_offset0 = (int)p.phase[0].w_offset; g.w_offset1 = (int)p.phase[1].w_offset;
    g.w_offset2 = (int)p.phase[2].w_offset; g.w_offset3 = (int)p.phase[3].w_offset;
    g.tiles_a = ((p.oh + 1) / 2 + TH - 1) / TH;
    g.tiles_b = ((p.ow + 1) / 2 + TW - 1) / TW;
    const bool wide = p.co > BCO;
    const int bco = wide ? 2 * BCO : BCO;
    const int64_t units = (int64_t)g.tiles_a * g.tiles_b * p.n * ((p.co + bco - 1) / bco);
    SG_REQUIRE(units <= INT32_MAX - 8, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    g.units = (int)units;
    const size_t lds = kInBytes + kWBytes * (wide ? 2 : 1);   // 59904 at most: below the 64 KB that need no opt-in
    const dim3 grid((unsigned)((units + 7) / 8 * 8));
    g.w_bytes = (unsigned)(9u * (unsigned)p.co * (unsigned)p.ci * 2u);
    const bool mod = p.in_scale != nullptr;
#define SG_UP2_LAUNCH(T, WAVES)                                                                                         \
    do {                                                                                                                \
        if (mod) hipLaunchKernelGGL((conv_up2_kernel<T, WAVES, true>), grid, dim3(64 * WAVES), lds, stream, g);         \
        else hipLaunchKernelGGL((conv_up2_kernel<T, WAVES, false>), grid, dim3(64 * WAVES), lds, stream, g);            \
    } while (0)
    if (p.dtype == SG_BF16) {
        if (wide) SG_UP2_LAUNCH(__hip_bfloat16, 8); else SG_UP2_LAUNCH(__hip_bfloat16, 4);
    } else {
        if (wide) SG_UP2_LAUNCH(__half, 8); else SG_UP2_LAUNCH(__half, 4);
    }
#undef SG_UP2_LAUNCH
    return check_launch("sg_conv2d");
}

}  // namespace sg
