// 3x3, unit-stride convolution of 16-bit tensors on the matrix cores: the shape that carries
// nearly all the arithmetic of the StyleGAN2 networks at resolutions of 32 and above (synthesis
// and discriminator layers, and - with transposed weights - their data gradients).  Same
// contract as the general kernel of sg_conv2d.hip (modulation on the input, demodulation /
// noise / bias / leaky ReLU / gain / clamp on the output, optional style-gradient reduction);
// sg_conv2d dispatches here when the geometry qualifies.
//
// What the specialisation buys over the general kernel:
//   * all loop bounds are constants: the 9 taps x 4 pixel tiles x 4 channel tiles of a chunk are one
//     straight-line block of 72 LDS reads and 144 MFMAs that the compiler software-pipelines;
//   * the next chunk's input packets and weights are loaded into registers before the multiply phase
//     and written to LDS after it, so HBM/L2 latency is hidden behind the MFMAs; during the last
//     chunk the epilogue's operands (scales, bias, noise, reduction operand) are loaded instead;
//   * LDS rows are 64 bytes with an XOR swizzle of the 16-byte slots: conflict-free 16-byte reads
//     without padding;
//   * pixels are the matrix rows and channels the columns, so a lane ends up with 4 consecutive
//     pixels of one channel and stores 8 bytes at a time without a transposition.
//
// Tile: 64 output channels x (8 x 32) output pixels of one image per workgroup of 4 waves; the K
// loop walks the input channels 32 at a time.  GEMM view: M = pixels, N = channels, K = 32 x 9.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TW = 32, TH = 8;            // output pixel tile
constexpr int KCH = 32;                   // input channels per chunk (64 bytes per pixel)
constexpr int BM = 64;                    // output channels per workgroup
constexpr int kRow = 64;                  // bytes per LDS row (one pixel or one output channel, 32 channels)
constexpr int kWBytes = 9 * BM * kRow;    // 36864

// Input footprint of a tile for convolution stride STEP (1 or 2) and its staging plan.
template <int STEP> struct Geo {
    static constexpr int LH = (TH - 1) * STEP + 3;                 // footprint rows: 10 / 17
    static constexpr int LW = (TW - 1) * STEP + 3;                 // footprint columns: 34 / 65
    static constexpr int NV = (LW + 7 + 7) / 8;                    // 8-pixel packets per row at any alignment: 6 / 9
    static constexpr int LWP = NV * 8;                             // pixels per LDS image row
    static constexpr int kInBytes = LH * LWP * kRow;               // 30720 / 78336
    static constexpr int ITEMS = LH * NV;                          // (row, packet) items per channel pair
    static constexpr int STEPS = (ITEMS + 15) / 16;                // items per lane: 4 / 10
    static constexpr int kInPadded = STEPS * 16 * 8 * kRow;        // the image with room for all 16 STEPS item slots: 32768 / 81920
    static constexpr int BLOCKS_PER_CU = (kInPadded + kWBytes) * 2 <= 160 * 1024 ? 2 : 1;
};

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

// Byte offset of (row, byte) in a swizzled image: the 16-byte slot s of row r lives at slot s ^ ((r >> 1) & 3),
// which spreads the 16 rows a ds_read_b128 lane group touches over all 16 bank slots.
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

template <class T> __device__ __forceinline__ float lo16(unsigned w) { return Pair<T>::lo(w); }
template <class T> __device__ __forceinline__ float hi16(unsigned w) { return Pair<T>::hi(w); }
template <class T> __device__ __forceinline__ unsigned pack16(float a, float b) { return Pair<T>::pack(a, b); }

// The layer epilogue on four consecutive pixels of one channel, two at a time (packed fp32 instructions: half the
// vector instructions of the scalar form, which is what a persistent workgroup with one wave per SIMD spends its
// epilogue on): x * scale + noise, + bias, leaky ReLU as max(z, z * slope) (0 <= slope <= 1, checked by the
// dispatcher), gain, clamp as a median of three; scale 1, noise 0, bias 0, slope 1, clamp +-inf are no-ops.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ uint2 finish4(const f32x4_t& a, float sc, f32x2_t nz01, f32x2_t nz23, float bs, float slope, float gain, float hi) {
    f32x2_t e0 = {a[0], a[1]}, e1 = {a[2], a[3]};
    e0 = e0 * f32x2_t{sc, sc} + nz01;
    e1 = e1 * f32x2_t{sc, sc} + nz23;
    e0 += f32x2_t{bs, bs};
    e1 += f32x2_t{bs, bs};
    const f32x2_t s0 = e0 * f32x2_t{slope, slope}, s1 = e1 * f32x2_t{slope, slope};
    e0 = f32x2_t{__builtin_fmaxf(e0[0], s0[0]), __builtin_fmaxf(e0[1], s0[1])} * f32x2_t{gain, gain};
    e1 = f32x2_t{__builtin_fmaxf(e1[0], s1[0]), __builtin_fmaxf(e1[1], s1[1])} * f32x2_t{gain, gain};
    return make_uint2(pack16<T>(__builtin_amdgcn_fmed3f(e0[0], -hi, hi), __builtin_amdgcn_fmed3f(e0[1], -hi, hi)),
                      pack16<T>(__builtin_amdgcn_fmed3f(e1[0], -hi, hi), __builtin_amdgcn_fmed3f(e1[1], -hi, hi)));
}

// Buffer addressing: a resource descriptor bounds every access to `bytes` from its base; a byte offset at or beyond the
// bound reads zeros.  A packet outside the image (or a channel beyond the last) therefore costs no branch and no
// select in the loop: its offset is set to kOutside once, or the bound is drawn in front of it.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOutside = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// The same from values that are wave-uniform but not provably so to the compiler (a tile index that went through a
// loop and a division): without the hint every load through the descriptor becomes a loop over the distinct
// descriptors among the lanes, and the wait-count bookkeeping around such loops falls back to "wait for everything".
__device__ __forceinline__ rsrc_t make_rsrc_uniform(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return make_rsrc(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo), (unsigned)__builtin_amdgcn_readfirstlane(bytes));
}
__device__ __forceinline__ uint4 load16(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant in
// every iteration's own code (register arrays indexed by it are then split into registers whatever the optimiser
// makes of the loop).
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// WAVES = 4: the tile above.  WAVES = 8 (layers with more than 64 output channels): 128 output channels per
// workgroup - waves 4..7 multiply the same staged pixels with the second 64 channels, so every staged input byte
// feeds twice the multiplies; one such workgroup per CU, two waves per SIMD.  MOD: the input is modulated
// (in_scale) while it is staged.
//
// Everything the staging needs per chunk is decided before the K loop: which (footprint row, packet) items a lane
// moves and where they lie relative to the chunk's first channel (voff, kOutside for packets outside the image),
// and where their eight pixels go in the LDS image.  A chunk then costs a handful of scalar instructions (the
// descriptors move on by one chunk of channels), one load instruction per packet and, per pair of elements, two
// multiplies and one conversion (or one byte permute without modulation).
constexpr bool conv3x3_persistent(int step, int waves) {
    return (Geo<1>::ITEMS * (step == 1) + Geo<2>::ITEMS * (step == 2) + 4 * waves - 1) / (4 * waves) <= 2;    // items per lane
}

template <class T, int STEP, int WAVES, bool MOD, int ADJ>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : Geo<STEP>::BLOCKS_PER_CU) conv3x3_kernel(Conv3x3Args g) {
    using G = Geo<STEP>;
    constexpr int NV = G::NV, LWP = G::LWP;
    constexpr int THREADS = 64 * WAVES, BMT = BM * (WAVES / 4);   // lanes and output channels per workgroup
    constexpr int SLOTS = THREADS / 16, STEPS = (G::ITEMS + SLOTS - 1) / SLOTS;   // staging: item slots, items per lane
    static_assert(SLOTS * STEPS * 8 * kRow <= G::kInPadded, "the padded image holds every item slot");
    // Persistent where the registers hold the next tile's first chunk beside the current tile's epilogue (two items per
    // lane: the 8-wave form at unit stride, which carries nearly all of this kernel's work); the other forms, launched
    // with one workgroup per tile, leave the loop after their tile (as a loop they spill 14-52 registers per lane).
    constexpr bool PERSISTENT = conv3x3_persistent(STEP, WAVES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_in = smem;
    unsigned char* s_w = smem + G::kInPadded;

    // What a lane derives from its index alone (its rows of the tile, its staging items, its LDS addresses) is derived
    // again at the head of every tile from an index the compiler cannot see through (lane_constants): kept across the
    // tile loop instead, those forty-odd values are live through the multiply phases of every chunk and the kernel
    // spills (42-128 registers per lane against none).
    int tid = threadIdx.x;
    int wave, half, lrow, lgrp;   // pixel rows 2 wave, 2 wave + 1 of the tile; channels 64 half ..; lane = (lrow, lgrp)

    // ---- which tiles: a workgroup is persistent and walks tiles.  Workgroups are dealt to the 8 XCDs round-robin; XCD k
    // takes the k-th contiguous eighth of the tiles and its workgroups walk it side by side, so that neighbouring tiles
    // (shared halo rows and packets) meet in one L2.  What a one-tile workgroup pays around its K loop - its dispatch,
    // the round trip of the first chunk's loads, the drain of its stores before the CU takes the next workgroup - came
    // to 9.5 us per tile against 3.6 us per chunk (a 128-channel layer has four chunks); here the next tile's first
    // chunk is requested before the epilogue of the current one, whose stores drain beside the next K loop. ------------
    const int per_xcd = (g.units + 7) >> 3, stride = gridDim.x >> 3;
    const int unit_end = min(g.units, ((int)(blockIdx.x & 7) + 1) * per_xcd);
    int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (unit >= unit_end) return;
    const int plane = g.h * g.wid;
    const unsigned plane_bytes = (unsigned)plane * 2u;
    auto decode = [&](int u, int& n, int& m0, int& ta, int& tb) {
        u = __builtin_amdgcn_readfirstlane(u);      // (uniform: said once, so that what follows stays scalar)
        tb = u % g.tiles_b; u /= g.tiles_b;
        ta = u % g.tiles_a; u /= g.tiles_a;
        n = u % g.n;
        m0 = (u / g.n) * BMT;
    };

    // ---- staging plan: lane = (channel pair cg = tid & 15, item slot tid >> 4); item = (footprint row, packet) -----
    // (ADJ = 2 at unit stride, where a footprint row has an even number of packets: neighbouring lanes take neighbouring
    // packets of one channel pair and row, so a load instruction touches half as many 32-byte runs: + 2-3 %)
    // What does not depend on the tile - an item's row and packet inside the footprint, its place in the LDS image - is
    // decided here.  The item slots are a few more than the footprint has items and the image is padded to hold them
    // all: a surplus item loads zeros and is written behind the footprint like any other (no branch around a commit).
    int cg, slot0;
    // LDS: item q lands in image rows (slot0 + SLOTS q) * 8 + j; the swizzle of such a row is that of j
    int st_addr[4];             // byte address of the dword of pixel j = 2 jp (+ 64 for j = 2 jp + 1), item q = 0
    // weights [tap][co][ci]: a lane moves, per tap, the 16-byte slot tid & 3 of output channel tid >> 2
    int w_c, w_st;                                                  // w_st + tap * BMT * kRow (BMT is a multiple of 8)
    const unsigned w_tap = (unsigned)(g.co * g.ci) * 2u;
    const T* __restrict__ w = (const T*)g.w;
    // multiply-phase addressing: n-tile j of wave w: output row 2w + (j >> 1), columns 16 (j & 1) .. +15; lane l
    // addresses pixel column l & 15
    int brow[4], a_lane;
    auto lane_constants = [&]() {
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        wave = (tid >> 6) & 3; half = tid >> 8;
        lrow = lane & 15; lgrp = lane >> 4;
        cg = (tid / ADJ) & 15; slot0 = (tid / (16 * ADJ)) * ADJ + (tid % ADJ);
#pragma unroll
        for (int jp = 0; jp < 4; jp++) st_addr[jp] = slot0 * 8 * kRow + swz(2 * jp, cg * 4);
        w_c = (tid & 3) * 8;
        w_st = swz(tid >> 2, (tid & 3) * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) brow[j] = (2 * wave + (j >> 1)) * STEP * LWP + ((j & 1) * 16 + lrow) * STEP + g.px_col;
        a_lane = lrow * kRow + (((lgrp ^ (lrow >> 1)) & 3) << 4);
    };

    // The tile whose packets are being requested (the next one, once the current tile's last chunk is multiplied): byte
    // offsets of the items' packets of channel 2 cg from a chunk's first channel, the lane's weight slot, the sample.
    // A tile that does not exist (after a workgroup's last) is planned with everything outside: its loads return zeros
    // without a memory access and the tile loop needs no branch around them.
    unsigned voff[STEPS];
    unsigned w_voff = kOutside;
    int n_st = 0;
    auto stage_plan = [&](int n, int m0, int a0, int b0, bool exists) {
        const int px_first = b0 * STEP + g.px_shift;                // first pixel column of the staged packets
        const int iy_first = a0 * STEP + g.dy0;
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const int item = slot0 + SLOTS * q, ry = item / NV;
            const int iy = iy_first + ry, px0 = px_first + (item - ry * NV) * 8;
            const bool ok = exists && item < G::ITEMS && iy >= 0 && iy < g.h && px0 >= 0 && px0 < g.wid;
            voff[q] = ok ? (unsigned)(2 * cg * plane + iy * g.wid + px0) * 2u : kOutside;
        }
        n_st = exists ? n : 0;
        const int w_o = m0 + (tid >> 2);
        w_voff = (exists && w_o < g.co) ? (unsigned)(w_o * g.ci + w_c) * 2u : kOutside;
    };

    uint4 va[STEPS], vb[STEPS], vw[9];
    float s0 = 1.f, s1 = 1.f;

    // The loads of a chunk: 2 STEPS input packets, the two modulation scales, 9 weight packets per lane.  The texture
    // path of a CU takes 16 cycles per 16-byte load instruction of a wave, 8 waves x 15 loads = 1,900 cycles per chunk:
    // issued as one batch in front of the multiply phase, the loads queue there and the waves start multiplying one
    // after the other as their batches get through (cycle stamps: the last wave of a SIMD entered its multiply
    // phase 3,000 cycles after the barrier, its SIMD partner waited 2,600 at the next one).  So only a tile's first chunk
    // is fetched as a batch (in front of the previous tile's epilogue); every other chunk's loads are dealt over the nine
    // taps of the multiply phase before it (fetch_part), one or two instructions beside 16 matrix instructions.
    rsrc_t d_x = make_rsrc(g.x, 0u), d_s = make_rsrc(g.x, 0u), d_w = make_rsrc(g.x, 0u);
    unsigned wv = kOutside;
    auto fetch_begin = [&](int c0) {
        const int left = min(KCH, g.ci - c0);                       // channels of this chunk
        const T* xs = (const T*)g.x + (unsigned)(n_st * g.ci * plane);   // fewer than 2^31 elements per tensor
        d_x = make_rsrc_uniform(xs + (size_t)c0 * plane, (unsigned)left * plane_bytes);
        if constexpr (MOD) d_s = make_rsrc_uniform(g.in_scale + (size_t)n_st * g.ci + c0, (unsigned)left * 4u);
        d_w = make_rsrc_uniform(w + c0, 9u * w_tap);
        wv = (w_c < left) ? w_voff : kOutside;       // the lane's slot of 8 channels lies in the chunk or not
    };
    auto fetch_part = [&](int t) {                   // t = 0 .. 8
#pragma unroll
        for (int l = 0; l < 2 * STEPS; l++) {        // load l = packet of channel 2 cg + (l & 1) of item l >> 1
            if ((l * 9) / (2 * STEPS) != t) continue;
            if (l & 1) vb[l >> 1] = load16(d_x, voff[l >> 1], plane_bytes);      // (ci is a multiple of 8)
            else va[l >> 1] = load16(d_x, voff[l >> 1], 0);
        }
        if constexpr (MOD) {
            // (two dword loads: hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b64 to a single dword load)
            if (t == 4) s0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s, cg * 8, 0, 0));   // zero beyond the last channel
            if (t == 5) s1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s, cg * 8, 4, 0));
        }
        vw[t] = load16(d_w, wv, (unsigned)t * w_tap);
    };
    auto fetch = [&](int c0) {
        fetch_begin(c0);
#pragma unroll
        for (int t = 0; t < 9; t++) fetch_part(t);
    };

    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const unsigned wa[4] = {va[q].x, va[q].y, va[q].z, va[q].w};
            const unsigned wb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                unsigned e0, e1;   // pixels 2 jp and 2 jp + 1: (channel 2 cg, channel 2 cg + 1)
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
        for (int q = 0; q < 9; q++) *reinterpret_cast<uint4*>(s_w + w_st + q * (BMT * kRow)) = vw[q];
    };

    f32x4_t acc[4][4];   // [channel tile m][pixel tile j]: rows = pixels 4 lgrp + r, column = channel lrow

    // ---- the tile being multiplied: where its outputs go; epilogue operands, loaded during its last multiply phase ---
    // (The bias stays in its 16 bits until the epilogue: widened where it is loaded - inside the branch on g.bias - each
    // of the four loads would be waited for on the spot, four memory round trips in a row per workgroup.)
    int n = 0, m0 = 0, ta = 0, tb = 0, a0 = 0, b0 = 0;
    float sc[4];
    unsigned bs[4];
    float4 nz[4];
    uint2 rx[4][4];
    int ch0 = 0;
    const int ch_stride = 16 * g.oh * g.ow;
    unsigned off0[4];
    bool inside[4];
    auto out_plan = [&]() {
        ch0 = m0 + 64 * half + lrow;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int a = a0 + 2 * wave + (j >> 1), b = b0 + (j & 1) * 16 + 4 * lgrp;
            inside[j] = a < g.oh && b < g.ow;
            off0[j] = (unsigned)(((n * g.co + ch0) * g.oh + a) * g.ow + b);
        }
    };
    auto fetch_epilogue = [&]() {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int ch = min(ch0 + 16 * m, g.co - 1);
            sc[m] = g.out_scale ? g.out_scale[n * g.co + ch] : 1.f;
            bs[m] = g.bias ? __builtin_bit_cast(unsigned short, ((const T*)g.bias)[ch]) : (unsigned short)0;   // widened in the epilogue
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            nz[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.noise && inside[j]) {
                const int a = a0 + 2 * wave + (j >> 1), b = b0 + (j & 1) * 16 + 4 * lgrp;
                nz[j] = *reinterpret_cast<const float4*>(g.noise + n * g.noise_stride + a * g.ow + b);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                rx[j][m] = make_uint2(0, 0);
                if (g.reduce_out && inside[j] && ch0 + 16 * m < g.co)
                    rx[j][m] = *reinterpret_cast<const uint2*>((const T*)g.reduce_x + off0[j] + m * ch_stride);
            }
        }
    };

    // ---- K loop ---------------------------------------------------------------------------------------------
    // 9 taps per chunk; the fragments of tap t + 1 are read while the MFMAs of tap t run (two register sets).
    auto read_tap = [&](int tap, uint4 (&a)[4], uint4 (&b)[4]) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = *reinterpret_cast<const uint4*>(s_w + (tap * BMT + 64 * half + m * 16) * kRow + a_lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = brow[j] + ky * LWP + kx;
            b[j] = *reinterpret_cast<const uint4*>(s_in + row * kRow + (((lgrp ^ (row >> 1)) & 3) << 4));
        }
    };
    auto multiply = [&](auto with_fetch) {
        uint4 fa[2][4], fb[2][4];
        read_tap(0, fa[0], fb[0]);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap + 1 < 9) read_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
            if constexpr (decltype(with_fetch)::value) fetch_part(tap);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < 4; m++) Mma16<T>::run(fb[tap & 1][j], fa[tap & 1][m], acc[m][j]);   // rows: pixels, columns: channels
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int c_last = ((g.ci - 1) / KCH) * KCH;
    const float noise_gain = (g.noise && g.noise_gain) ? *g.noise_gain : 1.f;
    T* __restrict__ y = (T*)g.y;
    lane_constants();
    decode(unit, n, m0, ta, tb);
    stage_plan(n, m0, ta * TH, tb * TW, true);
    fetch(0);
    for (;;) {
        lane_constants();
        a0 = ta * TH; b0 = tb * TW;
        out_plan();
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < c_last; c0 += KCH) {
            __syncthreads();   // everyone is done reading the previous chunk (or the previous tile's last)
            commit();
            __syncthreads();
            fetch_begin(c0 + KCH);
            multiply(std::true_type{});      // (the next chunk's loads are issued between its taps)
        }
        __syncthreads();
        commit();
        __syncthreads();
        fetch_epilogue();      // the last chunk's multiply phase covers the epilogue's loads instead
        multiply(std::false_type{});

        // ---- the next tile's first chunk goes out ahead of this tile's epilogue (loads return in order: the epilogue's
        // wait for its operands, requested before the multiply phase, leaves these in flight) -----------------------
        const int next = unit + stride;
        const bool more = PERSISTENT && next < unit_end;
        int n2 = n, m2 = m0, ta2 = ta, tb2 = tb;
        if constexpr (PERSISTENT) decode(more ? next : unit, n2, m2, ta2, tb2);
        if constexpr (PERSISTENT) {
            stage_plan(n2, m2, ta2 * TH, tb2 * TW, more);
            fetch(0);
        }

        // ---- epilogue: straight-line arithmetic (scale 1, noise 0, bias 0, slope 1, clamp +-inf are no-ops) -------
        float red[4] = {0.f, 0.f, 0.f, 0.f};
        const float bsf[4] = {lo16<T>(bs[0]), lo16<T>(bs[1]), lo16<T>(bs[2]), lo16<T>(bs[3])};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x2_t nz01 = {nz[j].x * noise_gain, nz[j].y * noise_gain}, nz23 = {nz[j].z * noise_gain, nz[j].w * noise_gain};
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (!inside[j] || ch0 + 16 * m >= g.co) continue;
                const f32x4_t v = acc[m][j];
                if (g.reduce_out)
                    red[m] += v[0] * lo16<T>(rx[j][m].x) + v[1] * hi16<T>(rx[j][m].x) + v[2] * lo16<T>(rx[j][m].y) + v[3] * hi16<T>(rx[j][m].y);
                *reinterpret_cast<uint2*>(y + off0[j] + m * ch_stride) = finish4<T>(v, sc[m], nz01, nz23, bsf[m], g.slope, g.gain, g.hi);
            }
        }
        if (g.reduce_out) {
            // lanes l, l+16, l+32, l+48 share their channels; then the four waves through LDS; one partial row per tile
            __syncthreads();   // every wave has left the multiply loop; LDS can be reused
            float* s_red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                float r = red[m];
                r += __shfl_xor(r, 16, 64);
                r += __shfl_xor(r, 32, 64);
                if (lgrp == 0) s_red[wave * BMT + 64 * half + m * 16 + lrow] = r;
            }
            __syncthreads();
            if (tid < BMT && m0 + tid < g.co) {
                const float r = s_red[tid] + s_red[BMT + tid] + s_red[2 * BMT + tid] + s_red[3 * BMT + tid];
                g.reduce_out[((n * g.tiles_a + ta) * g.tiles_b + tb) * g.co + m0 + tid] = r;
            }
        }
        if (!PERSISTENT || !more) break;
        unit = next; n = n2; m0 = m2; ta = ta2; tb = tb2;
    }
}


// The weights-resident kernel at unit stride hands a lane eight consecutive pixels of a channel (16-byte stores): row i
// of its pixel tile j is pixel 8 (i / 4) + 4 (j & 1) + i % 4 of the tile row, not 16 (j & 1) + i.  The 16 image rows a
// ds_read_b128 lane group then touches are four runs of four, eight apart, and the slot swizzle that keeps them on 16
// different bank slots takes bits 1 and 3 of the row (bits 1 and 2 for 16 rows in a run).
template <bool PERM> __device__ __forceinline__ int slot_swizzle(int row) {
    return PERM ? (((row >> 1) & 1) | ((row >> 2) & 2)) : ((row >> 1) & 3);
}

// ---------------------------------------------------------------------------------------------------------
// Narrow layers (at most 64 input and 64 output channels: the 256^2 ... 1024^2 layers, whose arithmetic
// intensity puts them at or near the HBM roofline): the whole weight tensor (<= 72 KB) stays in LDS for the
// lifetime of a persistent workgroup, which then only streams pixel tiles: all input channels of the next tile
// are loaded into registers while the current tile is multiplied, and the epilogue's operands are loaded ahead of
// the multiply phase.  CHUNKS = ceil(ci / 32), M_TILES = ceil(co / 16).  STEP = 2 (one chunk only: the strided
// layer behind the blur of the discriminator's first block, 32 -> 64 channels) has the LDS of a CU to itself; the
// general kernel would run it one tile per workgroup with nothing in flight while it multiplies.
template <class T, int CHUNKS, int M_TILES, int STEP, bool MOD>
__global__ void __launch_bounds__(kThreads, (CHUNKS == 1 && STEP == 1 && M_TILES <= 2) ? 2 : 1) conv3x3_resident_kernel(Conv3x3Args g, int units) {
    using G = Geo<STEP>;
    constexpr int NV = G::NV, LWP = G::LWP, BMW = 16 * M_TILES;
    constexpr bool PERM = (STEP == 1);      // a lane's outputs: 8 consecutive pixels (PERM) or 4 + 4 sixteen apart
    constexpr int kWChunk = 9 * BMW * kRow;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_w = smem;                             // [chunk][tap][BMW][32 channels]
    unsigned char* s_in = smem + CHUNKS * kWChunk;         // [chunk][footprint pixel][32 channels], kImage bytes per chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int plane = g.h * g.wid;
    const T* __restrict__ w = (const T*)g.w;

    // ---- weights: once per workgroup ------------------------------------------------------------------------
    for (int e = tid; e < CHUNKS * 9 * BMW * 4; e += kThreads) {
        const int slot = e & 3, row = (e >> 2) % (9 * BMW), ch = (e >> 2) / (9 * BMW);   // row = tap * BMW + o
        const int o = row % BMW, tap = row / BMW, c = ch * KCH + slot * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o < g.co && c < g.ci) v = *reinterpret_cast<const uint4*>(w + (unsigned)((tap * g.co + o) * g.ci + c));
        *reinterpret_cast<uint4*>(s_w + ch * kWChunk + swz(row, slot * 16)) = v;
    }

    // ---- input fetch / commit: as conv3x3_kernel, all chunks of a tile at once.  What does not depend on the tile
    // (an item's row and packet inside the footprint, its place in the LDS image) is decided here; per tile a lane
    // decides once per item whether the packet lies in the image (else its offset is kOutside: the load returns zeros).
    const int cg = tid & 15, slot0 = tid >> 4;
    // Every lane moves STEPS items; the 16 STEPS item slots are a few more than the footprint has items, and the image
    // of a chunk is padded to hold them all (kImage): the surplus items load zeros (kOutside) and are written behind the
    // footprint like any other.  (Skipping them was a branch around the last item's LDS writes - and around the wait
    // for its two loads: for the compiler those stayed in flight in the wave that skipped, and the next write to their
    // registers, the plan of the tile after, waited for everything outstanding, the previous tile's stores included:
    // 2,400 cycles per tile.)
    constexpr int STEPS = G::STEPS, kImage = G::kInPadded;
    int item_ry[STEPS], item_px[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; q++) {
        const int item = slot0 + 16 * q;
        item_ry[q] = (item < G::ITEMS) ? item / NV : (1 << 20);     // an item that does not exist lies below any image
        item_px[q] = (item % NV) * 8;
    }
    int st_addr[4];
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {          // image row slot0 * 8 + 2 jp (the rows of the lane's other items: + 128 q, same swizzle)
        const int row = slot0 * 8 + 2 * jp;
        st_addr[jp] = row * kRow + ((((cg >> 2) ^ slot_swizzle<PERM>(row)) & 3) << 4) + (cg & 3) * 4;
    }
    const unsigned plane_bytes = (unsigned)plane * 2u;
    uint4 va[CHUNKS * STEPS], vb[CHUNKS * STEPS];      // [chunk][item]
    float s0[CHUNKS], s1[CHUNKS];
    auto decode = [&](int unit, int& n, int& a0, int& b0) {
        unit = __builtin_amdgcn_readfirstlane(unit);      // (uniform: said once, so that what follows stays scalar)
        const int tb = unit % g.tiles_b; unit /= g.tiles_b;
        const int ta = unit % g.tiles_a;
        n = unit / g.tiles_a;
        a0 = ta * TH; b0 = tb * TW;
    };
    // A tile's loads: where its packets lie and the descriptors of its chunks (fetch_plan), then the loads themselves,
    // dealt over the CHUNKS x 9 taps of the multiply phase of the tile before (fetch_part): issued as one batch they
    // queue in the texture path - 8,000 cycles per tile for 64 -> 64 channels at 512^2, each 16-byte packet of a lane
    // pulling in a 128-byte line of which the tile uses 96 bytes at most - and the wave that issues them stands still
    // for that long; between the taps the same queue drains beside the matrix instructions.  A tile that does not exist
    // (after a workgroup's last) is planned with every packet outside the image: its loads return zeros without a
    // memory access, and the tile loop needs no branch inside the multiply phase.
    unsigned voff[STEPS];
    rsrc_t d_x[CHUNKS], d_s[CHUNKS];
    int p_n = 0, p_a0 = 0, p_b0 = 0;                       // the planned tile: sample, first output row and column
    auto fetch_plan = [&](int unit, bool exists) {
        int n, a0, b0;
        decode(unit, n, a0, b0);
        p_n = n; p_a0 = a0; p_b0 = b0;
        const int iy_first = a0 * STEP + g.dy0, px_first = b0 * STEP + g.px_shift;
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const int iy = iy_first + item_ry[q], px0 = px_first + item_px[q];
            const bool ok = exists && iy >= 0 && iy < g.h && px0 >= 0 && px0 < g.wid;
            voff[q] = ok ? (unsigned)(2 * cg * plane + iy * g.wid + px0) * 2u : kOutside;
        }
        n = exists ? n : 0;
        const T* xs = (const T*)g.x + (unsigned)(n * g.ci * plane);
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ch++) {
            const int left = max(min(KCH, g.ci - ch * KCH), 0);
            d_x[ch] = make_rsrc_uniform(xs + (size_t)ch * KCH * plane, (unsigned)left * plane_bytes);
            if constexpr (MOD) d_s[ch] = make_rsrc_uniform(g.in_scale + (size_t)n * g.ci + ch * KCH, (unsigned)left * 4u);
        }
    };
    constexpr int kTaps = CHUNKS * 9, kLoads = CHUNKS * STEPS * 2;     // 16-byte loads of a tile per lane
    auto fetch_part = [&](auto tap) __attribute__((always_inline)) {   // tap: std::integral_constant, 0 .. kTaps - 1
        constexpr int t = decltype(tap)::value;
        static_for<kLoads>([&](auto load) __attribute__((always_inline)) {
            // load l: chunk l / (2 STEPS), item (l / 2) % STEPS, channel 2 cg + (l & 1)
            constexpr int l = decltype(load)::value, ch = l / (2 * STEPS), q = (l >> 1) % STEPS;
            if constexpr ((l * kTaps) / kLoads == t) {
                if constexpr (l & 1) vb[l >> 1] = load16(d_x[ch], voff[q], plane_bytes);
                else va[l >> 1] = load16(d_x[ch], voff[q], 0);
            }
        });
        if constexpr (MOD && t % 9 == 4) s0[t / 9] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s[t / 9], cg * 8, 0, 0));
        if constexpr (MOD && t % 9 == 5) s1[t / 9] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s[t / 9], cg * 8, 4, 0));
    };
    auto commit = [&]() {
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ch++)
#pragma unroll
            for (int q = 0; q < STEPS; q++) {
                const uint4 pa = va[ch * STEPS + q], pb = vb[ch * STEPS + q];
                const unsigned wa[4] = {pa.x, pa.y, pa.z, pa.w};
                const unsigned wb[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                for (int jp = 0; jp < 4; jp++) {
                    unsigned e0, e1;
                    if constexpr (MOD) {
                        e0 = Pair<T>::pack(Pair<T>::lo(wa[jp]) * s0[ch], Pair<T>::lo(wb[jp]) * s1[ch]);
                        e1 = Pair<T>::pack(Pair<T>::hi(wa[jp]) * s0[ch], Pair<T>::hi(wb[jp]) * s1[ch]);
                    } else {
                        e0 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x05040100u);
                        e1 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x07060302u);
                    }
                    unsigned char* dst = s_in + ch * kImage + st_addr[jp] + q * (16 * 8 * kRow);
                    *reinterpret_cast<unsigned*>(dst) = e0;
                    *reinterpret_cast<unsigned*>(dst + kRow) = e1;
                }
            }
    };

    int brow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int px = PERM ? 8 * (lrow >> 2) + 4 * (j & 1) + (lrow & 3) : (j & 1) * 16 + lrow;   // pixel of tile row i = lrow
        brow[j] = (2 * wave + (j >> 1)) * STEP * LWP + px * STEP + g.px_col;
    }
    const int a_lane = lrow * kRow + (((lgrp ^ (lrow >> 1)) & 3) << 4);
    const float noise_gain = (g.noise && g.noise_gain) ? *g.noise_gain : 1.f;
    const int ch0 = lrow;
    const int ch_stride = 16 * g.oh * g.ow;
    T* __restrict__ y = (T*)g.y;
    // The bias is the same for every tile: loaded and widened once.  (Inside the tile loop the conversion made the
    // compiler wait for the 16-bit load on the spot - with every load and store the wave had in flight, the next
    // tile's packets included: four memory round trips per tile, 10,000 of its 22,000 cycles.)
    float bs[M_TILES];
#pragma unroll
    for (int m = 0; m < M_TILES; m++) {
        const unsigned short bits = g.bias ? __builtin_bit_cast(unsigned short, ((const T*)g.bias)[min(ch0 + 16 * m, g.co - 1)]) : (unsigned short)0;
        bs[m] = lo16<T>(bits);
    }

    // Workgroups are dealt to the 8 XCDs round-robin; each XCD (own L2) takes a contiguous range of the tiles and its
    // workgroups walk it together in raster order, so that the halo rows and the packets shared by neighbouring tiles
    // are found in that L2 instead of being fetched once per XCD.
    const int xcd = blockIdx.x & 7, per_xcd = (units + 7) >> 3, stride = gridDim.x >> 3;
    const int unit_end = min(units, (xcd + 1) * per_xcd);
    int unit = xcd * per_xcd + (blockIdx.x >> 3);
    if (unit < unit_end) {     // a workgroup's first tile: all at once
        fetch_plan(unit, true);
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ch++) {
#pragma unroll
            for (int q = 0; q < STEPS; q++) {
                va[ch * STEPS + q] = load16(d_x[ch], voff[q], 0);
                vb[ch * STEPS + q] = load16(d_x[ch], voff[q], plane_bytes);
            }
            if constexpr (MOD) {
                s0[ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s[ch], cg * 8, 0, 0));
                s1[ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_s[ch], cg * 8, 4, 0));
            }
        }
    }
    for (; unit < unit_end; unit += stride) {
        const int n = p_n, a0 = p_a0, b0 = p_b0;      // (decoded when the tile was planned)
        __syncthreads();   // everyone is done reading the previous tile (and, the first time, the weights are in place)
        commit();
        __syncthreads();
        fetch_plan(unit + stride, unit + stride < unit_end);

        // ---- epilogue operands of this tile, loaded ahead of the multiply phase - and ahead of the next tile's
        // packets: loads return in order, so the epilogue's wait for its operands leaves those in flight ---------------
        // (PERM: a lane holds pixels b .. b + 7 of output row a0 + 2 wave + h, h = 0, 1 - the first four from pixel tile 2 h,
        // the others from 2 h + 1; else pixels b .. b + 3 of tile j, b = 16 (j & 1) + 4 lgrp)
        float sc[M_TILES];
        float4 nz[4];
        uint2 rx[4][M_TILES];
        unsigned off0[4];
        bool inside[4];
#pragma unroll
        for (int m = 0; m < M_TILES; m++) {
            const int ch = min(ch0 + 16 * m, g.co - 1);
            sc[m] = g.out_scale ? g.out_scale[n * g.co + ch] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int a = a0 + 2 * wave + (j >> 1), b = b0 + (PERM ? 8 * lgrp + 4 * (j & 1) : (j & 1) * 16 + 4 * lgrp);
            inside[j] = a < g.oh && b < g.ow;
            off0[j] = (unsigned)(((n * g.co + ch0) * g.oh + a) * g.ow + b);
            nz[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.noise && inside[j]) nz[j] = *reinterpret_cast<const float4*>(g.noise + n * g.noise_stride + a * g.ow + b);
        }
        if (g.reduce_out) {     // (one scalar branch around the lot: a layer without the reduction pays nothing per load)
#pragma unroll
            for (int m = 0; m < M_TILES; m++) {
                if constexpr (PERM) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        uint4 v = make_uint4(0, 0, 0, 0);
                        if (inside[2 * h] && ch0 + 16 * m < g.co) v = *reinterpret_cast<const uint4*>((const T*)g.reduce_x + off0[2 * h] + m * ch_stride);
                        rx[2 * h][m] = make_uint2(v.x, v.y);
                        rx[2 * h + 1][m] = make_uint2(v.z, v.w);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        rx[j][m] = make_uint2(0, 0);
                        if (inside[j] && ch0 + 16 * m < g.co)
                            rx[j][m] = *reinterpret_cast<const uint2*>((const T*)g.reduce_x + off0[j] + m * ch_stride);
                    }
                }
            }
        }
        // ---- multiply: chunks x 9 taps, fragments of the next tap read while this one multiplies; the next tile's
        // loads go out between the taps ---------------------------------------------------------------------------
        f32x4_t acc[M_TILES][4];
#pragma unroll
        for (int m = 0; m < M_TILES; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint4 fa[2][M_TILES], fb[2][4];
        auto read_tap = [&](int t, uint4 (&a)[M_TILES], uint4 (&b)[4]) {   // t = chunk * 9 + tap
            const int ch = t / 9, tap = t % 9, ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int m = 0; m < M_TILES; m++)
                a[m] = *reinterpret_cast<const uint4*>(s_w + ch * kWChunk + (tap * BMW + m * 16) * kRow + a_lane);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = brow[j] + ky * LWP + kx;
                b[j] = *reinterpret_cast<const uint4*>(s_in + ch * kImage + row * kRow + (((lgrp ^ slot_swizzle<PERM>(row)) & 3) << 4));
            }
        };
        read_tap(0, fa[0], fb[0]);
        static_for<kTaps>([&](auto tap) __attribute__((always_inline)) {
            constexpr int t = decltype(tap)::value;
            if constexpr (t + 1 < kTaps) read_tap(t + 1, fa[(t + 1) & 1], fb[(t + 1) & 1]);
            fetch_part(tap);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < M_TILES; m++) Mma16<T>::run(fb[t & 1][j], fa[t & 1][m], acc[m][j]);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ---- epilogue -------------------------------------------------------------------------------------------
        float red[M_TILES];
#pragma unroll
        for (int m = 0; m < M_TILES; m++) red[m] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j += (PERM ? 2 : 1)) {
            f32x2_t nz01[2], nz23[2];
#pragma unroll
            for (int k = 0; k < (PERM ? 2 : 1); k++) {
                nz01[k] = f32x2_t{nz[j + k].x * noise_gain, nz[j + k].y * noise_gain};
                nz23[k] = f32x2_t{nz[j + k].z * noise_gain, nz[j + k].w * noise_gain};
            }
#pragma unroll
            for (int m = 0; m < M_TILES; m++) {
                if (!inside[j] || ch0 + 16 * m >= g.co) continue;       // (PERM: the width is a multiple of 8 - the pair is inside or not)
                uint2 out[2];
#pragma unroll
                for (int k = 0; k < (PERM ? 2 : 1); k++) {
                    const f32x4_t v = acc[m][j + k];
                    const uint2 r = rx[j + k][m];
                    if (g.reduce_out) red[m] += v[0] * lo16<T>(r.x) + v[1] * hi16<T>(r.x) + v[2] * lo16<T>(r.y) + v[3] * hi16<T>(r.y);
                    out[k] = finish4<T>(v, sc[m], nz01[k], nz23[k], bs[m], g.slope, g.gain, g.hi);
                }
                if constexpr (PERM) *reinterpret_cast<uint4*>(y + off0[j] + m * ch_stride) = make_uint4(out[0].x, out[0].y, out[1].x, out[1].y);
                else *reinterpret_cast<uint2*>(y + off0[j] + m * ch_stride) = out[0];
            }
        }
        if (g.reduce_out) {
            // lanes l, l+16, l+32, l+48 share their channels; then the four waves through the input image's LDS
            __syncthreads();   // every wave has left the multiply loop
            float* s_red = reinterpret_cast<float*>(s_in);
#pragma unroll
            for (int m = 0; m < M_TILES; m++) {
                float r = red[m];
                r += __shfl_xor(r, 16, 64);
                r += __shfl_xor(r, 32, 64);
                if (lgrp == 0) s_red[wave * BMW + m * 16 + lrow] = r;
            }
            __syncthreads();
            if (tid < BMW && tid < g.co) {
                const float r = s_red[tid] + s_red[BMW + tid] + s_red[2 * BMW + tid] + s_red[3 * BMW + tid];
                g.reduce_out[(unsigned)(unit * g.co + tid)] = r;   // unit = (n * tiles_a + ta) * tiles_b + tb
            }
        }
    }
}

}  // namespace

bool conv3x3_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 1 || p.in_step < 1 || p.in_step > 2 || p.out_step != 1) return false;
    const sg_conv2d_phase& ph = p.phase[0];
    if (ph.nky != 3 || ph.nkx != 3 || ph.oy0 != 0 || ph.ox0 != 0 || ph.w_offset != 0) return false;
    // smaller outputs would leave most of the 8 x 32 tile empty (half of it is accepted at stride 2: the strided 512 -> 512
    // layer at 32^2 -> 16^2 runs 1.6x the rate of the general kernel with every second column of its tiles idle)
    if (p.ow < (p.in_step == 2 ? TW / 2 : TW) || p.oh < 5) return false;
    if (p.act == 3 && !(p.alpha >= 0.f && p.alpha <= 1.f)) return false;   // the epilogue takes the leaky ReLU as a max
    if (p.w_in % 8 != 0 || p.ow % 4 != 0 || p.ci % 8 != 0) return false;
    if ((int64_t)p.ci * p.h * p.w_in >= (int64_t)1 << 30) return false;   // a sample's bytes bound the buffer offsets (< 2^31)
    if (!aligned16(p.x) || !aligned16(p.w)) return false;
    const auto aligned8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; };
    if (!aligned8(p.y) || !aligned8(p.reduce_x)) return false;   // reduce_x, not reduce_out: callers size reduce_out from this answer
    if (p.noise && (!aligned16(p.noise) || p.noise_stride_n % 4 != 0)) return false;
    // Unit stride with one or two chunks goes to the weights-resident kernel when the output is narrow too; a wide
    // output with a narrow input stays on the general kernel.  The strided form wins from 32 channels.
    if (p.in_step == 1 && p.ci <= 2 * KCH && p.co > BM) return false;
    if (p.in_step == 2 && p.ci < KCH) return false;
    return true;
}

bool conv3x3_resident(const sg_conv2d_params& p) {   // among the eligible geometries: the weights-resident kernel
    if (p.in_step == 2) return p.ci <= KCH && p.co <= BM && p.co > 32;
    // (at unit stride it moves eight output pixels per lane and access)
    return p.ci <= 2 * KCH && p.co <= BM && p.ow % 8 == 0 && aligned16(p.y) && aligned16(p.reduce_x);
}

int conv3x3_reduce_rows(const sg_conv2d_params& p) {
    return ((p.oh + TH - 1) / TH) * ((p.ow + TW - 1) / TW);
}

int compute_units() {   // of the current device; asked once per process and device
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int cu = cached[dev].load(std::memory_order_relaxed);
    if (cu == 0) {
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        cached[dev].store(cu, std::memory_order_relaxed);
    }
    return cu;
}

template <class T, int CHUNKS, int M_TILES, int STEP, bool MOD>
static int launch_resident_mod(const Conv3x3Args& g, int64_t units, hipStream_t stream) {
    constexpr size_t lds = (size_t)CHUNKS * (9 * 16 * M_TILES * kRow + Geo<STEP>::kInPadded);
    static_assert(lds <= 160 * 1024, "weights and tile do not fit the LDS of a CU");
    auto kernel = conv3x3_resident_kernel<T, CHUNKS, M_TILES, STEP, MOD>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    const int per_cu = (CHUNKS == 1) ? std::min<int>(2, (int)((160 * 1024) / lds)) : 1;
    const int64_t grid = std::max<int64_t>(8, std::min<int64_t>((units + 7) / 8 * 8, (int64_t)per_cu * compute_units()) / 8 * 8);   // a multiple of the 8 XCDs
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kThreads), lds, stream, g, (int)units);
    return check_launch("sg_conv2d");
}

template <class T, int CHUNKS, int M_TILES, int STEP = 1>
static int launch_resident(const Conv3x3Args& g, int64_t units, hipStream_t stream) {
    return g.in_scale ? launch_resident_mod<T, CHUNKS, M_TILES, STEP, true>(g, units, stream)
                      : launch_resident_mod<T, CHUNKS, M_TILES, STEP, false>(g, units, stream);
}

template <class T>
static int launch_resident_for(const Conv3x3Args& g, int64_t units, int ci, int co, int step, hipStream_t stream) {
    if (step == 2) return launch_resident<T, 1, 4, 2>(g, units, stream);
    const bool one = ci <= KCH, narrow = co <= 32;
    if (one) return narrow ? launch_resident<T, 1, 2>(g, units, stream) : launch_resident<T, 1, 4>(g, units, stream);
    return narrow ? launch_resident<T, 2, 2>(g, units, stream) : launch_resident<T, 2, 4>(g, units, stream);
}

template <class T, int STEP, int WAVES, bool MOD>
static int launch_step(Conv3x3Args g, hipStream_t stream) {
    constexpr int BMT = BM * (WAVES / 4);
    constexpr size_t lds = Geo<STEP>::kInPadded + kWBytes * (WAVES / 4);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS of a CU");
    constexpr int per_cu = (WAVES == 8) ? 1 : Geo<STEP>::BLOCKS_PER_CU;
    const int64_t units = (int64_t)g.tiles_a * g.tiles_b * g.n * ((g.co + BMT - 1) / BMT);
    SG_REQUIRE(units <= INT32_MAX - 8, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    g.units = (int)units;
    auto kernel = conv3x3_kernel<T, STEP, WAVES, MOD, STEP == 1 ? 2 : 1>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    // persistent workgroups: as many as the device holds at once, a multiple of the 8 XCDs; else one per tile
    const int64_t all = (units + 7) / 8 * 8;
    const int64_t grid = conv3x3_persistent(STEP, WAVES) ? std::max<int64_t>(8, std::min<int64_t>(all, (int64_t)per_cu * compute_units()) / 8 * 8) : all;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * WAVES), lds, stream, g);
    return check_launch("sg_conv2d");
}

// Layers with more than 64 output channels take the 8-wave form (128 channels per workgroup).
template <class T, bool MOD>
static int launch_wide_or_not_mod(const Conv3x3Args& g, int step, hipStream_t stream) {
    if (step == 1) return (g.co > BM) ? launch_step<T, 1, 8, MOD>(g, stream) : launch_step<T, 1, 4, MOD>(g, stream);
    return (g.co > BM) ? launch_step<T, 2, 8, MOD>(g, stream) : launch_step<T, 2, 4, MOD>(g, stream);
}
template <class T>
static int launch_wide_or_not(const Conv3x3Args& g, int step, hipStream_t stream) {
    return g.in_scale ? launch_wide_or_not_mod<T, true>(g, step, stream) : launch_wide_or_not_mod<T, false>(g, step, stream);
}

int conv3x3_launch(const sg_conv2d_params& p, hipStream_t stream) {
    Conv3x3Args g;
    g.x = p.x; g.w = p.w; g.y = p.y;
    g.in_scale = p.in_scale; g.out_scale = p.out_scale; g.bias = p.bias;
    g.noise = p.noise; g.noise_gain = p.noise_gain; g.reduce_x = p.reduce_x; g.reduce_out = p.reduce_out;
    g.n = p.n; g.ci = p.ci; g.co = p.co; g.h = p.h; g.wid = p.w_in; g.oh = p.oh; g.ow = p.ow;
    g.dy0 = p.phase[0].dy0;
    // input column of output column b, tap kx: b * step + dx0 + kx.  A tile's first output column is a multiple of 32,
    // so its packets start at the multiple of 8 at or below dx0, relative to b0 * step.
    const int dx0 = p.phase[0].dx0;
    const int fl = (dx0 >= 0) ? dx0 / 8 : -((-dx0 + 7) / 8);
    g.px_shift = fl * 8;
    g.px_col = dx0 - g.px_shift;    // column of the footprint's first pixel inside the LDS image, 0..7
    g.tiles_a = (p.oh + TH - 1) / TH;
    g.tiles_b = (p.ow + TW - 1) / TW;
    g.noise_stride = (int)p.noise_stride_n;
    g.slope = (p.act == 3) ? p.alpha : 1.f;
    g.gain = p.gain;
    g.hi = (p.clamp >= 0.f) ? p.clamp : __builtin_inff();
    const int64_t units = (int64_t)g.tiles_a * g.tiles_b * p.n * ((p.co + BM - 1) / BM);
    SG_REQUIRE(units <= INT32_MAX - 8, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    g.units = (int)units;
    if (conv3x3_resident(p))
        return (p.dtype == SG_BF16) ? launch_resident_for<__hip_bfloat16>(g, units, p.ci, p.co, p.in_step, stream)
                                    : launch_resident_for<__half>(g, units, p.ci, p.co, p.in_step, stream);
    if (p.dtype == SG_BF16) return launch_wide_or_not<__hip_bfloat16>(g, p.in_step, stream);
    return launch_wide_or_not<__half>(g, p.in_step, stream);
}

}  // namespace sg
