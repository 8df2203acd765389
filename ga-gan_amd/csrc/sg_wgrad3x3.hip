// Weight gradient of the 3x3, unit-stride convolution of 16-bit tensors:
//   dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * g_scale[n,cg] * x[n,cx,a+dy0+ky,b+dx0+kx] * x_scale[n,cx]
// GEMM view per tap: M = cg, N = cx, K = pixels.  Same contract as conv2d_wgrad_kernel of
// sg_conv2d.hip, which sg_conv2d_wgrad falls back to for every other geometry.
//
// A workgroup (the CU to itself: 95-114 KB of LDS; 4 waves, one per SIMD, with up to 512 registers per lane at
// stride 2, 8 waves at unit stride - see WAVES below) owns a 64 x 64 (cg x cx) block of dw for all 9 taps and walks the pixel tiles (8 x 32
// pixels of one image) assigned to it; each wave accumulates a 32 x 32 sub-block (2 x 2 MFMA
// tiles x 9 taps = 144 accumulator registers) over all 256 pixels of a tile (over a share of them where a
// block side has only 32 real channels and two or four waves work on the same sub-block).  Both operands are
// staged [pixel][channel] (160-byte rows: 128 payload + 32 pad), and the fragments - K = pixels
// runs down the rows - are fetched with the transposing LDS read, which makes the tap shift a
// row offset.  The packets of the next tile are loaded into registers before the multiply phase
// of the current one and committed to LDS after it.  Partial sums go to dw with float atomics.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <type_traits>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;

constexpr int kThreads = 256;
constexpr int TW = 32;
constexpr int BC = 64;                           // channels of either operand per workgroup
constexpr int kPitch = 160;                      // bytes per LDS pixel row

// Tile of gradient pixels and the input footprint it needs, for convolution stride STEP (1 or 2).
template <int STEP> struct WGeo {
    static constexpr int TH = (STEP == 1) ? 8 : 4;             // gradient tile rows (x TW columns)
    static constexpr int LH = (TH - 1) * STEP + 3;             // input footprint: 10 x 34 / 9 x 65 pixels
    static constexpr int LW = (TW - 1) * STEP + 3;
    static constexpr int NV = (LW + 7 + 7) / 8;                // 8-pixel packets covering a footprint row at any alignment
    static constexpr int LWP = NV * 8;                         // pixels per image row: whole packets (48 / 72), so that a commit needs no column test
    static constexpr int kXBytes = LH * LWP * kPitch;          // 76800 / 103680
    static constexpr int kGBytes = TH * TW * kPitch;           // 40960 / 20480
    static constexpr int X_STEPS = (LH * NV + 7) / 8;          // (row, packet) items per lane: 8 / 11
    static constexpr int G_STEPS = TH * (TW / 8) / 8;          // 4 / 2
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

// Buffer addressing: a byte offset at or beyond the descriptor's bound reads zeros, so a packet outside the image (or a
// channel beyond the last) costs one select on its offset when the tile is decoded, and nothing afterwards.
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

// 8 consecutive K values (pixels) of one channel for this lane: two transposing reads of 4 pixel rows each
__device__ __forceinline__ uint4 fetch_frag(const unsigned char* lo_addr, const unsigned char* hi_addr) {
    const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)lo_addr);
    const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)hi_addr);
    uint4 r;
    r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
    r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
    return r;
}

struct Wgrad3x3Args {
    const void* x; const void* g; float* dw;
    const float* x_scale; const float* g_scale;
    int n, cx, h, wid, cg, gh, gw;
    int dy0, px_shift, px_col;     // as in the forward kernel: packets start at b0 + px_shift, footprint column px_col inside them
    int tiles_a, tiles_b, tiles_total;
};

// WAVES = 8: waves 4..7 work on the sub-blocks of waves 0..3 and take every other k-step of theirs (two waves per
// SIMD: one's LDS reads and staging arithmetic run under the other's multiplies; the partial sums meet in the atomics).
template <class T, int STEP, int WAVES, bool XS, bool GS>   // XS / GS: x / g carry a per-sample channel scale
__global__ void __launch_bounds__(64 * WAVES, 1) wgrad3x3_kernel(Wgrad3x3Args p) {
    using G = WGeo<STEP>;
    constexpr int TH = G::TH, LH = G::LH, NV = G::NV, LWP = G::LWP;
    constexpr int ROWS = 2 * WAVES;                                   // item rows staged per step (32 lanes per row)
    constexpr int X_ITEMS = LH * NV, G_ITEMS = TH * (TW / 8);
    constexpr int X_STEPS = (X_ITEMS + ROWS - 1) / ROWS, G_STEPS = (G_ITEMS + ROWS - 1) / ROWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_x = smem;
    unsigned char* s_g = smem + G::kXBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, k_half = tid >> 8;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * BC, cx0 = blockIdx.z * BC;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ g = (const T*)p.g;
    const int x_plane = p.h * p.wid, g_plane = p.gh * p.gw;
    const unsigned x_plane_bytes = (unsigned)x_plane * 2u, g_plane_bytes = (unsigned)g_plane * 2u;
    const int x_left = min(BC, p.cx - cx0), g_left = min(BC, p.cg - cg0);   // real channels of this block

    // ---- fetch / commit: item = (channel pair cp = tid & 31, (row, packet) index r) ------------------------------
    // What does not depend on the tile is decided here: an item's row and first pixel inside the footprint, its
    // LDS address, whether it exists.  Per tile a lane decides once per item whether its packet lies in the image.
    const int cp = tid & 31, r0 = tid >> 5;
    int x_ry[X_STEPS], x_px[X_STEPS], g_ra[G_STEPS], g_pb[G_STEPS];
#pragma unroll
    for (int s = 0; s < X_STEPS; s++) {
        const int r = r0 + ROWS * s;
        x_ry[s] = (r < X_ITEMS) ? r / NV : (1 << 20);     // an item that does not exist lies below any image
        x_px[s] = (r % NV) * 8;
    }
#pragma unroll
    for (int s = 0; s < G_STEPS; s++) {
        const int r = r0 + ROWS * s;
        g_ra[s] = (r < G_ITEMS) ? (r >> 2) : (1 << 20);
        g_pb[s] = (r & 3) * 8;
    }
    const bool x_last = (X_ITEMS % ROWS) == 0 || r0 + ROWS * (X_STEPS - 1) < X_ITEMS;
    const bool g_last = (G_ITEMS % ROWS) == 0 || r0 + ROWS * (G_STEPS - 1) < G_ITEMS;
    uint4 xa[X_STEPS], xb[X_STEPS], ga[G_STEPS], gb[G_STEPS];
    float xs0 = 1.f, xs1 = 1.f, gs0 = 1.f, gs1 = 1.f;

    // The loads of a tile: X_STEPS + G_STEPS pairs of packets per lane (12 at unit stride, 26 at stride 2).  The texture
    // path of a CU takes 16 cycles per 16-byte load instruction of a wave, so a tile's loads keep it busy for 1,500 -
    // 1,700 cycles: issued as one batch they queue there while nothing multiplies.  Only the first tile is fetched as
    // a batch; every other tile's loads are dealt over the k-steps of the multiply phase before it (fetch_part).
    constexpr int PARTS = X_STEPS + G_STEPS;
    constexpr int K_STEPS = TH / (WAVES / 4);                      // k-steps of a wave per tile (fewer on narrow blocks)
    constexpr int PER_STEP = (PARTS + K_STEPS - 1) / K_STEPS;      // parts issued beside one k-step
    rsrc_t d_x = make_rsrc(x, 0u), d_g = make_rsrc(x, 0u);
    int f_iy = 0, f_px = 0, f_a0 = 0, f_b0 = 0, f_n = 0;
    auto fetch_begin = [&](int tile) {
        const int tb = tile % p.tiles_b; tile /= p.tiles_b;
        const int ta = tile % p.tiles_a;
        f_n = tile / p.tiles_a;
        f_a0 = ta * TH; f_b0 = tb * TW;
        d_x = make_rsrc(x + (size_t)(f_n * p.cx + cx0) * x_plane, (unsigned)x_left * x_plane_bytes);
        d_g = make_rsrc(g + (size_t)(f_n * p.cg + cg0) * g_plane, (unsigned)g_left * g_plane_bytes);
        f_iy = f_a0 * STEP + p.dy0; f_px = f_b0 * STEP + p.px_shift;
    };
    auto fetch_part = [&](int part) {                              // part: a compile-time constant at every call
        if (part < X_STEPS) {
            const int s = part < X_STEPS ? part : 0;
            const int iy = f_iy + x_ry[s], px0 = f_px + x_px[s];
            const bool ok = iy >= 0 && iy < p.h && px0 >= 0 && px0 < p.wid;
            const unsigned voff = ok ? (unsigned)(2 * cp * x_plane + iy * p.wid + px0) * 2u : kOutside;
            xa[s] = load16(d_x, voff, 0);
            xb[s] = load16(d_x, voff, x_plane_bytes);              // channel 2 cp + 1 (channel counts are multiples of 8)
            if constexpr (XS) {
                if (part == 0) {
                    const rsrc_t rs = make_rsrc(p.x_scale + (size_t)f_n * p.cx + cx0, (unsigned)x_left * 4u);
                    xs0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cp * 8, 0, 0));
                    xs1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cp * 8, 4, 0));
                }
            }
        } else {
            const int s = part >= X_STEPS ? part - X_STEPS : 0;
            const int a = f_a0 + g_ra[s], b = f_b0 + g_pb[s];
            const bool ok = a < p.gh && b < p.gw;
            const unsigned voff = ok ? (unsigned)(2 * cp * g_plane + a * p.gw + b) * 2u : kOutside;
            ga[s] = load16(d_g, voff, 0);
            gb[s] = load16(d_g, voff, g_plane_bytes);
            if constexpr (GS) {
                if (part == X_STEPS) {
                    const rsrc_t rs = make_rsrc(p.g_scale + (size_t)f_n * p.cg + cg0, (unsigned)g_left * 4u);
                    gs0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cp * 8, 0, 0));
                    gs1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cp * 8, 4, 0));
                }
            }
        }
    };
    auto fetch = [&](int tile) {
        fetch_begin(tile);
#pragma unroll
        for (int part = 0; part < PARTS; part++) fetch_part(part);
    };

    // pixels 0 .. 7 of one channel pair -> rows dst + j * pitch (the image rows hold whole packets: no column test)
    auto put8 = [&](unsigned char* dst, const uint4& a, const uint4& b, auto scaled, float s0, float s1) {
        const unsigned wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int jp = 0; jp < 4; jp++) {
            unsigned e0, e1;
            if constexpr (decltype(scaled)::value) {
                e0 = Pair<T>::pack(Pair<T>::lo(wa[jp]) * s0, Pair<T>::lo(wb[jp]) * s1);
                e1 = Pair<T>::pack(Pair<T>::hi(wa[jp]) * s0, Pair<T>::hi(wb[jp]) * s1);
            } else {
                e0 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x05040100u);
                e1 = __builtin_amdgcn_perm(wb[jp], wa[jp], 0x07060302u);
            }
            *reinterpret_cast<unsigned*>(dst + (2 * jp) * kPitch) = e0;
            *reinterpret_cast<unsigned*>(dst + (2 * jp + 1) * kPitch) = e1;
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < X_STEPS; s++) {
            if (s == X_STEPS - 1 && !x_last) continue;
            put8(s_x + (r0 + ROWS * s) * 8 * kPitch + cp * 4, xa[s], xb[s], std::integral_constant<bool, XS>{}, xs0, xs1);   // row ry, packet v: pixel (ry * NV + v) * 8
        }
#pragma unroll
        for (int s = 0; s < G_STEPS; s++) {
            if (s == G_STEPS - 1 && !g_last) continue;
            put8(s_g + (r0 + ROWS * s) * 8 * kPitch + cp * 4, ga[s], gb[s], std::integral_constant<bool, GS>{}, gs0, gs1);   // row ra, packet v: pixel (ra * 4 + v) * 8
        }
    };

    // ---- multiply-phase addressing ------------------------------------------------------------------------------
    // k-step ks = tile row ks (32 pixels).  Lane group g holds k = 8g .. 8g+7 of the step; k -> pixel column
    // 4 g + 16 hf + q for the read half hf and its row q (any bijection does, as both operands use it): the two
    // lane groups one LDS cycle serves then read 8 consecutive rows, which 160-byte rows spread over all banks.
    // Lane 4q + p of a group addresses row q, channels 4p .. 4p+3 of the 16-channel tile.
    // This wave's 32 x 32 sub-block of the 64 x 64 block.  Where the block holds no more than 32 real channels on a
    // side, the waves that would multiply padding take a share of the k-steps of the real sub-block instead (their
    // partial sums meet in the atomics at the end).
    int wm = wave >> 1, wn = wave & 1, ks0 = 0, ks_stride = 1;
    {
        const bool narrow_g = p.cg - cg0 <= 32, narrow_x = p.cx - cx0 <= 32;
        if (narrow_g && narrow_x) { ks0 = wave; ks_stride = 4; wm = wn = 0; }
        else if (narrow_g) { ks0 = wm; ks_stride = 2; wm = 0; }
        else if (narrow_x) { ks0 = wn; ks_stride = 2; wn = 0; }
        if (WAVES == 8) { ks0 = 2 * ks0 + k_half; ks_stride *= 2; }
    }
    const int q = lrow >> 2, pq = lrow & 3;
    int off_g[2], off_x[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int col = 4 * lgrp + 16 * hf + q;
        off_g[hf] = col * kPitch + pq * 8 + wm * 64;   // + m * 32 + ks * TW * kPitch
        off_x[hf] = (col * STEP + p.px_col) * kPitch + pq * 8 + wn * 64;   // + n * 32 + ((ks * STEP + ky) * LWP + kx) * kPitch
    }

    f32x4_t acc[9][2][2];
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int tile = blockIdx.x;
    if (tile < p.tiles_total) fetch(tile);
    for (; tile < p.tiles_total; tile += gridDim.x) {
        __syncthreads();   // everyone is done reading the previous tile
        commit();
        __syncthreads();
        const bool more = tile + (int)gridDim.x < p.tiles_total;
        if (more) fetch_begin(tile + gridDim.x);
        int done = 0;                                   // k-steps this wave has run (uniform)
#pragma unroll
        for (int i = 0; i < K_STEPS; i++) {
            const int ks = ks0 + i * ks_stride;
            if (ks >= TH) break;
            done = i + 1;
            if (more) {                                 // the next tile's loads of this k-step: in flight during the multiplies
#pragma unroll
                for (int part = i * PER_STEP; part < (i + 1) * PER_STEP && part < PARTS; part++) fetch_part(part);
            }
            uint4 a[2];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const unsigned char* base = s_g + ks * TW * kPitch + m * 32;
                a[m] = fetch_frag(base + off_g[0], base + off_g[1]);
            }
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int ky = tp / 3, kx = tp % 3;
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const unsigned char* base = s_x + ((ks * STEP + ky) * LWP + kx) * kPitch + n * 32;
                    const uint4 b = fetch_frag(base + off_x[0], base + off_x[1]);
#pragma unroll
                    for (int m = 0; m < 2; m++) Mma16<T>::run(a[m], b, acc[tp][m][n]);
                }
            }
        }
        if (more && done < K_STEPS) {                   // (a wave that shares its sub-block's k-steps ran fewer of them)
#pragma unroll
            for (int part = 0; part < PARTS; part++)
                if (part >= done * PER_STEP) fetch_part(part);
        }
    }

    // ---- reduce into dw[tap][cg][cx] ----------------------------------------------------------------------------
    float* __restrict__ dw = p.dw;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int cg = cg0 + wm * 32 + m * 16 + lgrp * 4 + r, cx = cx0 + wn * 32 + n * 16 + lrow;
                    if (cg < p.cg && cx < p.cx) atomicAdd(dw + (unsigned)((tp * p.cg + cg) * p.cx + cx), acc[tp][m][n][r]);
                }
}

}  // namespace

bool wgrad3x3_eligible(const sg_conv2d_wgrad_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.step < 1 || p.step > 2 || p.nky != 3 || p.nkx != 3) return false;
    if (p.gw < TW / 2 || p.gh < (p.step == 1 ? 8 : 4)) return false;   // (16 columns: half of every tile is idle, still ahead of the general kernel)
    if (p.w_in % 8 != 0 || p.gw % 8 != 0) return false;
    if (!aligned16(p.x) || !aligned16(p.g)) return false;
    if (p.cx < 32 || p.cg < 32) return false;   // a 64 x 64 block would be mostly padding
    if (p.cx % 8 != 0 || p.cg % 8 != 0) return false;
    if ((int64_t)p.h * p.w_in >= (1 << 23) || (int64_t)p.gh * p.gw >= (1 << 23)) return false;   // 64 planes bound the buffer offsets (< 2^31 bytes)
    return true;
}

template <class T, int STEP, bool XS, bool GS>
static int launch_wstep_scaled(Wgrad3x3Args& a, const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    using G = WGeo<STEP>;
    a.tiles_a = (p.gh + G::TH - 1) / G::TH;
    a.tiles_b = (p.gw + TW - 1) / TW;
    const int64_t tiles = (int64_t)a.tiles_a * a.tiles_b * p.n;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: too many tiles");
    a.tiles_total = (int)tiles;
    const int blocks_g = (p.cg + BC - 1) / BC, blocks_x = (p.cx + BC - 1) / BC;
    // one workgroup per CU: split the pixel tiles so that the (cg, cx) blocks times the split fill the chip
    int64_t split = (256 + (int64_t)blocks_g * blocks_x - 1) / ((int64_t)blocks_g * blocks_x);
    if (split > tiles) split = tiles;
    if (split < 1) split = 1;
    constexpr size_t lds = G::kXBytes + G::kGBytes;
    constexpr int WAVES = (STEP == 1) ? 8 : 4;   // (the strided form with two waves per SIMD: 300-410 against 340-470 TFLOP/s)
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS of a CU");
    auto kernel = wgrad3x3_kernel<T, STEP, WAVES, XS, GS>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL(kernel, dim3((unsigned)split, (unsigned)blocks_g, (unsigned)blocks_x), dim3(64 * WAVES), lds, stream, a);
    return check_launch("sg_conv2d_wgrad");
}

template <class T, int STEP>
static int launch_wstep(Wgrad3x3Args& a, const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    if (p.x_scale && p.g_scale) return launch_wstep_scaled<T, STEP, true, true>(a, p, stream);
    if (p.x_scale) return launch_wstep_scaled<T, STEP, true, false>(a, p, stream);
    if (p.g_scale) return launch_wstep_scaled<T, STEP, false, true>(a, p, stream);
    return launch_wstep_scaled<T, STEP, false, false>(a, p, stream);
}

int wgrad3x3_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    Wgrad3x3Args a;
    a.x = p.x; a.g = p.g; a.dw = p.dw; a.x_scale = p.x_scale; a.g_scale = p.g_scale;
    a.n = p.n; a.cx = p.cx; a.h = p.h; a.wid = p.w_in; a.cg = p.cg; a.gh = p.gh; a.gw = p.gw;
    a.dy0 = p.dy0;
    const int fl = (p.dx0 >= 0) ? p.dx0 / 8 : -((-p.dx0 + 7) / 8);
    a.px_shift = fl * 8;
    a.px_col = p.dx0 - a.px_shift;
    if (p.dtype == SG_BF16)
        return (p.step == 1) ? launch_wstep<__hip_bfloat16, 1>(a, p, stream) : launch_wstep<__hip_bfloat16, 2>(a, p, stream);
    return (p.step == 1) ? launch_wstep<__half, 1>(a, p, stream) : launch_wstep<__half, 2>(a, p, stream);
}

}  // namespace sg
