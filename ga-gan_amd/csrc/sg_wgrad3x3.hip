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
    static constexpr int kXBytes = LH * LW * kPitch;           // 54400 / 93600
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

template <class T> __device__ __forceinline__ float lo16(unsigned w) { return Elem<T>::load(__builtin_bit_cast(T, (unsigned short)(w & 0xffffu))); }
template <class T> __device__ __forceinline__ unsigned pack16(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(a)) |
           ((unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(b)) << 16);
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
template <class T, int STEP, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, 1) wgrad3x3_kernel(Wgrad3x3Args p) {
    using G = WGeo<STEP>;
    constexpr int TH = G::TH, LH = G::LH, LW = G::LW, NV = G::NV;
    constexpr int ROWS = 2 * WAVES;                                   // item rows staged per step (32 lanes per row)
    constexpr int X_STEPS = (LH * NV + ROWS - 1) / ROWS, G_STEPS = (TH * (TW / 8) + ROWS - 1) / ROWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_x = smem;
    unsigned char* s_g = smem + G::kXBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, k_half = tid >> 8;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * BC, cx0 = blockIdx.z * BC;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ g = (const T*)p.g;
    const int x_plane = p.h * p.wid, g_plane = p.gh * p.gw;

    // ---- fetch / commit: item = (channel pair cp = tid & 31, packet, row) --------------------------------------
    const int cp = tid & 31;
    uint4 xa[X_STEPS], xb[X_STEPS], ga[G_STEPS], gb[G_STEPS];
    float xs0 = 1.f, xs1 = 1.f, gs0 = 1.f, gs1 = 1.f;

    auto fetch = [&](int tile) {
        const int tb = tile % p.tiles_b; tile /= p.tiles_b;
        const int ta = tile % p.tiles_a;
        const int n = tile / p.tiles_a;
        const int a0 = ta * TH, b0 = tb * TW;
        {
            const int c = cx0 + 2 * cp;
            const bool c_ok = c < p.cx, two = c + 1 < p.cx;
            const T* base = x + (unsigned)((n * p.cx + c) * x_plane);
#pragma unroll
            for (int s = 0; s < X_STEPS; s++) {
                const int r = (tid >> 5) + ROWS * s;          // item row index = ry * NV + v
                const int v = r % NV, ry = r / NV;
                const int iy = a0 * STEP + p.dy0 + ry, px0 = b0 * STEP + p.px_shift + v * 8;
                xa[s] = make_uint4(0, 0, 0, 0);
                xb[s] = make_uint4(0, 0, 0, 0);
                if (c_ok && ry < LH && iy >= 0 && iy < p.h && px0 >= 0 && px0 < p.wid) {
                    const T* src = base + iy * p.wid + px0;
                    xa[s] = *reinterpret_cast<const uint4*>(src);
                    if (two) xb[s] = *reinterpret_cast<const uint4*>(src + x_plane);
                }
            }
            if (p.x_scale) {
                xs0 = c_ok ? p.x_scale[n * p.cx + c] : 0.f;
                xs1 = two ? p.x_scale[n * p.cx + c + 1] : 0.f;
            }
        }
        {
            const int c = cg0 + 2 * cp;
            const bool c_ok = c < p.cg, two = c + 1 < p.cg;
            const T* base = g + (unsigned)((n * p.cg + c) * g_plane);
#pragma unroll
            for (int s = 0; s < G_STEPS; s++) {
                const int r = (tid >> 5) + ROWS * s;          // = ra * 4 + v
                const int v = r & 3, ra = r >> 2;
                const int a = a0 + ra, b = b0 + v * 8;
                ga[s] = make_uint4(0, 0, 0, 0);
                gb[s] = make_uint4(0, 0, 0, 0);
                if (c_ok && ra < TH && a < p.gh && b < p.gw) {
                    const T* src = base + a * p.gw + b;
                    ga[s] = *reinterpret_cast<const uint4*>(src);
                    if (two) gb[s] = *reinterpret_cast<const uint4*>(src + g_plane);
                }
            }
            if (p.g_scale) {
                gs0 = c_ok ? p.g_scale[n * p.cg + c] : 0.f;
                gs1 = two ? p.g_scale[n * p.cg + c + 1] : 0.f;
            }
        }
    };

    auto put8 = [&](unsigned char* dst_row0, const uint4& a, const uint4& b, bool scaled, float s0, float s1, int first, int count) {
        // pixels first .. first + 7 of one channel pair -> rows dst_row0 + j * pitch, rows outside [0, count) skipped
        const unsigned wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned ea = wa[j >> 1] >> (16 * (j & 1)), eb = wb[j >> 1] >> (16 * (j & 1));
            const unsigned out = scaled ? pack16<T>(lo16<T>(ea) * s0, lo16<T>(eb) * s1) : ((ea & 0xffffu) | (eb << 16));
            const int col = first + j;
            if (col >= 0 && col < count) *reinterpret_cast<unsigned*>(dst_row0 + col * kPitch + cp * 4) = out;
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < X_STEPS; s++) {
            const int r = (tid >> 5) + ROWS * s;
            const int v = r % NV, ry = r / NV;
            if (ry < LH) put8(s_x + ry * LW * kPitch, xa[s], xb[s], p.x_scale != nullptr, xs0, xs1, v * 8 - p.px_col, LW);
        }
#pragma unroll
        for (int s = 0; s < G_STEPS; s++) {
            const int r = (tid >> 5) + ROWS * s;
            const int v = r & 3, ra = r >> 2;
            if (ra >= TH) continue;
            put8(s_g + ra * TW * kPitch, ga[s], gb[s], p.g_scale != nullptr, gs0, gs1, v * 8, TW);
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
        off_x[hf] = col * STEP * kPitch + pq * 8 + wn * 64;   // + n * 32 + ((ks * STEP + ky) * LW + kx) * kPitch
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
        if (tile + (int)gridDim.x < p.tiles_total) fetch(tile + gridDim.x);   // in flight during the multiply phase
#pragma unroll 2
        for (int ks = ks0; ks < TH; ks += ks_stride) {
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
                    const unsigned char* base = s_x + ((ks * STEP + ky) * LW + kx) * kPitch + n * 32;
                    const uint4 b = fetch_frag(base + off_x[0], base + off_x[1]);
#pragma unroll
                    for (int m = 0; m < 2; m++) Mma16<T>::run(a[m], b, acc[tp][m][n]);
                }
            }
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
    return true;
}

template <class T, int STEP>
static int launch_wstep(Wgrad3x3Args& a, const sg_conv2d_wgrad_params& p, hipStream_t stream) {
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
    constexpr int WAVES = (STEP == 1) ? 8 : 4;   // (the strided form spills registers with two waves per SIMD and loses)
    auto kernel = wgrad3x3_kernel<T, STEP, WAVES>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL(kernel, dim3((unsigned)split, (unsigned)blocks_g, (unsigned)blocks_x), dim3(64 * WAVES), lds, stream, a);
    return check_launch("sg_conv2d_wgrad");
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
