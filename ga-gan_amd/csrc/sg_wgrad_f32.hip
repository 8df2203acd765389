// Weight gradient of float32 1x1 and 3x3 convolutions (the low-resolution layers of a mixed-precision network):
//   dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * g_scale[n,cg] * x[n,cx,a*step+dy0+ky,b*step+dx0+kx] * x_scale[n,cx]
// GEMM view per tap: M = cg, N = cx, K = pixels.  Same contract as conv2d_wgrad_kernel of
// sg_conv2d.hip, which sg_conv2d_wgrad falls back to for narrow operands.
//
// The products run on the half-precision matrix instruction without giving up float accuracy, as in the float
// path of sg_conv2d.hip: each operand tensor is multiplied by the power of two that brings its largest magnitude
// (measured on the device in front of the kernel, sg_absmax.h) to 2^14 .. 2^15, every element is cut into two
// halves h0 + h1 that carry 22 of its 24 significant bits, and a product is accumulated in float from h0 h0',
// h0 h1' and h1 h0'.
//
// A workgroup (4 waves, one per SIMD, the CU to itself) owns a 64 x 64 (cg x cx) block of dw for all taps and
// walks the pixel tiles assigned to it; each wave accumulates a 32 x 32 sub-block (2 x 2 MFMA tiles x up to 9
// taps = 144 accumulator registers) over every pixel of a tile.  Both operands are staged [pixel][plane][channel]
// (288-byte rows: two planes of 64 halves + 32 pad).  The matrix instruction wants, per lane, 8 consecutive K
// (pixel) values of one channel: ds_read_b64_tr_b16 delivers 4 pixel rows x 16 channels transposed, so a fragment
// is two such reads.  K slot -> pixel of a 32-pixel step is chosen so that the 8 rows one 32-lane half reads are
// consecutive (lane group q reads pixels 4q .. 4q+3 and 16+4q .. 16+4q+3): with rows 32 bytes apart modulo 256
// every read is free of bank conflicts for unit steps.  A tile is tw x th pixels of imgs images with tw in
// {8, 16, 32}; a step covers 32 / tw rows.  Partial sums go to dw with float atomics.
#include <algorithm>

#include "sg_absmax.h"
#include "sg_common.h"
#include "sg_conv3x3.h"

namespace sg {
namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BC = 64;             // channels of either operand per workgroup
constexpr int kRowBytes = 288;     // bytes per LDS pixel row
constexpr int kMaxPixelRows = 560; // x footprint + g tile, in pixel rows: 157.5 KB

struct WTile {
    int tw, th, imgs;             // gradient pixels per tile: tw (8, 16 or 32) x th (a multiple of 32 / tw) x imgs
    int lw, lh;                   // x footprint of one image's tile
    int tiles_a, tiles_b, tiles_img;
};

__device__ __forceinline__ int floor_div4(int a) { return (a >= 0) ? a / 4 : -((-a + 3) / 4); }

// One float (already scaled into a half's range) -> its two half pieces, at `dst` and 128 bytes further.
__device__ __forceinline__ void store_split(unsigned char* dst, float v) {
    const _Float16 h0 = (_Float16)v;
    *reinterpret_cast<_Float16*>(dst) = h0;
    *reinterpret_cast<_Float16*>(dst + 128) = (_Float16)(v - (float)h0);
}

// Copy rows [iy0, iy0 + lh) x columns [ix0, ix0 + lw) of channels c0 .. c0 + 63 and images img0 .. img0 + imgs - 1
// of a dense NCHW float tensor into LDS as [image][row][column][plane][channel] halves, times scale[n, c] (when
// given) and the power of two `mul`; zero outside the tensor.  A lane owns one channel and moves aligned 16-byte
// packets (VEC) or single floats.
template <bool VEC>
__device__ __forceinline__ void stage(unsigned char* s_dst, const float* __restrict__ x, const float* __restrict__ scale, float mul,
                                      int n_total, int c_total, int h, int w, int c0, int img0, int imgs, int iy0, int ix0, int lh,
                                      int lw) {
    const int64_t plane = (int64_t)h * w;
    const int ch = threadIdx.x & 63, c = c0 + ch;
    if constexpr (VEC) {
        const int vx0 = floor_div4(ix0);
        const int nv = floor_div4(ix0 + lw - 1) - vx0 + 1;
        const int items = imgs * lh * nv;
        constexpr int U = 12;   // packets in flight per lane: the loop is bound by memory latency, not by bandwidth
        for (int it0 = threadIdx.x >> 6; it0 < items; it0 += (kThreads / 64) * U) {
            float4 val[U];
            int dst[U], rx0[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int it = it0 + u * (kThreads / 64);
                const int v = it % nv;
                const int r = it / nv;
                const int ry = r % lh, ri = r / lh;
                const int iy = iy0 + ry, n = img0 + ri, px0 = (vx0 + v) * 4;
                val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool inside = it < items && iy >= 0 && iy < h && n < n_total && px0 >= 0 && px0 < w && c < c_total;
                if (inside) val[u] = *reinterpret_cast<const float4*>(x + ((int64_t)n * c_total + c) * plane + (int64_t)iy * w + px0);
                dst[u] = (it < items) ? ((ri * lh + ry) * lw) * kRowBytes + ch * 2 : -1;
                rx0[u] = px0 - ix0;
                float s = mul;
                if (scale && inside) s *= scale[(int64_t)n * c_total + c];
                val[u].x *= s; val[u].y *= s; val[u].z *= s; val[u].w *= s;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (dst[u] < 0) continue;
                unsigned char* row = s_dst + dst[u];
                const float e[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (rx0[u] + j >= 0 && rx0[u] + j < lw) store_split(row + (rx0[u] + j) * kRowBytes, e[j]);
            }
        }
    } else {
        const int items = imgs * lh * lw;
        for (int it = threadIdx.x >> 6; it < items; it += kThreads / 64) {
            const int rx = it % lw;
            const int r = it / lw;
            const int ry = r % lh, ri = r / lh;
            const int iy = iy0 + ry, ix = ix0 + rx, n = img0 + ri;
            float val = 0.f;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w && n < n_total && c < c_total) {
                val = x[((int64_t)n * c_total + c) * plane + (int64_t)iy * w + ix] * mul;
                if (scale) val *= scale[(int64_t)n * c_total + c];
            }
            store_split(s_dst + (int64_t)it * kRowBytes + ch * 2, val);
        }
    }
}

// 4 pixel rows x 16 channels of halves, transposed: this lane's channel, the 4 rows in its elements.
__device__ __forceinline__ f16x4_t read_tr(const unsigned char* p) {
    return __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p)));
}

// The 8 K values of one operand fragment: pixels of the step's first and second half.
__device__ __forceinline__ f16x8_t fragment(const unsigned char* first, const unsigned char* second) {
    const f16x4_t lo = read_tr(first), hi = read_tr(second);
    return f16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NK>   // NK x NK taps
__global__ void __launch_bounds__(kThreads, 1) wgrad_f32_kernel(sg_conv2d_wgrad_params p, WTile t, int tiles_total, int x_vec,
                                                                int g_vec, const unsigned* amax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_x = smem;
    unsigned char* s_g = s_x + (size_t)t.imgs * t.lh * t.lw * kRowBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * BC, cx0 = blockIdx.z * BC;
    constexpr int MAX_TAPS = NK * NK;
    const float* __restrict__ x = (const float*)p.x;
    const float* __restrict__ g = (const float*)p.g;

    float mul_x, inv_x, mul_g, inv_g;
    half_range_scale(amax, 0, 2, p.x_scale != nullptr, mul_x, inv_x);
    half_range_scale(amax, 1, 3, p.g_scale != nullptr, mul_g, inv_g);

    f32x4_t acc[MAX_TAPS][2][2];
#pragma unroll
    for (int tp = 0; tp < MAX_TAPS; tp++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // lane constants of the transposed reads: lane 16 q + 4 r + c4 supplies pixel row r of its group's block
    // (pixels 4 q + r and 16 + 4 q + r of the 32-pixel step), channels 4 c4 .. 4 c4 + 3 of the fragment's 16
    int g_lane[2], x_lane[2];
    {
        const int q = lane >> 4, r = (lane >> 2) & 3, c4 = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int kp = 16 * j + 4 * q + r;
            const int dr = kp / t.tw, dc = kp - dr * t.tw;
            g_lane[j] = (dr * t.tw + dc) * kRowBytes + c4 * 8 + (wave >> 1) * 64;
            x_lane[j] = (dr * p.step * t.lw + dc * p.step) * kRowBytes + c4 * 8 + (wave & 1) * 64;
        }
    }
    const int rows_per_step = 32 / t.tw;

    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        int rest = tile;
        const int tb = rest % t.tiles_b; rest /= t.tiles_b;
        const int ta = rest % t.tiles_a; rest /= t.tiles_a;
        const int img0 = rest * t.imgs;
        const int a0 = ta * t.th, b0 = tb * t.tw;
        __syncthreads();
        if (x_vec) stage<true>(s_x, x, p.x_scale, mul_x, p.n, p.cx, p.h, p.w_in, cx0, img0, t.imgs, a0 * p.step + p.dy0, b0 * p.step + p.dx0, t.lh, t.lw);
        else stage<false>(s_x, x, p.x_scale, mul_x, p.n, p.cx, p.h, p.w_in, cx0, img0, t.imgs, a0 * p.step + p.dy0, b0 * p.step + p.dx0, t.lh, t.lw);
        if (g_vec) stage<true>(s_g, g, p.g_scale, mul_g, p.n, p.cg, p.gh, p.gw, cg0, img0, t.imgs, a0, b0, t.th, t.tw);
        else stage<false>(s_g, g, p.g_scale, mul_g, p.n, p.cg, p.gh, p.gw, cg0, img0, t.imgs, a0, b0, t.th, t.tw);
        __syncthreads();

        for (int ri = 0; ri < t.imgs; ri++) {
            for (int ra = 0; ra < t.th; ra += rows_per_step) {
                const unsigned char* gr = s_g + ((ri * t.th + ra) * t.tw) * kRowBytes;
                const unsigned char* xr = s_x + ((ri * t.lh + ra * p.step) * t.lw) * kRowBytes;
                // a[m][plane], b[n][plane]; the fragments of the next tap are fetched while the MFMAs of this one run
                f16x8_t a[2][2], b[2][2];
                auto fetch_b = [&](int tp, f16x8_t (&dst)[2][2]) {
                    const unsigned char* xt = xr + ((tp / NK) * t.lw + (tp % NK)) * kRowBytes;
#pragma unroll
                    for (int n = 0; n < 2; n++)
#pragma unroll
                        for (int pl = 0; pl < 2; pl++)
                            dst[n][pl] = fragment(xt + x_lane[0] + n * 32 + pl * 128, xt + x_lane[1] + n * 32 + pl * 128);
                };
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int pl = 0; pl < 2; pl++)
                        a[m][pl] = fragment(gr + g_lane[0] + m * 32 + pl * 128, gr + g_lane[1] + m * 32 + pl * 128);
                fetch_b(0, b);
#pragma unroll
                for (int tp = 0; tp < MAX_TAPS; tp++) {
                    f16x8_t bn[2][2];
                    if (tp + 1 < MAX_TAPS) fetch_b(tp + 1, bn);
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int n = 0; n < 2; n++) {
                            f32x4_t c = acc[tp][m][n];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][1], b[n][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][0], b[n][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][0], b[n][0], c, 0, 0, 0);
                            acc[tp][m][n] = c;
                        }
                    if (tp + 1 < MAX_TAPS) {
#pragma unroll
                        for (int n = 0; n < 2; n++)
#pragma unroll
                            for (int pl = 0; pl < 2; pl++) b[n][pl] = bn[n][pl];
                    }
                }
            }
        }
    }

    // ---- add into dw[tap][cg][cx]: lanes of a row run along cx ------------------------------------------------
    float* __restrict__ dw = p.dw;
#pragma unroll
    for (int tp = 0; tp < MAX_TAPS; tp++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int cg = cg0 + (wave >> 1) * 32 + m * 16 + lgrp * 4 + r;
                    const int cx = cx0 + (wave & 1) * 32 + n * 16 + lrow;
                    if (cg < p.cg && cx < p.cx) atomicAdd(dw + ((int64_t)tp * p.cg + cg) * p.cx + cx, acc[tp][m][n][r] * inv_x * inv_g);
                }
}

WTile make_wtile(const sg_conv2d_wgrad_params& p) {
    WTile t;
    t.tw = p.gw > 16 ? 32 : (p.gw > 8 ? 16 : 8);
    t.lw = (t.tw - 1) * p.step + p.nkx;
    const int rps = 32 / t.tw;   // rows per 32-pixel step
    t.th = std::min((p.gh + rps - 1) / rps * rps, 16);
    auto rows = [&](int th, int imgs) { return imgs * (((th - 1) * p.step + p.nky) * t.lw + th * t.tw); };
    while (t.th > rps && rows(t.th, 1) > kMaxPixelRows) t.th -= rps;
    t.imgs = 1;
    while (t.imgs < 8 && t.imgs < p.n && rows(t.th, t.imgs + 1) <= kMaxPixelRows) t.imgs++;
    t.lh = (t.th - 1) * p.step + p.nky;
    t.tiles_b = (p.gw + t.tw - 1) / t.tw;
    t.tiles_a = (p.gh + t.th - 1) / t.th;
    t.tiles_img = (p.n + t.imgs - 1) / t.imgs;
    return t;
}

}  // namespace

bool wgrad_f32_eligible(const sg_conv2d_wgrad_params& p) {
    if (p.dtype != SG_F32 || p.cg < 32 || p.cx < 32 || p.nky != p.nkx || (p.nky != 1 && p.nky != 3)) return false;
    const WTile t = make_wtile(p);
    return (t.imgs * (t.lh * t.lw + t.th * t.tw)) <= kMaxPixelRows;   // (a single step of a very wide strided layer may not fit)
}

int wgrad_f32_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    const WTile t = make_wtile(p);
    const int64_t tiles = (int64_t)t.tiles_a * t.tiles_b * t.tiles_img;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: too many tiles");
    const int blocks_g = (p.cg + BC - 1) / BC, blocks_x = (p.cx + BC - 1) / BC;
    // one or two workgroups per CU: the pixel tiles are divided over `split` workgroups per block of dw, each of
    // which ends in taps x 64 x 64 atomic additions
    int64_t split = std::max<int64_t>(1, 512 / ((int64_t)blocks_g * blocks_x));
    split = std::min<int64_t>(split, tiles);
    const size_t lds = (size_t)t.imgs * (t.lh * t.lw + t.th * t.tw) * kRowBytes;
    auto kernel = (p.nky == 1) ? wgrad_f32_kernel<1> : wgrad_f32_kernel<3>;
    if (lds > 64 * 1024) {
        hipError_t e = allow_large_lds((const void*)kernel);
        if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    const unsigned* amax = nullptr;
    if (int rc = absmax_measure(stream, (const float*)p.x, (int64_t)p.n * p.cx * p.h * p.w_in, (const float*)p.g,
                                (int64_t)p.n * p.cg * p.gh * p.gw, p.x_scale, (int64_t)p.n * p.cx, p.g_scale, (int64_t)p.n * p.cg, &amax))
        return rc;
    const int x_vec = (p.w_in % 4 == 0) && aligned16(p.x), g_vec = (p.gw % 4 == 0) && aligned16(p.g);
    hipLaunchKernelGGL(kernel, dim3((unsigned)split, (unsigned)blocks_g, (unsigned)blocks_x), dim3(kThreads), lds, stream, p, t,
                       (int)tiles, x_vec, g_vec, amax);
    return check_launch("sg_conv2d_wgrad (f32)");
}

}  // namespace sg
