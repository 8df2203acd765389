// Weight gradient of float32 1x1 and 3x3 convolutions (the low-resolution layers of a mixed-precision network):
//   dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * g_scale[n,cg] * x[n,cx,a*step+dy0+ky,b*step+dx0+kx] * x_scale[n,cx]
// GEMM view per tap: M = cg, N = cx, K = pixels.  Same contract as conv2d_wgrad_kernel of
// sg_conv2d.hip, which sg_conv2d_wgrad falls back to for narrow operands.
//
// A workgroup (4 waves, one per SIMD, the CU to itself) owns a 64 x 64 (cg x cx) block of dw for
// all taps and walks the pixel tiles assigned to it; each wave accumulates a 32 x 32 sub-block
// (2 x 2 MFMA tiles x up to 9 taps = 144 accumulator registers) over every pixel of a tile, so a
// staged pixel feeds 36 MFMAs per wave instead of the 9 of a 16 x 16 block.  Both operands are
// staged [pixel][channel] as floats (320-byte rows: 256 payload + 64 pad, so that the four pixel
// rows one 16x16x4 fragment read touches fall on different banks).  A tile is tw x th pixels of
// imgs images with tw a multiple of 4: the four pixels of a k-step are neighbours in a row and
// their LDS addresses are a per-lane constant plus a uniform offset.  Partial sums go to dw with
// float atomics.
#include <algorithm>

#include "sg_common.h"
#include "sg_conv3x3.h"

namespace sg {
namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BC = 64;            // channels of either operand per workgroup
constexpr int kPitchF = 80;       // floats per LDS pixel row
constexpr int kMaxPixelRows = 480;   // x footprint + g tile, in pixel rows: 150 KB

struct WTile {
    int tw, th, imgs;             // gradient pixels per tile: tw (multiple of 4) x th x imgs
    int lw, lh;                   // x footprint of one image's tile
    int tiles_a, tiles_b, tiles_img;
};

__device__ __forceinline__ int floor_div4(int a) { return (a >= 0) ? a / 4 : -((-a + 3) / 4); }

// Copy rows [iy0, iy0 + lh) x columns [ix0, ix0 + lw) of channels c0 .. c0 + 63 and images img0 .. img0 + imgs - 1
// of a dense NCHW float tensor into LDS as [image][row][column][channel], times scale[n, c] when given; zero
// outside the tensor.  A lane owns one channel and moves aligned 16-byte packets (VEC) or single floats.
template <bool VEC>
__device__ __forceinline__ void stage(float* s_dst, const float* __restrict__ x, const float* __restrict__ scale, int n_total,
                                      int c_total, int h, int w, int c0, int img0, int imgs, int iy0, int ix0, int lh, int lw) {
    const int64_t plane = (int64_t)h * w;
    const int ch = threadIdx.x & 63, c = c0 + ch;
    if constexpr (VEC) {
        const int vx0 = floor_div4(ix0);
        const int nv = floor_div4(ix0 + lw - 1) - vx0 + 1;
        const int items = imgs * lh * nv;
        constexpr int U = 8;   // packets in flight per lane: the loop is bound by memory latency, not by bandwidth
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
                if (it < items && iy >= 0 && iy < h && n < n_total && px0 >= 0 && px0 < w && c < c_total)
                    val[u] = *reinterpret_cast<const float4*>(x + ((int64_t)n * c_total + c) * plane + (int64_t)iy * w + px0);
                dst[u] = (it < items) ? ((ri * lh + ry) * lw) * kPitchF + ch : -1;
                rx0[u] = px0 - ix0;
                if (scale && it < items && n < n_total && c < c_total) {
                    const float s = scale[(int64_t)n * c_total + c];
                    val[u].x *= s; val[u].y *= s; val[u].z *= s; val[u].w *= s;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (dst[u] < 0) continue;
                float* row = s_dst + dst[u];
                const float e[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (rx0[u] + j >= 0 && rx0[u] + j < lw) row[(rx0[u] + j) * kPitchF] = e[j];
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
                val = x[((int64_t)n * c_total + c) * plane + (int64_t)iy * w + ix];
                if (scale) val *= scale[(int64_t)n * c_total + c];
            }
            s_dst[(int64_t)it * kPitchF + ch] = val;
        }
    }
}

template <int NK>   // NK x NK taps
__global__ void __launch_bounds__(kThreads, 1) wgrad_f32_kernel(sg_conv2d_wgrad_params p, WTile t, int tiles_total, int x_vec,
                                                                int g_vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_x = reinterpret_cast<float*>(smem);
    float* s_g = s_x + (size_t)t.imgs * t.lh * t.lw * kPitchF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * BC, cx0 = blockIdx.z * BC;
    constexpr int MAX_TAPS = NK * NK;
    const float* __restrict__ x = (const float*)p.x;
    const float* __restrict__ g = (const float*)p.g;

    f32x4_t acc[MAX_TAPS][2][2];
#pragma unroll
    for (int tp = 0; tp < MAX_TAPS; tp++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // lane constants: the pixel of the k-step this lane group supplies, and the channel column of its fragments
    const int g_lane = lgrp * kPitchF + (wave >> 1) * 32 + lrow;
    const int x_lane = lgrp * p.step * kPitchF + (wave & 1) * 32 + lrow;

    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        int rest = tile;
        const int tb = rest % t.tiles_b; rest /= t.tiles_b;
        const int ta = rest % t.tiles_a; rest /= t.tiles_a;
        const int img0 = rest * t.imgs;
        const int a0 = ta * t.th, b0 = tb * t.tw;
        __syncthreads();
        if (x_vec) stage<true>(s_x, x, p.x_scale, p.n, p.cx, p.h, p.w_in, cx0, img0, t.imgs, a0 * p.step + p.dy0, b0 * p.step + p.dx0, t.lh, t.lw);
        else stage<false>(s_x, x, p.x_scale, p.n, p.cx, p.h, p.w_in, cx0, img0, t.imgs, a0 * p.step + p.dy0, b0 * p.step + p.dx0, t.lh, t.lw);
        if (g_vec) stage<true>(s_g, g, p.g_scale, p.n, p.cg, p.gh, p.gw, cg0, img0, t.imgs, a0, b0, t.th, t.tw);
        else stage<false>(s_g, g, p.g_scale, p.n, p.cg, p.gh, p.gw, cg0, img0, t.imgs, a0, b0, t.th, t.tw);
        __syncthreads();

        for (int ri = 0; ri < t.imgs; ri++) {
            for (int ra = 0; ra < t.th; ra++) {
                const float* gr = s_g + ((ri * t.th + ra) * t.tw) * kPitchF + g_lane;
                const float* xr = s_x + ((ri * t.lh + ra * p.step) * t.lw) * kPitchF + x_lane;
                // fragments of the next k-step are fetched while the MFMAs of the current one run
                float av[2], bv[MAX_TAPS][2];
                auto fetch = [&](int rb, float (&a)[2], float (&b)[MAX_TAPS][2]) {
                    a[0] = gr[rb * kPitchF];
                    a[1] = gr[rb * kPitchF + 16];
                    const float* xs = xr + rb * p.step * kPitchF;
#pragma unroll
                    for (int tp = 0; tp < MAX_TAPS; tp++) {
                        const float* xt = xs + ((tp / NK) * t.lw + (tp % NK)) * kPitchF;
                        b[tp][0] = xt[0];
                        b[tp][1] = xt[16];
                    }
                };
                fetch(0, av, bv);
                for (int rb = 0; rb < t.tw; rb += 4) {
                    float an[2], bn[MAX_TAPS][2];
                    fetch((rb + 4 < t.tw) ? rb + 4 : rb, an, bn);   // (the last step re-reads its own: no branch)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tp = 0; tp < MAX_TAPS; tp++) {
                        acc[tp][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[tp][0], acc[tp][0][0], 0, 0, 0);
                        acc[tp][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[tp][1], acc[tp][0][1], 0, 0, 0);
                        acc[tp][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[tp][0], acc[tp][1][0], 0, 0, 0);
                        acc[tp][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[tp][1], acc[tp][1][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    av[0] = an[0]; av[1] = an[1];
#pragma unroll
                    for (int tp = 0; tp < MAX_TAPS; tp++) { bv[tp][0] = bn[tp][0]; bv[tp][1] = bn[tp][1]; }
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
                    if (cg < p.cg && cx < p.cx) atomicAdd(dw + ((int64_t)tp * p.cg + cg) * p.cx + cx, acc[tp][m][n][r]);
                }
}

WTile make_wtile(const sg_conv2d_wgrad_params& p) {
    WTile t;
    t.tw = std::min((p.gw + 3) / 4 * 4, 32);
    t.lw = (t.tw - 1) * p.step + p.nkx;
    t.th = std::min(p.gh, 16);
    auto rows = [&](int th, int imgs) { return imgs * (((th - 1) * p.step + p.nky) * t.lw + th * t.tw); };
    while (t.th > 1 && rows(t.th, 1) > kMaxPixelRows) t.th = (t.th + 1) / 2;
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
    return (t.imgs * (t.lh * t.lw + t.th * t.tw)) <= kMaxPixelRows;   // (a single row of a very wide strided layer may not fit)
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
    const size_t lds = (size_t)t.imgs * (t.lh * t.lw + t.th * t.tw) * kPitchF * sizeof(float);
    auto kernel = (p.nky == 1) ? wgrad_f32_kernel<1> : wgrad_f32_kernel<3>;
    if (lds > 64 * 1024) {
        hipError_t e = allow_large_lds((const void*)kernel);
        if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    const int x_vec = (p.w_in % 4 == 0) && aligned16(p.x), g_vec = (p.gw % 4 == 0) && aligned16(p.g);
    hipLaunchKernelGGL(kernel, dim3((unsigned)split, (unsigned)blocks_g, (unsigned)blocks_x), dim3(kThreads), lds, stream, p, t,
                       (int)tiles, x_vec, g_vec);
    return check_launch("sg_conv2d_wgrad (f32)");
}

}  // namespace sg
