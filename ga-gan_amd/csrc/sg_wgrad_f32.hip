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

// Copy rows [iy0, iy0 + lh) x columns [ix0, ix0 + lw) of chunks cb0 and cb0 + 1 (64 channels) and images img0 ..
// img0 + imgs - 1 of a tensor given as split rows (sg_absmax.h: per chunk of 32 channels and pixel 128 bytes,
// [plane 0][plane 1]) into LDS as [image][row][column] rows of kRowBytes; zero outside the tensor.  16-byte pieces,
// U per lane; `load` only issues the requests (so that the caller can put both operands' in flight before it waits
// for either), `store` writes them to LDS and copies what lies beyond the register window (rare geometries).
template <int U> struct Stager {
    uint4 val[U];
    bool keep[U];
    const unsigned char* xs;
    int64_t lane_off;
    int chunks, n_total, h, w, img0, iy0, ix0, lh, lw, items, piece;
    bool chunk_ok;

    __device__ __forceinline__ void setup(const unsigned char* xs_, int chunks_, int cb0, int n_total_, int h_, int w_, int imgs,
                                          int lh_, int lw_) {
        xs = xs_; chunks = chunks_; n_total = n_total_; h = h_; w = w_; lh = lh_; lw = lw_;
        items = imgs * lh * lw * 16;
        piece = threadIdx.x & 15;
        chunk_ok = cb0 + (piece >> 3) < chunks;
        lane_off = (int64_t)(cb0 + (piece >> 3)) * h * w * 128 + (piece & 7) * 16;   // (chunk-major rows)
    }
    __device__ __forceinline__ uint4 fetch(int e, int rx, int ry, int ri, bool& inside) const {
        const int iy = iy0 + ry, ix = ix0 + rx, n = img0 + ri;
        inside = e < items && iy >= 0 && iy < h && ix >= 0 && ix < w && n < n_total && chunk_ok;
        // (no branch: a piece outside the tensor is read at the tensor's start and dropped)
        return *reinterpret_cast<const uint4*>(xs + (inside ? (((int64_t)n * chunks * h + iy) * w + ix) * 128 + lane_off : 0));
    }
    __device__ __forceinline__ void load(int img0_, int iy0_, int ix0_) {
        img0 = img0_; iy0 = iy0_; ix0 = ix0_;
        // pixel of this lane's first piece; later pieces lie kThreads / 16 pixels further each
        int rx = (threadIdx.x >> 4) % lw, ry = ((threadIdx.x >> 4) / lw) % lh, ri = (threadIdx.x >> 4) / (lw * lh);
#pragma unroll
        for (int u = 0; u < U; u++) {
            val[u] = fetch(threadIdx.x + u * kThreads, rx, ry, ri, keep[u]);
            rx += kThreads / 16;               // 16 pixels on: at most two rows (lw >= 8) and so at most two images
#pragma unroll
            for (int twice = 0; twice < 2; twice++) {
                const int wrap = rx >= lw;
                rx -= wrap ? lw : 0; ry += wrap;
            }
#pragma unroll
            for (int twice = 0; twice < 2; twice++) {
                const int wrap = ry >= lh;
                ry -= wrap ? lh : 0; ri += wrap;
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* s_dst) const {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = threadIdx.x + u * kThreads;
            const uint4 v = keep[u] ? val[u] : make_uint4(0, 0, 0, 0);
            if (e < items) *reinterpret_cast<uint4*>(s_dst + (e >> 4) * kRowBytes + piece * 16) = v;
        }
        for (int e = threadIdx.x + U * kThreads; e < items; e += kThreads) {
            const int pix = e >> 4;
            bool inside;
            const uint4 v = fetch(e, pix % lw, (pix / lw) % lh, pix / (lw * lh), inside);
            *reinterpret_cast<uint4*>(s_dst + pix * kRowBytes + piece * 16) = inside ? v : make_uint4(0, 0, 0, 0);
        }
    }
};

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
__global__ void __launch_bounds__(kThreads, 1) wgrad_f32_kernel(sg_conv2d_wgrad_params p, WTile t, int tiles_total,
                                                                const unsigned char* __restrict__ xs,
                                                                const unsigned char* __restrict__ gs, const unsigned* amax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_x = smem;
    unsigned char* s_g = s_x + (size_t)t.imgs * t.lh * t.lw * kRowBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * BC, cx0 = blockIdx.z * BC;
    constexpr int MAX_TAPS = NK * NK;
    const int chunks_x = (p.cx + 31) / 32, chunks_g = (p.cg + 31) / 32;

    float mul, inv_x, inv_g;
    half_range_scale(amax, 0, 2, p.x_scale != nullptr, mul, inv_x);
    half_range_scale(amax, 1, 3, p.g_scale != nullptr, mul, inv_g);

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
            g_lane[j] = (dr * t.tw + dc) * kRowBytes + c4 * 8 + (wave >> 1) * 128;   // (a wave's 32 channels: one chunk)
            x_lane[j] = (dr * p.step * t.lw + dc * p.step) * kRowBytes + c4 * 8 + (wave & 1) * 128;
        }
    }
    const int rows_per_step = 32 / t.tw;
    Stager<22> stage_x;   // (352 + 256 pixel rows in registers: the usual tiles whole)
    Stager<16> stage_g;
    stage_x.setup(xs, chunks_x, cx0 / 32, p.n, p.h, p.w_in, t.imgs, t.lh, t.lw);
    stage_g.setup(gs, chunks_g, cg0 / 32, p.n, p.gh, p.gw, t.imgs, t.th, t.tw);

    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        int rest = tile;
        const int tb = rest % t.tiles_b; rest /= t.tiles_b;
        const int ta = rest % t.tiles_a; rest /= t.tiles_a;
        const int img0 = rest * t.imgs;
        const int a0 = ta * t.th, b0 = tb * t.tw;
        __syncthreads();
        stage_x.load(img0, a0 * p.step + p.dy0, b0 * p.step + p.dx0);
        stage_g.load(img0, a0, b0);
        stage_x.store(s_x);
        stage_g.store(s_g);
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
                            dst[n][pl] = fragment(xt + x_lane[0] + n * 32 + pl * 64, xt + x_lane[1] + n * 32 + pl * 64);
                };
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int pl = 0; pl < 2; pl++)
                        a[m][pl] = fragment(gr + g_lane[0] + m * 32 + pl * 64, gr + g_lane[1] + m * 32 + pl * 64);
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
    const int64_t x_pixels = (int64_t)p.h * p.w_in, g_pixels = (int64_t)p.gh * p.gw;
    if (int rc = absmax_measure(stream, (const float*)p.x, (int64_t)p.n * p.cx * x_pixels, (const float*)p.g,
                                (int64_t)p.n * p.cg * g_pixels, p.x_scale, (int64_t)p.n * p.cx, p.g_scale, (int64_t)p.n * p.cg, &amax))
        return rc;
    const size_t x_bytes = split_bytes((int64_t)p.n * x_pixels, p.cx), need = x_bytes + split_bytes((int64_t)p.n * g_pixels, p.cg);
    unsigned char* work = (unsigned char*)p.workspace;
    if (work != nullptr) {
        SG_REQUIRE(p.workspace_bytes >= need && (reinterpret_cast<uintptr_t>(work) & 255u) == 0, SG_ERR_INVALID_ARGUMENT,
                   "conv2d_wgrad: workspace of %zu bytes (need %zu, 256-byte aligned)", p.workspace_bytes, need);
    } else {
        work = (unsigned char*)split_workspace(stream, need);
        if (work == nullptr) return (int)hipErrorOutOfMemory;
    }
    if (int rc = split_pixels(stream, (const float*)p.x, p.x_scale, p.n, p.cx, x_pixels, amax, 0, 2, work)) return rc;
    if (int rc = split_pixels(stream, (const float*)p.g, p.g_scale, p.n, p.cg, g_pixels, amax, 1, 3, work + x_bytes)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)split, (unsigned)blocks_g, (unsigned)blocks_x), dim3(kThreads), lds, stream, p, t,
                       (int)tiles, (const unsigned char*)work, (const unsigned char*)(work + x_bytes), amax);
    return check_launch("sg_conv2d_wgrad (f32)");
}

}  // namespace sg

