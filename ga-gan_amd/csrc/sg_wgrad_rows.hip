// Weight gradient of the 3x3, unit-stride, same-size convolution of 16-bit tensors with 32 channels on both sides
// (the highest-resolution layers of a generator: few channels, millions of pixels, bound by memory):
//   dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * m * g_scale[n,cg] * x[n,cx,a+ky-1,b+kx-1] * x_scale[n,cx]
// with m the optional activation mask of sg_conv2d_wgrad_params.mask_y.  Same contract as the other kernels behind
// sg_conv2d_wgrad.
//
// GEMM view per tap: M = cg, N = cx, K = pixels - and in NCHW memory the pixels of a channel ARE contiguous, so the
// operands need no transposition anywhere: a tile (4 rows x 64 columns of one sample; for x with a halo of one row
// and one 8-pixel packet on either side) goes to LDS as [channel][row][column] by plain 16-byte copies, and a
// fragment of the matrix instruction (one channel, 8 consecutive pixels per lane) is one aligned ds_read_b128 of a
// channel's row.  The horizontal taps shift the 8 pixels by one: the lane reads the dword before and after its
// packet as well and funnel-shifts (5 v_alignbit_b32 give both shifted fragments).  Mask, scales and rounding are
// applied to the 16-byte packets on their way from global memory to LDS.
//
// A workgroup (4 waves) accumulates dw for all 9 taps in registers over the tiles it walks and adds it to dw with
// float atomics at the end.  A wave owns one 16 x 16 tile of dw (36 accumulators) and two workgroups share a CU (48 KB
// of LDS each), so that one stages while the other multiplies.  (With 64 channels on a side the workgroup would
// need the CU to itself and the transposing kernel of sg_wgrad3x3.hip is faster: those layers stay there.)
#include <algorithm>

#include "sg_common.h"
#include "sg_conv3x3.h"

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TW = 64, TH = 4;                 // gradient pixels per tile
constexpr int XW = TW + 16, XH = TH + 2;       // staged input: image columns b0 - 8 .. b0 + TW + 7, rows a0 - 1 .. a0 + TH
constexpr int kGPitch = TH * TW * 2 + 16;      // bytes per channel: 16 lanes (channels) of a read fall on 16 different
constexpr int kXPitch = XH * XW * 2 + 16;      //   16-byte bank slots (528 = 2 x 256 + 16, 976 = 3 x 256 + 13 x 16)

struct RowsArgs {
    const void* x; const void* g; const void* mask_y;
    float* dw;
    const float* x_scale; const float* g_scale;
    int n, cx, cg, h, w;
    int tiles_a, tiles_b, tiles_total;
    float mask_alpha, mask_clamp;
};

template <class T> struct MmaRows;
template <> struct MmaRows<__hip_bfloat16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct MmaRows<__half> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// 8 values of a packet times `s` (and, with a mask packet, times 1 / alpha / 0 by the saved output), rounded back.
template <class T, bool MASK>
__device__ __forceinline__ uint4 scale_packet(const uint4& v, const uint4& y, float s, float alpha, float bound) {
    const Pack<T> pv = __builtin_bit_cast(Pack<T>, v), py = __builtin_bit_cast(Pack<T>, y);
    Pack<T> out;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float f = (float)Elem<T>::load(pv.v[j]) * s;
        if constexpr (MASK) {
            const float yv = (float)Elem<T>::load(py.v[j]);
            f *= (yv > 0.f) ? 1.f : alpha;
            if (bound >= 0.f && !(yv > -bound && yv < bound)) f = 0.f;
        }
        out.v[j] = Elem<T>::store(f);
    }
    return __builtin_bit_cast(uint4, out);
}

template <class T, bool MASK>
__global__ void __launch_bounds__(kThreads, 2) wgrad_rows_kernel(RowsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_g = smem;
    unsigned char* s_x = smem + (size_t)p.cg * kGPitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ my = (const T*)p.mask_y;

    // A wave accumulates one 16 x 16 tile of the 32 x 32 block dw over every 32-pixel step: 36 accumulator registers,
    // which leaves room for a second workgroup on the CU.
    constexpr int MT = 1;                       // tiles per wave and side
    constexpr int kpart = 0, kwaves = 1;
    const int m_first = (wave >> 1) * 16, n_first = (wave & 1) * 16;   // first channel of the wave's rows / columns of dw

    f32x4_t acc[9][MT][MT];
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int n = 0; n < MT; n++) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int a_base[MT], b_base[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        a_base[m] = (m_first + m * 16 + lrow) * kGPitch + lgrp * 16;
        b_base[m] = (n_first + m * 16 + lrow) * kXPitch + (8 + lgrp * 8) * 2;   // (LDS column 8 = image column b0)
    }

    const int64_t plane = (int64_t)p.h * p.w;
    const int g_items = p.cg * TH * (TW / 8), x_items = p.cx * XH * (XW / 8);
    constexpr int CH = 32;
    constexpr int G_IT = CH * TH * (TW / 8) / kThreads, X_IT = (CH * XH * (XW / 8) + kThreads - 1) / kThreads;   // 4, 8

    // The packets of a tile are requested into registers one tile ahead (behind the multiplies of the tile before)
    // and go to LDS - masked, scaled, rounded - when their turn comes.
    uint4 gv[G_IT], yv[MASK ? G_IT : 1], xv[X_IT];
    float gs[G_IT], xs[X_IT];
    auto issue = [&](int tile) {
        int rest = tile;
        const int tb = rest % p.tiles_b; rest /= p.tiles_b;
        const int ta = rest % p.tiles_a;
        const int n = rest / p.tiles_a;
        const int a0 = ta * TH, b0 = tb * TW;
#pragma unroll
        for (int k = 0; k < G_IT; k++) {
            const int e = tid + k * kThreads;
            const int pk = e & 7, row = (e >> 3) & 3, c = e >> 5;
            gs[k] = 1.f;
            if (e < g_items) {
                const int64_t off = ((int64_t)n * p.cg + c) * plane + (int64_t)(a0 + row) * p.w + b0 + pk * 8;
                gv[k] = *reinterpret_cast<const uint4*>(g + off);
                if constexpr (MASK) yv[k] = *reinterpret_cast<const uint4*>(my + off);
                if (p.g_scale) gs[k] = p.g_scale[(int64_t)n * p.cg + c];
            }
        }
#pragma unroll
        for (int k = 0; k < X_IT; k++) {
            const int e = tid + k * kThreads;
            const int pk = e % (XW / 8);
            const int q = e / (XW / 8);
            const int row = q % XH, c = q / XH;
            const int iy = a0 - 1 + row, ix = b0 - 8 + pk * 8;
            xv[k] = make_uint4(0, 0, 0, 0);
            xs[k] = 1.f;
            if (e < x_items && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {   // (else: outside the image, zero)
                xv[k] = *reinterpret_cast<const uint4*>(x + ((int64_t)n * p.cx + c) * plane + (int64_t)iy * p.w + ix);
                if (p.x_scale) xs[k] = p.x_scale[(int64_t)n * p.cx + c];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < G_IT; k++) {
            const int e = tid + k * kThreads;
            if (e >= g_items) continue;
            const int pk = e & 7, row = (e >> 3) & 3, c = e >> 5;
            uint4 v = gv[k];
            if (MASK || p.g_scale) v = scale_packet<T, MASK>(v, yv[MASK ? k : 0], gs[k], p.mask_alpha, p.mask_clamp);
            *reinterpret_cast<uint4*>(s_g + c * kGPitch + (row * TW + pk * 8) * 2) = v;
        }
#pragma unroll
        for (int k = 0; k < X_IT; k++) {
            const int e = tid + k * kThreads;
            if (e >= x_items) continue;
            const int pk = e % (XW / 8);
            const int q = e / (XW / 8);
            const int row = q % XH, c = q / XH;
            uint4 v = xv[k];
            if (p.x_scale) v = scale_packet<T, false>(v, v, xs[k], 1.f, -1.f);
            *reinterpret_cast<uint4*>(s_x + c * kXPitch + (row * XW + pk * 8) * 2) = v;
        }
    };

    // Workgroups are dealt to the 8 XCDs in turn, and each XCD has an L2 of its own: XCD k takes the k-th eighth of
    // the tiles, its workgroups striding through it side by side, so that the halo a tile shares with its neighbours
    // (whole 128-byte lines for one column of pixels) is fetched from HBM once.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int share = (p.tiles_total + 7) / 8;
    const int t_begin = xcd * share, t_end = min(p.tiles_total, t_begin + share);
    if (t_begin + slot < t_end) issue(t_begin + slot);
    for (int tile = t_begin + slot; tile < t_end; tile += slots) {
        __syncthreads();   // every wave has read the previous tile
        commit();
        __syncthreads();
        if (tile + slots < t_end) issue(tile + slots);

        // ---- multiply: 32-pixel steps (row, half of the row) ------------------------------------------------------
        for (int ks = kpart; ks < TH * (TW / 32); ks += kwaves) {
            const int row = ks >> 1, col0 = (ks & 1) * 32;
            uint4 a[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) a[m] = *reinterpret_cast<const uint4*>(s_g + a_base[m] + (row * TW + col0) * 2);
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
#pragma unroll
                for (int n2 = 0; n2 < MT; n2++) {
                    const unsigned char* src = s_x + b_base[n2] + ((row + ky) * XW + col0) * 2;
                    const uint4 d = *reinterpret_cast<const uint4*>(src);
                    const unsigned before = *reinterpret_cast<const unsigned*>(src - 4);
                    const unsigned after = *reinterpret_cast<const unsigned*>(src + 16);
                    const unsigned f0 = __builtin_amdgcn_alignbit(d.x, before, 16), f1 = __builtin_amdgcn_alignbit(d.y, d.x, 16);
                    const unsigned f2 = __builtin_amdgcn_alignbit(d.z, d.y, 16), f3 = __builtin_amdgcn_alignbit(d.w, d.z, 16);
                    const unsigned f4 = __builtin_amdgcn_alignbit(after, d.w, 16);
                    const uint4 left = make_uint4(f0, f1, f2, f3), right = make_uint4(f1, f2, f3, f4);
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        MmaRows<T>::run(a[m], left, acc[ky * 3 + 0][m][n2]);
                        MmaRows<T>::run(a[m], d, acc[ky * 3 + 1][m][n2]);
                        MmaRows<T>::run(a[m], right, acc[ky * 3 + 2][m][n2]);
                    }
                }
            }
        }
    }

    // ---- add into dw[tap][cg][cx]: lanes of a row run along cx ------------------------------------------------
    float* __restrict__ dw = p.dw;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int n = 0; n < MT; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int cg = m_first + m * 16 + lgrp * 4 + r;
                    const int cx = n_first + n * 16 + lrow;
                    atomicAdd(dw + ((int64_t)tp * p.cg + cg) * p.cx + cx, acc[tp][m][n][r]);
                }
}

template <class T>
int launch_rows(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    RowsArgs a;
    a.x = p.x; a.g = p.g; a.mask_y = p.mask_y; a.dw = p.dw; a.x_scale = p.x_scale; a.g_scale = p.g_scale;
    a.n = p.n; a.cx = p.cx; a.cg = p.cg; a.h = p.h; a.w = p.w_in;
    a.tiles_a = p.h / TH; a.tiles_b = p.w_in / TW;
    const int64_t tiles = (int64_t)a.tiles_a * a.tiles_b * p.n;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: too many tiles");
    a.tiles_total = (int)tiles;
    a.mask_alpha = p.mask_alpha; a.mask_clamp = p.mask_clamp;
    const size_t lds = (size_t)p.cg * kGPitch + (size_t)p.cx * kXPitch;
    const int per_cu = 2;
    const unsigned grid = (unsigned)std::max<int64_t>(8, std::min<int64_t>((tiles + 7) / 8 * 8, (int64_t)compute_units() * per_cu) / 8 * 8);   // a multiple of 8
    auto kernel = p.mask_y ? wgrad_rows_kernel<T, true> : wgrad_rows_kernel<T, false>;
    if (lds > 64 * 1024) {
        hipError_t e = allow_large_lds((const void*)kernel);
        if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds, stream, a);
    return check_launch("sg_conv2d_wgrad (rows)");
}

}  // namespace

bool wgrad_rows_eligible(const sg_conv2d_wgrad_params& p) {
    if (p.dtype != SG_BF16 && p.dtype != SG_F16) return false;
    if (p.nky != 3 || p.nkx != 3 || p.step != 1 || p.dy0 != -1 || p.dx0 != -1 || p.gh != p.h || p.gw != p.w_in) return false;
    if (p.cg != 32 || p.cx != 32) return false;
    if (p.w_in % TW != 0 || p.h % TH != 0 || p.n < 1) return false;
    if ((int64_t)p.n * std::max(p.cg, p.cx) * p.h * p.w_in > INT32_MAX) return false;
    return aligned16(p.x) && aligned16(p.g) && (p.mask_y == nullptr || aligned16(p.mask_y));
}

int wgrad_rows_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    return p.dtype == SG_BF16 ? launch_rows<__hip_bfloat16>(p, stream) : launch_rows<__half>(p, stream);
}

}  // namespace sg
