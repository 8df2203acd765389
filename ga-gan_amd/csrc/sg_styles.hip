// Styles and demodulation coefficients of one modulated layer, and their gradients, in two launches each.
//
//   styles[n, c] = gain * (a_gain * sum_k w[n, k] a[c, k] + b_gain * b[c])                  (the layer's affine map of its latent row)
//   dcoefs[n, o] = rsqrt(sum_c styles[n, c]^2 wsq_t[c, o] + eps)                            (wsq_t[c, o] = sum over taps of W[o, c, ky, kx]^2)
//
// Everything here is a few million multiply-adds on [batch, channels] tensors: what matters is the number of launches
// (the framework's operators take five for the forward and about fifteen for the backward of every layer, each a
// few microseconds in which the chip does nothing else), not the arithmetic rate.  So: plain vector code, one lane
// per output element, operands that lanes share staged through LDS, 64 to 512 workgroups per launch.
#include "sg_common.h"

#include <algorithm>

namespace sg {
namespace {

constexpr int kLanes = 256;
constexpr int kRows = 8;        // samples per workgroup
constexpr int kTile = 16;       // outputs per sample and workgroup
constexpr int kSlices = 16;     // lanes that share one output's sum
constexpr int kSpan = 32;       // terms per lane and round
constexpr int kRound = kSlices * kSpan;   // 512 terms per round

// All four products below have the shape  out[r][j] = f(sum_i X[r][i] M(i, j))  for kRows samples r at once, with
// sums of a few hundred terms and far too few outputs to fill the chip: what a launch costs is the latency of its
// longest chain of dependent loads.  So a sum is cut into kSlices slices of kSpan terms, a lane issues all the loads
// of its slice before it uses the first (kSpan / 4 16-byte loads where the matrix is contiguous along the sum,
// kSpan 4-byte loads that 16 neighbouring lanes coalesce where it is contiguous along the outputs), X sits in LDS for
// the workgroup, and the slices meet in LDS.  A workgroup: kTile outputs x kRows samples; lane = (output, slice).
enum Product { kStyles, kDemod, kTotal, kLatent };

template <int MODE> struct Shape;      // sum length, outputs; M contiguous along the sum or along the outputs
template <> struct Shape<kStyles> { static constexpr bool along_sum = true;
    static __device__ int terms(const sg_styles_bwd_params& p) { return p.f.k; }  static __device__ int outputs(const sg_styles_bwd_params& p) { return p.f.c; } };
template <> struct Shape<kDemod> { static constexpr bool along_sum = false;
    static __device__ int terms(const sg_styles_bwd_params& p) { return p.f.c; }  static __device__ int outputs(const sg_styles_bwd_params& p) { return p.f.o; } };
template <> struct Shape<kTotal> { static constexpr bool along_sum = true;
    static __device__ int terms(const sg_styles_bwd_params& p) { return p.f.o; }  static __device__ int outputs(const sg_styles_bwd_params& p) { return p.f.c; } };
template <> struct Shape<kLatent> { static constexpr bool along_sum = false;
    static __device__ int terms(const sg_styles_bwd_params& p) { return p.f.c; }  static __device__ int outputs(const sg_styles_bwd_params& p) { return p.f.k; } };

template <int MODE>
__device__ __forceinline__ void product(const sg_styles_bwd_params& p, int tile, int n0) {
    __shared__ __attribute__((aligned(16))) float s_x[kRows][kRound];
    __shared__ float s_part[kSlices][kRows][kTile];
    const sg_styles_params& f = p.f;
    const int tid = threadIdx.x, jl = tid & (kTile - 1), slice = tid >> 4;
    const int terms = Shape<MODE>::terms(p), outputs = Shape<MODE>::outputs(p);
    const int j = tile * kTile + jl, rows = min(kRows, f.n - n0);
    const float* __restrict__ m = (MODE == kStyles || MODE == kLatent) ? f.a : f.wsq_t;
    const int ld = (MODE == kStyles) ? f.k : (MODE == kLatent) ? f.k : f.o;   // row length of M as stored
    float acc[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) acc[r] = 0.f;
    const int ib = slice * kSpan;                                  // this lane's terms of a round: ib .. ib + kSpan
    for (int i0 = 0; i0 < terms; i0 += kRound) {
        // the lane's piece of M first, then its share of X: all of a round's global loads are in flight together
        float mv[kSpan];
        if (Shape<MODE>::along_sum) {      // (terms is a multiple of 4 and rows of M are 16-byte aligned: checked by the host)
            const float* __restrict__ row = m + (int64_t)min(j, outputs - 1) * ld + i0 + ib;
#pragma unroll
            for (int q = 0; q < kSpan / 4; q++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + ib + 4 * q < terms) v = *reinterpret_cast<const float4*>(row + 4 * q);
                mv[4 * q] = v.x; mv[4 * q + 1] = v.y; mv[4 * q + 2] = v.z; mv[4 * q + 3] = v.w;
            }
        } else {
            const int jc = min(j, outputs - 1);
#pragma unroll
            for (int q = 0; q < kSpan; q++) mv[q] = (i0 + ib + q < terms) ? m[(int64_t)(i0 + ib + q) * ld + jc] : 0.f;
        }
        float xv_in[kRows * kRound / kLanes];
#pragma unroll
        for (int it = 0; it < kRows * kRound / kLanes; it++) {     // X of this round (zero beyond the rows / the terms)
            const int r = it / (kRound / kLanes), col = (it % (kRound / kLanes)) * kLanes + tid, i = i0 + col;
            float v = 0.f;
            if (r < rows && i < terms) {
                const int n = n0 + r;
                if (MODE == kStyles) v = f.w[(int64_t)n * f.w_stride + i];
                if (MODE == kDemod) v = f.styles[(int64_t)n * f.c + i];
                if (MODE == kTotal) v = f.dcoefs[(int64_t)n * f.o + i];
                if (MODE == kLatent) v = p.t[(int64_t)n * f.c + i];
            }
            xv_in[it] = v;
        }
        float x2_in[kRows * kRound / kLanes];
        if (MODE == kTotal) {
#pragma unroll
            for (int it = 0; it < kRows * kRound / kLanes; it++) {
                const int r = it / (kRound / kLanes), col = (it % (kRound / kLanes)) * kLanes + tid, i = i0 + col;
                x2_in[it] = (r < rows && i < terms) ? p.ddcoefs[(int64_t)(n0 + r) * f.o + i] : 0.f;
            }
        }
        if (i0) __syncthreads();
#pragma unroll
        for (int it = 0; it < kRows * kRound / kLanes; it++) {
            const int r = it / (kRound / kLanes), col = (it % (kRound / kLanes)) * kLanes + tid;
            float v = xv_in[it];
            if (MODE == kDemod) v = v * v;
            if (MODE == kTotal) v = -x2_in[it] * v * v * v;
            s_x[r][col] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kSpan; q += 4) {
#pragma unroll
            for (int r = 0; r < kRows; r++) {
                const float4 xv = *reinterpret_cast<const float4*>(&s_x[r][ib + q]);
                acc[r] = fmaf(mv[q], xv.x, fmaf(mv[q + 1], xv.y, fmaf(mv[q + 2], xv.z, fmaf(mv[q + 3], xv.w, acc[r]))));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRows; r++) s_part[slice][r][jl] = acc[r];
    __syncthreads();
    if (tid < kRows * kTile) {
        const int r = tid >> 4, jo = tile * kTile + (tid & (kTile - 1));
        if (r < rows && jo < outputs) {
            float dot = 0.f;
#pragma unroll
            for (int sl = 0; sl < kSlices; sl++) dot += s_part[sl][r][tid & (kTile - 1)];
            const int n = n0 + r;
            if (MODE == kStyles) f.styles[(int64_t)n * f.c + jo] = f.gain * (f.a_gain * dot + (f.b ? f.b_gain * f.b[jo] : 0.f));
            if (MODE == kDemod) f.dcoefs[(int64_t)n * f.o + jo] = rsqrtf(dot + f.eps);
            if (MODE == kTotal) {
                const int64_t at = (int64_t)n * f.c + jo;
                p.t[at] = (p.dstyles ? p.dstyles[at] : 0.f) + f.styles[at] * dot;
            }
            if (MODE == kLatent) p.dw[(int64_t)n * f.k + jo] = f.gain * f.a_gain * dot;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(kLanes) styles_product_kernel(sg_styles_bwd_params p) {
    product<MODE>(p, blockIdx.x, blockIdx.y * kRows);
}

// The whole gradient with respect to the styles when no gradient arrives through the coefficients: a copy.
__global__ void __launch_bounds__(kLanes) styles_copy_kernel(sg_styles_bwd_params p) {
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i < (int64_t)p.f.n * p.f.c) p.t[i] = p.dstyles ? p.dstyles[i] : 0.f;
}

// Backward, second launch.  The first `latent_blocks` workgroups: dw[n, k] = gain a_gain sum_c t[n, c] a[c, k] (the
// product above).  Then rows of workgroups, lanes along the contiguous index of what they write, sums over the samples:
//   [0, c / 4)        da[c, k]      = gain a_gain sum_n t[n, c] w[n, k]        and db[c] = gain b_gain sum_n t[n, c]
//   [c / 4, c / 2)    dwsq_t[c, o]  = 1/2 sum_n q[n, o] styles[n, c]^2
// four channels c per row of workgroups.
__global__ void __launch_bounds__(kLanes) styles_grads_kernel(sg_styles_bwd_params p, int latent_tiles, int latent_blocks, int col_blocks) {
    int block = blockIdx.x;
    if (block < latent_blocks) {
        product<kLatent>(p, block % latent_tiles, (block / latent_tiles) * kRows);
        return;
    }
    block -= latent_blocks;
    const int tid = threadIdx.x;
    const int col = (block % col_blocks) * kLanes + tid;
    const int row4 = (block / col_blocks) * 4;          // four channels per workgroup: a loaded w[n, col] serves all four
    const float ga = p.f.gain * p.f.a_gain;
    const int c4 = (p.f.c + 3) & ~3;
    if (row4 < c4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n0 = 0; n0 < p.f.n; n0 += kRows) {
            float wv[kRows];
#pragma unroll
            for (int r = 0; r < kRows; r++)
                wv[r] = (n0 + r < p.f.n && col < p.f.k) ? p.f.w[(int64_t)(n0 + r) * p.f.w_stride + col] : 0.f;
#pragma unroll
            for (int r = 0; r < kRows; r++) {
                if (n0 + r >= p.f.n) continue;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float tv = (row4 + u < p.f.c) ? p.t[(int64_t)(n0 + r) * p.f.c + row4 + u] : 0.f;
                    s[u] = fmaf(tv, wv[r], s[u]);
                    sb[u] += tv;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (row4 + u >= p.f.c) continue;
            if (p.da && col < p.f.k) p.da[(int64_t)(row4 + u) * p.f.k + col] = ga * s[u];
            if (p.db && block % col_blocks == 0 && tid == 0) p.db[row4 + u] = p.f.gain * p.f.b_gain * sb[u];
        }
        return;
    }
    const int row = row4 - c4 + 0;                      // then four channels of dwsq_t per workgroup
    if (!p.dwsq_t || col >= p.f.o) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < p.f.n; n++) {
        const float d = p.f.dcoefs[(int64_t)n * p.f.o + col];
        const float q = -p.ddcoefs[(int64_t)n * p.f.o + col] * d * d * d;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float st = (row + u < p.f.c) ? p.f.styles[(int64_t)n * p.f.c + row + u] : 0.f;
            s[u] = fmaf(q, st * st, s[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (row + u < p.f.c) p.dwsq_t[(int64_t)(row + u) * p.f.o + col] = 0.5f * s[u];
}

int check(const sg_styles_params& p) {
    SG_REQUIRE(p.w && p.a && p.styles, SG_ERR_INVALID_ARGUMENT, "styles: w, a and styles must be given");
    SG_REQUIRE(p.n > 0 && p.k > 0 && p.c > 0 && p.n <= 65535 * kRows && p.w_stride >= p.k, SG_ERR_INVALID_ARGUMENT,
               "styles: n %d, k %d, c %d, row stride %lld", p.n, p.k, p.c, (long long)p.w_stride);
    SG_REQUIRE(p.k % 4 == 0 && aligned16(p.a), SG_ERR_UNSUPPORTED, "styles: k must be a multiple of 4 and `a` 16-byte aligned");
    SG_REQUIRE((p.wsq_t != nullptr) == (p.dcoefs != nullptr) && (!p.wsq_t || p.o > 0), SG_ERR_INVALID_ARGUMENT,
               "styles: wsq_t and dcoefs come together, with o > 0");
    SG_REQUIRE(!p.wsq_t || (p.o % 4 == 0 && aligned16(p.wsq_t)), SG_ERR_UNSUPPORTED,
               "styles: o must be a multiple of 4 and wsq_t 16-byte aligned");
    return SG_OK;
}

unsigned tiles(int outputs) { return (unsigned)((outputs + kTile - 1) / kTile); }
unsigned row_blocks(int n) { return (unsigned)((n + kRows - 1) / kRows); }

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" int sg_styles(const sg_styles_params* p, sg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check(*p)) return rc;
    sg_styles_bwd_params q{};
    q.f = *p;
    hipLaunchKernelGGL(styles_product_kernel<kStyles>, dim3(tiles(p->c), row_blocks(p->n)), dim3(kLanes), 0, stream, q);
    if (int rc = check_launch("sg_styles")) return rc;
    if (p->dcoefs) {
        hipLaunchKernelGGL(styles_product_kernel<kDemod>, dim3(tiles(p->o), row_blocks(p->n)), dim3(kLanes), 0, stream, q);
        return check_launch("sg_styles (demodulation)");
    }
    return SG_OK;
}

extern "C" int sg_styles_backward(const sg_styles_bwd_params* p, sg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const sg_styles_params& f = p->f;
    if (int rc = check(f)) return rc;
    SG_REQUIRE(p->t != nullptr, SG_ERR_INVALID_ARGUMENT, "styles backward: the workspace t [n, c] must be given");
    SG_REQUIRE(!p->ddcoefs || f.dcoefs, SG_ERR_INVALID_ARGUMENT, "styles backward: ddcoefs without dcoefs");
    SG_REQUIRE(!p->dwsq_t || p->ddcoefs, SG_ERR_INVALID_ARGUMENT, "styles backward: dwsq_t needs ddcoefs");
    if (p->ddcoefs) {
        hipLaunchKernelGGL(styles_product_kernel<kTotal>, dim3(tiles(f.c), row_blocks(f.n)), dim3(kLanes), 0, stream, *p);
    } else {
        const int64_t count = (int64_t)f.n * f.c;
        hipLaunchKernelGGL(styles_copy_kernel, dim3((unsigned)((count + kLanes - 1) / kLanes)), dim3(kLanes), 0, stream, *p);
    }
    if (int rc = check_launch("sg_styles_backward (styles)")) return rc;
    if (!p->dw && !p->da && !p->db && !p->dwsq_t) return SG_OK;
    const int latent_tiles = p->dw ? (int)tiles(f.k) : 0, latent_blocks = latent_tiles * (int)row_blocks(f.n);
    const int cols = std::max(f.k, p->dwsq_t ? f.o : 0), col_blocks = (cols + kLanes - 1) / kLanes;
    const int quads = (f.c + 3) / 4, rows = (p->da || p->db || p->dwsq_t) ? quads + (p->dwsq_t ? quads : 0) : 0;
    const int64_t blocks = (int64_t)latent_blocks + (int64_t)rows * col_blocks;
    SG_REQUIRE(blocks <= INT32_MAX, SG_ERR_UNSUPPORTED, "styles backward: %lld workgroups", (long long)blocks);
    hipLaunchKernelGGL(styles_grads_kernel, dim3((unsigned)blocks), dim3(kLanes), 0, stream, *p, latent_tiles, latent_blocks, col_blocks);
    return check_launch("sg_styles_backward");
}
