// 1x1 convolution of 16-bit tensors with 32 or more channels on both sides: the residual skips of the discriminator
// (a 1x1 contraction behind the down-sampling filter) and their data gradients.  Per image this is the product
// Y[co][pixel] = W[co][ci] X[ci][pixel] with both planes row-major in memory, 20 .. 170 multiply-adds per byte moved:
// bound by HBM, so the kernel is built around moving X once in and Y once out in whole, coalesced packets.
//
// A workgroup (4 waves) owns 256 consecutive pixels of one image's flattened planes and up to 128 output channels;
// each wave 64 of the pixels.  GEMM view: M = pixels, N = output channels, K = input channels, 32 at a time:
//   * the X chunk [32 channels][256 pixels] goes to LDS exactly as it lies in memory (16-byte copies, one channel row
//     per 32 lanes) - no transposition in registers.  The matrix instruction wants 8 consecutive K values of one
//     pixel per lane; ds_read_b64_tr_b16 delivers 4 channel rows x 16 pixels transposed, so a fragment is two such
//     reads.  Rows are 544 bytes (32 modulo 256) and the K slots of lane group g are channels 4g .. 4g+3 and
//     16+4g .. 16+4g+3, so that the 8 rows a 32-lane half reads are consecutive: no bank conflicts;
//   * the W chunk [channels out][32 channels in] is staged in that K order (each 16-byte piece of a weight row lands
//     as two 8-byte halves), 64-byte rows with the 16-byte slots XOR-swizzled;
//   * the next chunk's packets are requested before the multiplies of the current one and written to the other LDS
//     buffer after them: one barrier per chunk.
// A lane ends up with 4 consecutive pixels of one output channel per accumulator and stores them as 8 bytes after
// bias, leaky ReLU, gain and clamp.  Workgroups that share a pixel tile (more than 128 output channels) follow each
// other on one XCD, so the tile's second reading comes from that XCD's L2.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TP = 256;                 // pixels per workgroup
constexpr int KCH = 32;                 // input channels per chunk
constexpr int kXPitch = TP * 2 + 32;    // bytes per channel row of the staged X chunk
constexpr int kXBytes = KCH * kXPitch;  // 17408
constexpr int kWRow = 64;               // bytes per staged weight row

struct PwArgs {
    const void* x; const void* w; void* y; const void* bias;
    const void* addend;   // added to the result after the epilogue, or null
    int n, ci, co;
    int64_t hw;           // pixels per plane, a multiple of 8
    int tiles_img;        // pixel tiles per image
    int64_t tiles;        // pixel tiles in all
    int cblocks;          // output-channel blocks
    float slope, gain, hi;
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

__device__ __forceinline__ int wswz(int row, int slot) { return row * kWRow + (((slot ^ (row >> 1)) & 3) << 4); }

__device__ __forceinline__ uint2 read_tr(const unsigned char* p) {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p)));
}

template <class T> __device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(a)) |
           ((unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(b)) << 16);
}

template <class T> __device__ __forceinline__ float unpack_lo(unsigned v) {
    return Elem<T>::load(__builtin_bit_cast(T, (unsigned short)(v & 0xffffu)));
}
template <class T> __device__ __forceinline__ float unpack_hi(unsigned v) {
    return Elem<T>::load(__builtin_bit_cast(T, (unsigned short)(v >> 16)));
}

template <class T, int NTL, bool ADD>   // NTL: 16-channel output tiles per workgroup; ADD: an addend comes with the output
__global__ void __launch_bounds__(kThreads, NTL == 8 ? 2 : 3) pointwise_kernel(PwArgs a) {
    constexpr int BN = 16 * NTL;
    constexpr int W_IT = (BN * 4 + kThreads - 1) / kThreads;   // 16-byte weight pieces per lane and chunk
    constexpr int kWBytes = BN * kWRow;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;

    // workgroup -> (pixel tile, channel block): the blocks of a tile are neighbours on one XCD
    const int64_t u = blockIdx.x;
    const int xcd = (int)(u & 7);
    const int64_t v = u >> 3;
    const int cb = (int)(v % a.cblocks);
    const int64_t tile = (v / a.cblocks) * 8 + xcd;
    if (tile >= a.tiles) return;
    const int img = (int)(tile / a.tiles_img);
    const int64_t p0 = (tile % a.tiles_img) * TP;
    const int co0 = cb * BN;
    const int chunks = a.ci / KCH;
    const int nbuf = chunks > 1 ? 2 : 1;

    // (the buffers are named by offsets into smem: pointers picked from an array at run time lose their address space,
    // and every access through them becomes a flat one that waits for the global loads in flight as well)
    const int x_second = (nbuf - 1) * kXBytes, w_first = nbuf * kXBytes, w_second = (nbuf - 1) * kWBytes;
    auto s_x = [&](int buf) __attribute__((always_inline)) { return smem + buf * x_second; };
    auto s_w = [&](int buf) __attribute__((always_inline)) { return smem + w_first + buf * w_second; };

    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ w = (const T*)a.w;

    // ---- staging plan -----------------------------------------------------------------------------------------
    const int xrow = tid >> 5, xpk = tid & 31;                      // channel row (of 8 per pass) and 8-pixel packet
    const bool x_ok = p0 + xpk * 8 < a.hw;
    const T* x_src = x + ((int64_t)img * a.ci + xrow) * a.hw + p0 + xpk * 8;
    uint4 xreg[4], wreg[W_IT];
    auto load = [&](int c) __attribute__((always_inline)) {
        const int c0 = c * KCH;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            xreg[it] = make_uint4(0, 0, 0, 0);
            if (x_ok) xreg[it] = *reinterpret_cast<const uint4*>(x_src + (int64_t)(c0 + it * 8) * a.hw);
        }
#pragma unroll
        for (int k = 0; k < W_IT; k++) {
            const int e = tid + k * kThreads, row = e >> 2, q = e & 3;
            wreg[k] = make_uint4(0, 0, 0, 0);
            if (e < BN * 4 && co0 + row < a.co) wreg[k] = *reinterpret_cast<const uint4*>(w + (int64_t)(co0 + row) * a.ci + c0 + q * 8);
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; it++)
            *reinterpret_cast<uint4*>(s_x(buf) + (it * 8 + xrow) * kXPitch + xpk * 16) = xreg[it];
#pragma unroll
        for (int k = 0; k < W_IT; k++) {
            const int e = tid + k * kThreads, row = e >> 2, q = e & 3;
            if (e < BN * 4) {
                // piece q holds channels 8q .. 8q+7: groups 2q and 2q+1 of four; group h is K slot (h & 3, h >> 2)
                const int h0 = 2 * q, h1 = 2 * q + 1;
                *reinterpret_cast<uint2*>(s_w(buf) + wswz(row, h0 & 3) + (h0 >> 2) * 8) = make_uint2(wreg[k].x, wreg[k].y);
                *reinterpret_cast<uint2*>(s_w(buf) + wswz(row, h1 & 3) + (h1 >> 2) * 8) = make_uint2(wreg[k].z, wreg[k].w);
            }
        }
    };

    f32x4_t acc[4][NTL];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < NTL; j++) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // this lane's addresses: transposing reads (lane 4q+p of a 16-lane group supplies row q, pixels 4p .. 4p+3)
    const int a_off = (4 * lgrp + (lrow >> 2)) * kXPitch + (wave * 64 + (lrow & 3) * 4) * 2;
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        uint4 af[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint2 lo = read_tr(s_x(buf) + a_off + t * 32);
            const uint2 hi = read_tr(s_x(buf) + a_off + t * 32 + 16 * kXPitch);
            af[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int j = 0; j < NTL; j++) {
            const uint4 bf = *reinterpret_cast<const uint4*>(s_w(buf) + wswz(j * 16 + lrow, lgrp));
#pragma unroll
            for (int t = 0; t < 4; t++) Mma16<T>::run(af[t], bf, acc[t][j]);
        }
    };

    load(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < chunks; c++) {
        if (c + 1 < chunks) load(c + 1);
        multiply(c & 1);
        if (c + 1 < chunks) commit((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: rows 4 lgrp .. 4 lgrp + 3 of tile t are 4 consecutive pixels, the column is the channel ------------
    // All of a lane's bias values are requested before the first is used, and the addend packets of channel tile j + 1
    // while tile j is finished (unconditional loads: a packet outside the tensor is read from zeros).  Loaded where they
    // are used - the bias inside `a.bias ? ... : 0`, an addend packet inside `if (add)` - each was waited for on the
    // spot: up to 8 + 32 memory round trips in a row at the end of a workgroup whose K loop is a few hundred cycles long.
    T* __restrict__ y = (T*)a.y;
    const T* __restrict__ add = (const T*)a.addend;
    const int64_t px0 = p0 + wave * 64 + lgrp * 4;
    unsigned short bsr[NTL];
#pragma unroll
    for (int j = 0; j < NTL; j++) {
        const int ch = co0 + j * 16 + lrow;
        bsr[j] = load2_or_zero(a.bias != nullptr && ch < a.co, (const T*)a.bias + ch);
    }
    uint2 av[2][4];
    auto load_addend = [&](int j, uint2 (&v)[4]) __attribute__((always_inline)) {
        if constexpr (ADD) {
            const int ch = co0 + j * 16 + lrow;
            const T* src = add + ((int64_t)img * a.co + ch) * a.hw;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t px = px0 + t * 16;
                v[t] = load8_or_zero(ch < a.co && px < a.hw, src + px);
            }
        }
    };
    load_addend(0, av[0]);
#pragma unroll
    for (int j = 0; j < NTL; j++) {
        if (j + 1 < NTL) load_addend(j + 1, av[(j + 1) & 1]);
        const int ch = co0 + j * 16 + lrow;
        const float bs = unpack_lo<T>(bsr[j]);
        T* dst = y + ((int64_t)img * a.co + ch) * a.hw;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int64_t px = px0 + t * 16;
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float z = acc[t][j][e] + bs;
                z = (z > 0.f) ? z : z * a.slope;
                r[e] = fminf(fmaxf(z * a.gain, -a.hi), a.hi);
            }
            if constexpr (ADD) {     // the other branch of a residual block: both summands are rounded values of the tensor's type, the sum is rounded once
                const uint2 v = av[j & 1][t];
                r[0] = Elem<T>::load(Elem<T>::store(r[0])) + unpack_lo<T>(v.x);
                r[1] = Elem<T>::load(Elem<T>::store(r[1])) + unpack_hi<T>(v.x);
                r[2] = Elem<T>::load(Elem<T>::store(r[2])) + unpack_lo<T>(v.y);
                r[3] = Elem<T>::load(Elem<T>::store(r[3])) + unpack_hi<T>(v.y);
            }
            if (ch < a.co && px < a.hw) *reinterpret_cast<uint2*>(dst + px) = make_uint2(pack2<T>(r[0], r[1]), pack2<T>(r[2], r[3]));
        }
    }
}

template <class T, int NTL>
int launch_tiles(const PwArgs& a, hipStream_t stream) {
    auto kernel = a.addend ? pointwise_kernel<T, NTL, true> : pointwise_kernel<T, NTL, false>;
    const int chunks = a.ci / KCH, nbuf = chunks > 1 ? 2 : 1;
    const size_t lds = (size_t)nbuf * (kXBytes + 16 * NTL * kWRow);
    const int64_t groups = (a.tiles + 7) / 8 * 8 * a.cblocks;
    SG_REQUIRE(groups <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(kThreads), lds, stream, a);
    return check_launch("sg_conv2d (pointwise)");
}

template <class T>
int launch_pointwise(const sg_conv2d_params& p, hipStream_t stream) {
    PwArgs a;
    a.x = p.x; a.w = p.w; a.y = p.y; a.bias = p.bias; a.addend = p.addend;
    a.n = p.n; a.ci = p.ci; a.co = p.co;
    a.hw = (int64_t)p.h * p.w_in;
    a.tiles_img = (int)((a.hw + TP - 1) / TP);
    a.tiles = (int64_t)a.tiles_img * p.n;
    a.slope = (p.act == 3) ? p.alpha : 1.f;
    a.gain = p.gain;
    a.hi = (p.clamp >= 0.f) ? p.clamp : __builtin_huge_valf();
    if (p.co <= 32) { a.cblocks = 1; return launch_tiles<T, 2>(a, stream); }
    if (p.co <= 64) { a.cblocks = 1; return launch_tiles<T, 4>(a, stream); }
    a.cblocks = (p.co + 127) / 128;
    return launch_tiles<T, 8>(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of those layers: dw[cg][cx] += sum over samples and pixels of g[n, cg, p] x[n, cx, p].  Both operands
// lie in memory as [channel][pixel], which is the K-contiguous row order the matrix instruction wants of both of its
// operands: a chunk of 128 pixels goes to LDS as it is (16-byte copies, rows of 272 bytes: the 16 rows a ds_read_b128
// lane group touches fall on 16 different bank slots) and every fragment is one 16-byte read - no transposition at
// all.  A workgroup owns a block of 128 x 64 weights (the general kernel's 32 x 32 blocks read g twice and x four
// times for 64 -> 128 channels: 2.7 times the bytes of the tensors; here each is read once per block column / row),
// its four waves 64 x 32 each; the pixels are divided over blockIdx.x and the partial blocks meet in dw by float
// atomics.  The next chunk's packets are requested before the multiplies of the current one.
constexpr int kWgPx = 128;                       // pixels per chunk
constexpr int kWgBG = 128, kWgBX = 64;           // block of dw: gradient channels x input channels
constexpr int kWgRowBytes = kWgPx * 2 + 16;      // 272
constexpr int kWgIt = (kWgBG + kWgBX) * (kWgPx / 8) / kThreads;   // 16-byte packets per lane and chunk: 12

struct Wg1x1Args {
    const void* g; const void* x; float* dw;
    int n, cg, cx;
    int64_t hw;
    int chunks_img;      // pixel chunks per image
    int64_t chunks;      // in all
    int per_wg;          // chunks per workgroup (blockIdx.x)
};

template <class T>
__global__ void __launch_bounds__(kThreads, 2) wgrad1x1_kernel(Wg1x1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [128 + 64 rows][272 bytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int g0 = blockIdx.y * kWgBG, x0 = blockIdx.z * kWgBX;
    const T* __restrict__ g = (const T*)a.g;
    const T* __restrict__ x = (const T*)a.x;

    // staging: packet e = tid + 256 k is 16-byte piece e & 15 of row e >> 4 (rows 0 .. 127: g, 128 .. 191: x)
    uint4 v[kWgIt];
    auto load = [&](int64_t chunk) __attribute__((always_inline)) {
        const int img = (int)(chunk / a.chunks_img);
        const int64_t p0 = (chunk - (int64_t)img * a.chunks_img) * kWgPx;
#pragma unroll
        for (int k = 0; k < kWgIt; k++) {
            const int e = tid + k * kThreads, row = e >> 4, pk = e & 15;
            const bool is_g = row < kWgBG;
            const int ch = is_g ? g0 + row : x0 + row - kWgBG;
            const bool ok = p0 + pk * 8 < a.hw && (is_g ? ch < a.cg : ch < a.cx);
            const T* src = is_g ? g + ((int64_t)img * a.cg + ch) * a.hw : x + ((int64_t)img * a.cx + ch) * a.hw;
            v[k] = load16_or_zero(ok, src + p0 + pk * 8);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < kWgIt; k++) {
            const int e = tid + k * kThreads;
            *reinterpret_cast<uint4*>(smem + (e >> 4) * kWgRowBytes + (e & 15) * 16) = v[k];
        }
    };

    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* a_src = smem + ((wave >> 1) * 64 + lrow) * kWgRowBytes + lgrp * 16;
    const unsigned char* b_src = smem + (kWgBG + (wave & 1) * 32 + lrow) * kWgRowBytes + lgrp * 16;

    const int64_t c0 = (int64_t)blockIdx.x * a.per_wg, c1 = min(a.chunks, c0 + a.per_wg);
    if (c0 >= c1) return;
    load(c0);
    for (int64_t c = c0; c < c1; c++) {
        __syncthreads();          // the previous chunk's reads are done
        commit();
        __syncthreads();
        if (c + 1 < c1) load(c + 1);
#pragma unroll
        for (int ks = 0; ks < kWgPx / 32; ks++) {
            uint4 fa[4], fb[2];
#pragma unroll
            for (int m = 0; m < 4; m++) fa[m] = *reinterpret_cast<const uint4*>(a_src + m * 16 * kWgRowBytes + ks * 64);
#pragma unroll
            for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const uint4*>(b_src + j * 16 * kWgRowBytes + ks * 64);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int j = 0; j < 2; j++) Mma16<T>::run(fa[m], fb[j], acc[m][j]);     // rows: g channels, columns: x channels
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int cg = g0 + (wave >> 1) * 64 + m * 16 + lgrp * 4 + r, cx = x0 + (wave & 1) * 32 + j * 16 + lrow;
                if (cg < a.cg && cx < a.cx) atomicAdd(a.dw + (int64_t)cg * a.cx + cx, acc[m][j][r]);
            }
}

}  // namespace

bool pointwise_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 1 || p.in_step != 1 || p.out_step != 1) return false;
    const sg_conv2d_phase& ph = p.phase[0];
    if (ph.nky != 1 || ph.nkx != 1 || ph.oy0 != 0 || ph.ox0 != 0 || ph.dy0 != 0 || ph.dx0 != 0 || ph.w_offset != 0) return false;
    if (p.oh != p.h || p.ow != p.w_in) return false;
    if (p.in_scale || p.out_scale || p.noise || p.reduce_x || p.mask_y) return false;
    if (p.ci < 32 || p.ci % KCH != 0 || p.co < 16) return false;
    const int64_t hw = (int64_t)p.h * p.w_in;
    if (hw % 8 != 0 || hw < 1024) return false;
    return aligned16(p.x) && aligned16(p.y) && aligned16(p.w) && aligned16(p.addend);
}

int pointwise_launch(const sg_conv2d_params& p, hipStream_t stream) {
    return (p.dtype == SG_BF16) ? launch_pointwise<__hip_bfloat16>(p, stream) : launch_pointwise<__half>(p, stream);
}

bool wgrad1x1_eligible(const sg_conv2d_wgrad_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.nky != 1 || p.nkx != 1 || p.step != 1 || p.dy0 != 0 || p.dx0 != 0 || p.gh != p.h || p.gw != p.w_in) return false;
    if (p.x_scale || p.g_scale || p.mask_y) return false;
    if (p.cx < 32 || p.cg < 32) return false;
    const int64_t hw = (int64_t)p.h * p.w_in;
    if (hw % 8 != 0 || hw * p.n < 16 * kWgPx) return false;
    return aligned16(p.x) && aligned16(p.g);
}

int wgrad1x1_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    Wg1x1Args a;
    a.g = p.g; a.x = p.x; a.dw = p.dw;
    a.n = p.n; a.cg = p.cg; a.cx = p.cx;
    a.hw = (int64_t)p.h * p.w_in;
    a.chunks_img = (int)((a.hw + kWgPx - 1) / kWgPx);
    a.chunks = (int64_t)a.chunks_img * p.n;
    const int blocks = ((p.cg + kWgBG - 1) / kWgBG) * ((p.cx + kWgBX - 1) / kWgBX);
    // about four workgroups per CU, and at least eight chunks each: every workgroup ends in 8,192 atomic adds
    int64_t split = std::max<int64_t>(1, std::min<int64_t>((1024 + blocks - 1) / blocks, a.chunks / 8));
    a.per_wg = (int)((a.chunks + split - 1) / split);
    split = (a.chunks + a.per_wg - 1) / a.per_wg;
    const dim3 grid((unsigned)split, (unsigned)((p.cg + kWgBG - 1) / kWgBG), (unsigned)((p.cx + kWgBX - 1) / kWgBX));
    const size_t lds = (size_t)(kWgBG + kWgBX) * kWgRowBytes;
    if (p.dtype == SG_BF16) hipLaunchKernelGGL((wgrad1x1_kernel<__hip_bfloat16>), grid, dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((wgrad1x1_kernel<__half>), grid, dim3(kThreads), lds, stream, a);
    return check_launch("sg_conv2d_wgrad (1x1)");
}

}  // namespace sg
