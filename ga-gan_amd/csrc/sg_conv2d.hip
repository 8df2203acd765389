// Implicit-GEMM 2-D convolution on the matrix cores, with the StyleGAN2 layer fused
// around it:   y = clamp(act((conv(x * in_scale[n, ci], W) * out_scale[n, co]
//                              + noise * noise_gain + bias[co])) * gain)
// One launch covers the modulated 3x3 convolution of a synthesis layer (in_scale =
// styles, out_scale = demodulation coefficients), a plain discriminator layer (no
// scales), the 1x1 to-RGB / from-RGB contractions, strided and transposed (polyphase)
// convolutions, and - with transposed weights and the two scales exchanged - the data
// gradient of any of them.
//
// GEMM view per image: M = output channels, N = output pixels, K = input channels x
// taps.  A workgroup (4 waves) owns BM output channels x 256 output pixels (IMGS x TH
// x TW).  The K loop walks the input channels in chunks of 64 bytes per pixel (32
// bf16/f16 or 16 f32 channels):
//   * the input footprint of the pixel tile (with its filter halo) is read from the
//     NCHW tensor with row-contiguous loads, multiplied by in_scale and written to LDS
//     as [pixel][channel-chunk], so that a lane's 8 (4) consecutive K values are one
//     16-byte LDS read;
//   * the weight chunk [tap][BM][channel-chunk] is copied to LDS with 16-byte accesses
//     from the pre-packed weight tensor [phase][ky][kx][co][ci];
//   * every wave then issues, per tap, BM/16 A reads + 4 B reads and BM/16 x 4 MFMAs
//     (v_mfma_f32_16x16x32 for 16-bit types).
// Pixel rows are padded to 80 bytes in LDS so that the 16 lanes of a read group fall
// on different 16-byte bank slots.  Accumulation is float; the epilogue runs on the
// accumulators and stores once.
//
// float32 tensors run on the half-precision matrix instruction as well: each operand tensor is multiplied by the
// power of two that brings its largest magnitude (measured on the device in front of the kernel, sg_absmax.h)
// to 2^14 .. 2^15, every element is cut into two halves h0 + h1 that carry 22 of its 24 significant bits, both
// operands are staged to LDS as two such planes, and a product is accumulated from the three partial products
// h0 h0', h0 h1', h1 h0' (what is dropped lies below 2^-21 of the product; measured against float64 the layers
// stay at the 1e-6 relative error of a float32 evaluation).  Three 16x16x32 instructions take 48 cycles per 32
// channels where eight 16x16x4 float ones take 256.  A chunk is 32 channels (2 x 64 bytes per pixel row, pitch
// 144); the weight chunk is staged three taps at a time so that the tile fits LDS.
// Layers with enough pixels to give every CU a workgroup of 512 pixels run with such tiles (each of the eight waves
// then owns 64 channels x 64 pixels): a staged weight feeds twice the multiplies, which halves the weight bytes
// that the workgroups pull through L2 - the larger share of what they read.
#include "sg_absmax.h"
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kPitch = 80;        // bytes per LDS pixel row: 64 payload + 16 pad
constexpr int kChunkBytes = 64;   // K payload per pixel per chunk
constexpr int kPixels = 256;      // output pixels per workgroup
constexpr int kPixelsWide = 512;  // ... of the float kernel on layers with enough pixels to fill the chip with such tiles
constexpr int kPixelsHalf = 128;  // ... of the float kernel at stride 2, where LDS holds the footprint of 128 output pixels only
constexpr int kThreads = 256;
constexpr int kPitchSplit = 128;  // bytes per LDS row of a float tensor split into 2 half planes (swizzled, no padding)
constexpr int kSplitTaps = 3;     // taps of the weight chunk held in LDS at a time (split float path)
constexpr size_t kPreparedHeader = 256;   // prepared weights: [largest magnitude in word 1 of 64][split rows]

__device__ __forceinline__ int floor_div_i(int a, int b);

template <class T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<__half> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// The float path multiplies in half precision without giving up float accuracy: every operand element v (times a
// power of two that brings the tensor's largest magnitude to 2^14 .. 2^15, see sg_absmax.h) is cut into
// v = h0 + h1 + e with h0 = half(v), h1 = half(v - h0), both rounded to nearest, so |e| <= 2^-22 |v| (an element
// more than 2^17 below the tensor's maximum keeps an absolute error of 2^-40 of that maximum instead); the
// products h0 h0', h0 h1' and h1 h0' are accumulated in float by the matrix instructions, and what is dropped,
// h1 h1', is below 2^-22 of the product as well.  A non-finite input poisons the outputs it reaches.
// The three partial products of the split float path, smallest first.
__device__ __forceinline__ void mma_split(const uint4 (&a)[2], const uint4 (&b)[2], f32x4_t& c) {
    using M = Mma<__half>;
    M::run(a[1], b[0], c);
    M::run(a[0], b[1], c);
    M::run(a[0], b[0], c);
}

struct Tile {
    int split_k;          // > 1: input-channel chunks are divided over blockIdx.z, partial sums added atomically
    int px;               // pixel slots per workgroup: 256, or 512 (float kernel, large layers)
    int tw, th, imgs;     // pixel tile in phase coordinates; tw * th * imgs <= px (slots beyond it are idle)
    int lw, lh;           // LDS footprint of one image's tile (pixels)
    int tiles_a, tiles_b, tiles_img;
    int vec_epi;          // epilogue through LDS with 16-byte global accesses (8 pixels of one channel per lane)
    // float path (sg_absmax.h): largest magnitudes of x, w and in_scale (bit patterns); x and the packed weights as
    // split rows; chunks of 32 input channels
    const unsigned* amax;
    const unsigned* wmax;   // the four words whose word 1 is the weights' largest magnitude (amax, or the prepared weights' header)
    const unsigned char* xs;
    const unsigned char* ws;
    int chunks;
    int64_t w_rows;       // rows of the whole packed weight tensor
    int max_taps;         // most taps of any phase
    int x_double;         // LDS holds two input footprints (the next chunk's is written while this one is read)
};

// Pixel slot (0..255) of a workgroup -> column, row and image within its tile; ri >= t.imgs marks an idle slot.
__device__ __forceinline__ void slot_decode(const Tile& t, int slot, int& rb, int& ra, int& ri) {
    rb = slot % t.tw;
    const int q = slot / t.tw;
    ra = q % t.th;
    ri = q / t.th;
}


// Copy the footprint [imgs][lh][lw] (top-left input pixel (iy0, ix0), images img0..) of channels
// c0 .. c0 + chunk of a dense NCHW tensor into LDS as [pixel][channel], multiplied by
// scale[n, c] when given.  Pixels outside the tensor are zero.  A lane owns pixels (coalesced
// along x) and walks the chunk's channels: the coordinates are decoded once, a channel step is
// one pointer increment.
template <class T>
__device__ __forceinline__ void stage_pixels(unsigned char* s_dst, const T* __restrict__ x, const float* __restrict__ scale,
                                             int n_total, int c_total, int h, int w, int c0, int img0, int imgs, int iy0,
                                             int ix0, int lh, int lw) {
    constexpr int KCH = kChunkBytes / (int)sizeof(T);
    const int64_t plane = (int64_t)h * w;
    const int pixels = imgs * lh * lw;
    for (int pix = threadIdx.x; pix < pixels; pix += kThreads) {
        const int rx = pix % lw;
        const int ry = (pix / lw) % lh;
        const int ri = pix / (lw * lh);
        const int iy = iy0 + ry, ix = ix0 + rx, n = img0 + ri;
        const bool inside = iy >= 0 && iy < h && ix >= 0 && ix < w && n < n_total;
        unsigned char* dst = s_dst + pix * kPitch;
        if (!inside) {
#pragma unroll
            for (int q = 0; q < kChunkBytes / 16; q++) *reinterpret_cast<uint4*>(dst + q * 16) = make_uint4(0, 0, 0, 0);
            continue;
        }
        const T* src = x + ((int64_t)n * c_total + c0) * plane + (int64_t)iy * w + ix;
        const float* sc = scale ? scale + (int64_t)n * c_total + c0 : nullptr;
        const int left = c_total - c0;   // channels that exist from c0 on
        if constexpr (sizeof(T) == 2) {
#pragma unroll 4
            for (int cp = 0; cp < KCH / 2; cp++) {
                const int c = 2 * cp;
                T h0 = Elem<T>::store(0), h1 = Elem<T>::store(0);
                if (c < left) h0 = src[(int64_t)c * plane];
                if (c + 1 < left) h1 = src[(int64_t)(c + 1) * plane];
                if (sc) {
                    if (c < left) h0 = Elem<T>::store(Elem<T>::load(h0) * sc[c]);
                    if (c + 1 < left) h1 = Elem<T>::store(Elem<T>::load(h1) * sc[c + 1]);
                }
                *reinterpret_cast<unsigned*>(dst + cp * 4) =
                    (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            }
        } else {
#pragma unroll 4
            for (int c = 0; c < KCH; c++) {
                float v = 0;
                if (c < left) {
                    v = Elem<T>::load(src[(int64_t)c * plane]);
                    if (sc) v *= sc[c];
                }
                *reinterpret_cast<float*>(dst + c * 4) = v;
            }
        }
    }
}

__device__ __forceinline__ int floor_div_i(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// Vector form of stage_pixels for rows that are a whole number of 16-byte packets (w % (16 /
// sizeof(T)) == 0, 16-byte aligned base): a lane loads one aligned 16-byte packet (8 or 4
// consecutive pixels) per channel - two channels at a time for 16-bit types - scales it, and
// scatters its pixels into the [pixel][channel] image.  Lanes run over channels fastest, so a
// wave's LDS stores land in consecutive banks, and its loads touch one or two cache lines per
// channel plane.
template <class T>
__device__ __forceinline__ void stage_pixels_vec(unsigned char* s_dst, const T* __restrict__ x,
                                                 const float* __restrict__ scale, int n_total, int c_total, int h,
                                                 int w, int c0, int img0, int imgs, int iy0, int ix0, int lh, int lw) {
    constexpr int VEC = 16 / (int)sizeof(T);            // pixels per packet
    constexpr int KCH = kChunkBytes / (int)sizeof(T);   // channels per chunk
    constexpr int CPL = (sizeof(T) == 2) ? 2 : 1;       // channels a lane handles at once (one LDS dword)
    constexpr int CG = KCH / CPL;                       // = 16 lane slots across the chunk's channels
    static_assert(CG == 16, "chunk is 16 dwords per pixel");
    const int64_t plane = (int64_t)h * w;
    const int vx0 = floor_div_i(ix0, VEC);
    const int nv = floor_div_i(ix0 + lw - 1, VEC) - vx0 + 1;
    const int items = imgs * lh * nv * CG;
    for (int it = threadIdx.x; it < items; it += kThreads) {
        const int cg = it & (CG - 1);
        int r = it >> 4;
        const int v = r % nv; r /= nv;
        const int ry = r % lh;
        const int ri = r / lh;
        const int iy = iy0 + ry, n = img0 + ri, px0 = (vx0 + v) * VEC;
        const int c = c0 + cg * CPL;
        const bool ok = iy >= 0 && iy < h && n < n_total && px0 >= 0 && px0 < w && c < c_total;
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        if (ok) {
            const T* src = x + ((int64_t)n * c_total + c) * plane + (int64_t)iy * w + px0;
            va = *reinterpret_cast<const uint4*>(src);
            if (CPL == 2 && c + 1 < c_total) vb = *reinterpret_cast<const uint4*>(src + plane);
        }
        unsigned out[VEC];   // one dword per pixel: (channel c, channel c+1) or one float
        if constexpr (sizeof(T) == 2) {
            const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
            if (scale && ok) {
                const float s0 = scale[(int64_t)n * c_total + c];
                const float s1 = (c + 1 < c_total) ? scale[(int64_t)n * c_total + c + 1] : 0.f;
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    const unsigned short ea = (unsigned short)(wa[j >> 1] >> (16 * (j & 1)));
                    const unsigned short eb = (unsigned short)(wb[j >> 1] >> (16 * (j & 1)));
                    const T ha = Elem<T>::store(Elem<T>::load(__builtin_bit_cast(T, ea)) * s0);
                    const T hb = Elem<T>::store(Elem<T>::load(__builtin_bit_cast(T, eb)) * s1);
                    out[j] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; j += 2) {
                    out[j] = (wa[j >> 1] & 0xffffu) | (wb[j >> 1] << 16);
                    out[j + 1] = (wa[j >> 1] >> 16) | (wb[j >> 1] & 0xffff0000u);
                }
            }
        } else {
            const unsigned wa[4] = {va.x, va.y, va.z, va.w};
            const float s0 = (scale && ok) ? scale[(int64_t)n * c_total + c] : 1.f;
#pragma unroll
            for (int j = 0; j < VEC; j++) out[j] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, wa[j]) * s0);
        }
        unsigned char* row = s_dst + ((ri * lh + ry) * lw) * kPitch + cg * 4;
        const int rx0 = px0 - ix0;
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const int rx = rx0 + j;
            if (rx >= 0 && rx < lw) *reinterpret_cast<unsigned*>(row + rx * kPitch) = out[j];
        }
    }
}

template <class T>
__device__ __forceinline__ void stage_tile(unsigned char* s_dst, const T* __restrict__ x, const float* __restrict__ scale,
                                           int n_total, int c_total, int h, int w, int c0, int img0, int imgs, int iy0,
                                           int ix0, int lh, int lw) {
    const bool vec = (w % (16 / (int)sizeof(T)) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    if (vec) stage_pixels_vec<T>(s_dst, x, scale, n_total, c_total, h, w, c0, img0, imgs, iy0, ix0, lh, lw);
    else stage_pixels<T>(s_dst, x, scale, n_total, c_total, h, w, c0, img0, imgs, iy0, ix0, lh, lw);
}

// Threads per workgroup: the float kernel has room for one workgroup per CU (LDS), so it runs eight waves, two
// per SIMD, each on half of the pixel slots a wave of the 16-bit kernels covers: one wave's LDS traffic and
// staging arithmetic then overlaps the other's multiplies.
template <class T> constexpr int conv_threads() { return sizeof(T) == 4 ? 512 : 256; }

template <class T, int M_TILES, int PX = kPixels>
__global__ void __launch_bounds__(conv_threads<T>(), sizeof(T) == 4 ? 1 : 2) conv2d_mfma_kernel(sg_conv2d_params p, Tile t) {
    using S = float;
    constexpr bool SPLIT = sizeof(T) == 4;              // float: two half planes per operand
    constexpr int THREADS = conv_threads<T>(), WAVES = THREADS / 64;
    constexpr int NT = PX / (16 * WAVES);               // 16-pixel n-tiles per wave: 4 or 2 (4 with 512 pixels)
    constexpr int WPX = 16 * NT;                        // pixel slots per wave
    constexpr int KCH = SPLIT ? 32 : kChunkBytes / (int)sizeof(T);   // channels per chunk
    constexpr int PITCH = SPLIT ? kPitchSplit : kPitch;
    constexpr int BM = 16 * M_TILES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;

    // ---- which tile ------------------------------------------------------------------
    const int phase = blockIdx.z % p.num_phases, ksplit = blockIdx.z / p.num_phases;
    const sg_conv2d_phase ph = p.phase[phase];
    const int AH = (p.oh - ph.oy0 + p.out_step - 1) / p.out_step;   // phase-local output extent
    const int AW = (p.ow - ph.ox0 + p.out_step - 1) / p.out_step;
    int tile = blockIdx.x;
    const int tb = tile % t.tiles_b; tile /= t.tiles_b;
    const int ta = tile % t.tiles_a; tile /= t.tiles_a;
    const int img0 = tile * t.imgs;
    const int a0 = ta * t.th, b0 = tb * t.tw;
    if (a0 >= AH || b0 >= AW) return;
    const int m0 = blockIdx.y * BM;
    const int taps = ph.nky * ph.nkx;
    const int lw = (t.tw - 1) * p.in_step + ph.nkx;
    const int lh = (t.th - 1) * p.in_step + ph.nky;
    const int iy0 = a0 * p.in_step + ph.dy0, ix0 = b0 * p.in_step + ph.dx0;

    unsigned char* s_in = smem;
    unsigned char* s_w = smem + (size_t)t.imgs * t.lh * t.lw * PITCH;   // sized for the largest phase

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w + (int64_t)ph.w_offset;

    // ---- per-lane pixel slots of the wave's four n-tiles -----------------------------------
    int b_addr[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        int rb, ra, ri;
        slot_decode(t, wave * WPX + j * 16 + lrow, rb, ra, ri);
        if (ri >= t.imgs) rb = ra = ri = 0;   // idle slot: any staged pixel will do, its products are dropped
        b_addr[j] = ((ri * lh + ra * p.in_step) * lw + rb * p.in_step) * PITCH + lgrp * 16;
    }
    const int a_addr = lrow * PITCH + lgrp * 16;

    f32x4_t acc[M_TILES][NT];
#pragma unroll
    for (int m = 0; m < M_TILES; m++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const bool w_vec = (p.ci % (16 / (int)sizeof(T)) == 0);

    // float path: the inverses of the powers of two that brought both operands into the range of a half
    float inv_x = 1.f, inv_w = 1.f;
    if constexpr (SPLIT) {
        float mul;
        half_range_scale(t.amax, 0, 2, p.in_scale != nullptr, mul, inv_x);
        half_range_scale(t.wmax, 1, 3, false, mul, inv_w);
    }

    // ---- split float path -------------------------------------------------------------------------------------
    // LDS images of split rows: 128 bytes per row (pixel or weight row: 2 planes x 32 channels), no padding; the
    // 16-byte slot s of row r lives at slot s ^ ((r >> 1) & 7), which puts the 16 rows of a ds_read_b128 lane group
    // on 16 different bank slots at any row offset (a tap shift moves all rows alike).
    auto split_addr = [](int row, int slot) { return row * 128 + (((slot ^ (row >> 1)) & 7) << 4); };
    int b_row[NT];            // footprint pixel of the lane's column in n-tile j, tap (0, 0)
    int a_lane = 0;           // plane 0 of weight row lrow (rows m * 16 + lrow, tap * BM + ...: same swizzle); plane 1 at ^ 64
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < NT; j++) b_row[j] = b_addr[j] / PITCH;
        a_lane = split_addr(lrow, lgrp);
    }
    // Multiply the staged taps [g0, g0 + gtaps) of the weight chunk.  Steps (tap, n-tile) run in one fixed order; the
    // pixel fragments of the next step and - two steps before a tap ends - the weight fragments of the next tap are
    // read into a second register set while the 12 matrix instructions of the current step run.
    // (the images are named by their byte offsets in smem, so that the reads stay LDS instructions)
    auto split_multiply = [&](int w_off, int x_off, int g0, int gtaps, auto&& beside_first_tap) {
        const unsigned char* s_w = smem + w_off;
        const unsigned char* s_in = smem + x_off;
        int tap_row[kSplitTaps];
#pragma unroll
        for (int tl = 0; tl < kSplitTaps; tl++) {
            const int tap = g0 + tl, ky = tap / ph.nkx;
            tap_row[tl] = ky * lw + (tap - ky * ph.nkx);
        }
        uint4 a[2][M_TILES][2], b[2][2];
        auto read_a = [&](int tl, uint4 (&dst)[M_TILES][2]) {
#pragma unroll
            for (int m = 0; m < M_TILES; m++) {
                const unsigned char* r = s_w + (tl * BM + m * 16) * 128;
                dst[m][0] = *reinterpret_cast<const uint4*>(r + a_lane);
                dst[m][1] = *reinterpret_cast<const uint4*>(r + (a_lane ^ 64));
            }
        };
        auto read_b = [&](int tl, int j, uint4 (&dst)[2]) {
            const int ad = split_addr(b_row[j] + tap_row[tl], lgrp);
            dst[0] = *reinterpret_cast<const uint4*>(s_in + ad);
            dst[1] = *reinterpret_cast<const uint4*>(s_in + (ad ^ 64));
        };
        read_a(0, a[0]);
        read_b(0, 0, b[0]);
#pragma unroll
        for (int tl = 0; tl < kSplitTaps; tl++) {
            if (tl >= gtaps) break;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                constexpr int kAhead = NT >= 2 ? NT - 2 : 0;          // the step that fetches the next tap's weights
                const int step = tl * NT + j;
                if (j + 1 < NT) read_b(tl, j + 1, b[(step + 1) & 1]);
                else if (tl + 1 < gtaps) read_b(tl + 1, 0, b[(step + 1) & 1]);
                if (j == kAhead && tl + 1 < gtaps) read_a(tl + 1, a[(tl + 1) & 1]);
                if (tl == 0) beside_first_tap(j);                     // (global loads of the caller, a few per step)
                __builtin_amdgcn_sched_barrier(0);                    // (reads first: they have the whole step to land)
#pragma unroll
                for (int m = 0; m < M_TILES; m++) mma_split(a[tl & 1][m], b[step & 1], acc[m][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if constexpr (SPLIT) {
        // Both operands come as split rows (sg_absmax.h): per pixel (or weight row) and chunk of 32 channels 128
        // contiguous bytes, the LDS row.  A pipeline over (chunk, tap group) stages.  The weights of the next stage
        // (kSplitTaps taps) go from global memory straight into the LDS buffer the current stage does not read
        // (global_load_lds: no staging registers, no LDS write instructions; a lane's LDS destination is fixed by
        // its position in the wave, so the swizzle is applied by choosing which 16 bytes it fetches).  The footprint
        // of the next chunk is requested into registers at the first stage of the current one and written - to a
        // second footprint buffer when LDS has room for it (t.x_double), else after one more barrier - at its last.
        constexpr int W_PIECES = kSplitTaps * BM * 8;                         // 16-byte pieces of a weight stage
        constexpr int W_IT = (W_PIECES + THREADS - 1) / THREADS;              // ... per lane
        constexpr int X_IT = PX == kPixels ? 8 : 10;                          // input pieces per lane held in registers
        const unsigned char* __restrict__ xs = t.xs;
        const unsigned char* __restrict__ ws = t.ws + (ph.w_offset / p.ci) * (int64_t)128;   // (this phase's first row)
        const int x_items = t.imgs * lh * lw * 8;
        const int64_t x_plane8 = (int64_t)p.h * p.w_in * 8;   // 16-byte units from one chunk of a sample to the next
        const int chunks = t.chunks, chunks_per_split = (chunks + t.split_k - 1) / t.split_k;
        const int cb_begin = ksplit * chunks_per_split, cb_end = min(chunks, cb_begin + chunks_per_split);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // weights: LDS piece e = tid + k THREADS is slot e & 7 of row e >> 3 = (tap in stage) * BM + channel
        int w_rel[W_IT], w_tl[W_IT];       // source offset from the stage's first row; tap in stage (kSplitTaps: none)
#pragma unroll
        for (int k = 0; k < W_IT; k++) {
            const int e = tid + k * THREADS, row = e >> 3, o = row % BM, tl = row / BM;
            const int piece = ((e & 7) ^ (row >> 1)) & 7;
            w_tl[k] = (e < W_PIECES && m0 + o < p.co) ? tl : kSplitTaps;
            w_rel[k] = (tl * p.co + m0 + o) * 128 + piece * 16;
        }
        auto w_issue = [&](int cb, int g0, int w_off) {
            const int gtaps = min(kSplitTaps, taps - g0);
            const unsigned char* base = ws + ((int64_t)cb * t.w_rows + (int64_t)g0 * p.co) * 128;
#pragma unroll
            for (int k = 0; k < W_IT; k++) {
                if (w_tl[k] < gtaps)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + w_rel[k]),
                                                     (__attribute__((address_space(3))) void*)(smem + w_off + (k * THREADS + wave_u * 64) * 16),
                                                     16, 0, 0);
            }
        };
        // Where piece e of the footprint image comes from in the split tensor, in 16-byte units for chunk 0 (-1:
        // outside the tensor, zero).  The same for every chunk, so the divisions are done once per lane.
        // (the wide tile keeps the offsets in 32 bits - the host picks it for tensors below 2^31 such units only)
        using XOff = typename std::conditional<PX == kPixels, int64_t, int>::type;
        auto x_where = [&](int e) -> XOff {
            const int row = e >> 3, piece = ((e & 7) ^ (row >> 1)) & 7;
            int pix = row;
            const int rx = pix % lw; pix /= lw;
            const int ry = pix % lh;
            const int ri = pix / lh;
            const int iy = iy0 + ry, ix = ix0 + rx, n = img0 + ri;
            if (e >= x_items || iy < 0 || iy >= p.h || ix < 0 || ix >= p.w_in || n >= p.n) return -1;
            return (XOff)((((int64_t)n * chunks * p.h + iy) * p.w_in + ix) * 8 + piece);
        };
        XOff xoff[X_IT];
        uint4 xreg[X_IT];
#pragma unroll
        for (int k = 0; k < X_IT; k++) xoff[k] = x_where(tid + k * THREADS);
        auto x_fetch = [&](int cb, XOff where) -> uint4 {   // (no branch: a piece outside is read at the tensor's start and dropped)
            const uint4 v = *reinterpret_cast<const uint4*>(xs + (where < 0 ? 0 : (where + (int64_t)cb * x_plane8) * 16));
            return where < 0 ? make_uint4(0, 0, 0, 0) : v;
        };
        auto x_load = [&](int cb) {
#pragma unroll
            for (int k = 0; k < X_IT; k++) xreg[k] = x_fetch(cb, xoff[k]);
        };
        // The next chunk's footprint pieces, dealt over the NT steps of a stage's first tap (as one batch in front of
        // the stage they queue in the texture path, 16 cycles per load of a wave, and the waves start multiplying one
        // after the other); cb < 0: none wanted in this stage.
        int x_next = -1;
        auto x_load_part = [&](int j) {
            if (x_next < 0) return;
#pragma unroll
            for (int k = 0; k < X_IT; k++) {
                if ((k * NT) / X_IT != j) continue;
                xreg[k] = x_fetch(x_next, xoff[k]);
            }
        };
        auto x_commit = [&](int x_off, int cb) {
            unsigned char* s_in = smem + x_off;
#pragma unroll
            for (int k = 0; k < X_IT; k++) {
                const int e = tid + k * THREADS;
                if (e < x_items) *reinterpret_cast<uint4*>(s_in + e * 16) = xreg[k];
            }
            for (int e0 = X_IT * THREADS + tid; e0 < x_items; e0 += 4 * THREADS) {   // (footprints beyond the register window)
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = x_fetch(cb, x_where(e0 + u * THREADS));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * THREADS;
                    if (e < x_items) *reinterpret_cast<uint4*>(s_in + e * 16) = v[u];
                }
            }
        };
        auto advance = [&](int& cb, int& g0) {
            g0 += kSplitTaps;
            if (g0 >= taps) { g0 = 0; cb++; }
        };
        // every lane's copies into LDS have landed (then a barrier, then the reads)
        auto copies_landed = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        const int x_bytes = t.imgs * t.lh * t.lw * PITCH, w_bytes = min(t.max_taps, kSplitTaps) * BM * PITCH;
        const int x_second = t.x_double ? x_bytes : 0, w_first = (t.x_double ? 2 : 1) * x_bytes;
        auto s_x = [&](int cb) { return (cb & 1) ? x_second : 0; };            // byte offsets in smem
        auto s_wb = [&](int stage) { return w_first + ((stage & 1) ? w_bytes : 0); };
        int cb = cb_begin, g0 = 0;          // the stage being multiplied
        int cb1 = cb, g1 = g0;              // the stage after it
        advance(cb1, g1);
        int sidx = 0;                       // stage counter: picks the weight buffer
        if (cb < cb_end) {
            w_issue(cb, g0, s_wb(0));
            x_load(cb);
            x_commit(s_x(cb), cb);
            copies_landed();
            __syncthreads();
        }
        // One barrier per stage (two where a single footprint buffer is rewritten).  The weight buffer that receives
        // the next stage was last read in the stage before this one, which every wave has left.
        while (cb < cb_end) {
            const bool first = g0 == 0, last = cb1 != cb;
            if (cb1 < cb_end) w_issue(cb1, g1, s_wb(sidx + 1));
            x_next = (first && cb + 1 < cb_end) ? cb + 1 : -1;
            split_multiply(s_wb(sidx), s_x(cb), g0, min(kSplitTaps, taps - g0), x_load_part);
            if (last && cb1 < cb_end) {
                if (!t.x_double) __syncthreads();   // every wave has read the footprint of this chunk
                x_commit(s_x(cb1), cb1);
            }
            copies_landed();
            __syncthreads();
            cb = cb1; g0 = g1;
            advance(cb1, g1);
            sidx++;
        }
    }
    const int chunks = (p.ci + KCH - 1) / KCH, chunks_per_split = (chunks + t.split_k - 1) / t.split_k;
    const int c_begin = ksplit * chunks_per_split * KCH;
    const int c_end = SPLIT ? c_begin : min(p.ci, c_begin + chunks_per_split * KCH);
    for (int c0 = c_begin; c0 < c_end; c0 += KCH) {
        __syncthreads();   // everyone is done reading the previous chunk
        if constexpr (!SPLIT) {
        stage_tile<T>(s_in, x, p.in_scale, p.n, p.ci, p.h, p.w_in, c0, img0, t.imgs, iy0, ix0, lh, lw);
        // ---- stage the weight chunk: [tap][BM][channel] ----------------------------------------
        if (w_vec) {
            for (int e = tid; e < taps * BM * 4; e += THREADS) {
                const int q = e & 3, row = e >> 2;               // row = tap * BM + o
                const int o = row % BM, tap = row / BM;
                const int c = c0 + q * (16 / (int)sizeof(T));
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m0 + o < p.co && c < p.ci)
                    v = *reinterpret_cast<const uint4*>(w + ((int64_t)tap * p.co + m0 + o) * p.ci + c);
                *reinterpret_cast<uint4*>(s_w + row * kPitch + q * 16) = v;
            }
        } else {
            for (int e = tid; e < taps * BM * KCH; e += THREADS) {
                const int cc = e % KCH, row = e / KCH;
                const int o = row % BM, tap = row / BM;
                T v = Elem<T>::store(0);
                if (m0 + o < p.co && c0 + cc < p.ci) v = w[((int64_t)tap * p.co + m0 + o) * p.ci + c0 + cc];
                *reinterpret_cast<T*>(s_w + row * kPitch + cc * sizeof(T)) = v;
            }
        }
        __syncthreads();

        // ---- multiply --------------------------------------------------------------------------
        for (int ky = 0; ky < ph.nky; ky++) {
            for (int kx = 0; kx < ph.nkx; kx++) {
                const int tap = ky * ph.nkx + kx;
                uint4 a[M_TILES];
#pragma unroll
                for (int m = 0; m < M_TILES; m++)
                    a[m] = *reinterpret_cast<const uint4*>(s_w + (tap * BM + m * 16) * kPitch + a_addr);
                const int tap_off = (ky * lw + kx) * kPitch;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const uint4 b = *reinterpret_cast<const uint4*>(s_in + b_addr[j] + tap_off);
#pragma unroll
                    for (int m = 0; m < M_TILES; m++) {
                        if constexpr (!SPLIT) Mma<T>::run(a[m], b, acc[m][j]);
                    }
                }
            }
        }
        }   // !SPLIT
    }

    if constexpr (SPLIT) {
#pragma unroll
        for (int m = 0; m < M_TILES; m++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[m][j] = acc[m][j] * inv_x * inv_w;
    }

    // ---- epilogue, wide form: the accumulators go through LDS as [pixel][channel] floats and come back as
    // 8 consecutive pixels of one channel per lane, so that the output (and the reduction operand, and the
    // noise) move in 16-byte packets and the per-channel constants are fetched once per packet ------------
    if (t.vec_epi) {
        constexpr int EP = BM + 4;   // row pitch in floats: the 8 rows of a 16-byte store group fall on all 32 banks
        __syncthreads();             // every wave has left the multiply loop: its LDS is free
        float* s_out = reinterpret_cast<float*>(smem) + wave * WPX * EP;
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int m = 0; m < M_TILES; m++)
                *reinterpret_cast<f32x4_t*>(s_out + (j * 16 + lrow) * EP + m * 16 + lgrp * 4) = acc[m][j];
        __syncthreads();
        const S noise_gain = (p.noise && p.noise_gain) ? *p.noise_gain : (S)1;
        S red[M_TILES];
#pragma unroll
        for (int m = 0; m < M_TILES; m++) red[m] = 0;
#pragma unroll
        for (int it = 0; it < (NT / 2) * M_TILES; it++) {
            const int cblk = it % M_TILES, half = it / M_TILES;
            const int chl = cblk * 16 + lrow, ch = m0 + chl;
            int rb, ra, ri;
            slot_decode(t, wave * WPX + half * 32 + lgrp * 8, rb, ra, ri);
            const int b = b0 + rb, a = a0 + ra, n = img0 + ri;
            if (ch >= p.co || a >= AH || b >= AW || n >= p.n || ri >= t.imgs) continue;
            S v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = s_out[(half * 32 + lgrp * 8 + q) * EP + chl];
            const int64_t off = (((int64_t)n * p.co + ch) * p.oh + a) * p.ow + b;
            if (p.reduce_out) {
                S sum = 0;
                constexpr int PN = Pack<T>::N;
#pragma unroll
                for (int h = 0; h < 8 / PN; h++) {
                    const Pack<T> rx = load_pack((const T*)p.reduce_x + off + h * PN);
#pragma unroll
                    for (int q = 0; q < PN; q++) sum += v[h * PN + q] * (S)Elem<T>::load(rx.v[q]);
                }
                if (t.imgs == 1) red[cblk] += sum;
                else if (sum != 0) atomicAdd(p.reduce_out + (int64_t)n * p.co + ch, sum);
            }
            const S sc = p.out_scale ? p.out_scale[(int64_t)n * p.co + ch] : (S)1;
            const S bs = p.bias ? (S)Elem<T>::load(((const T*)p.bias)[ch]) : (S)0;
            S nz[8];
            if (p.noise) {
                const float4* np = reinterpret_cast<const float4*>(p.noise + (int64_t)n * p.noise_stride_n + (int64_t)a * p.ow + b);
                const float4 n0 = np[0], n1 = np[1];
                nz[0] = n0.x; nz[1] = n0.y; nz[2] = n0.z; nz[3] = n0.w; nz[4] = n1.x; nz[5] = n1.y; nz[6] = n1.z; nz[7] = n1.w;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                S x = v[q];
                if (p.out_scale) x *= sc;
                if (p.noise) x += nz[q] * noise_gain;
                if (p.bias) x += bs;
                if (p.act == 3) x = (x > 0) ? x : x * p.alpha;
                x *= p.gain;
                if (p.clamp >= 0) x = fminf(fmaxf(x, -p.clamp), p.clamp);
                v[q] = x;
            }
            constexpr int PN = Pack<T>::N;
#pragma unroll
            for (int h = 0; h < 8 / PN; h++) {
                Pack<T> out;
#pragma unroll
                for (int q = 0; q < PN; q++) out.v[q] = Elem<T>::store(v[h * PN + q]);
                store_pack((T*)p.y + off + h * PN, out);
            }
        }
        if (p.reduce_out && t.imgs == 1) {   // lanes l, l+16, l+32, l+48 hold the same channels; then the four waves
            __syncthreads();
            float* s_red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int m = 0; m < M_TILES; m++) {
                S r = red[m];
                r += __shfl_xor(r, 16, 64);
                r += __shfl_xor(r, 32, 64);
                if (lgrp == 0) s_red[wave * BM + m * 16 + lrow] = r;
            }
            __syncthreads();
            if (tid < BM && m0 + tid < p.co && img0 < p.n) {
                float r = 0.f;
#pragma unroll
                for (int wv = 0; wv < WAVES; wv++) r += s_red[wv * BM + tid];
                const int rows = t.tiles_a * t.tiles_b * p.num_phases;
                const int row = (ta * t.tiles_b + tb) * p.num_phases + phase;
                p.reduce_out[((int64_t)img0 * rows + row) * p.co + m0 + tid] = r;
            }
        }
        return;
    }

    // ---- epilogue: scale, noise, bias, activation, gain, clamp; one store per element ----------
    const S noise_gain = (p.noise && p.noise_gain) ? *p.noise_gain : (S)1;
    T* __restrict__ y = (T*)p.y;
    S red[M_TILES][4];   // (4 accumulator rows) per-lane share of sum_pixels acc * reduce_x
#pragma unroll
    for (int m = 0; m < M_TILES; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[m][r] = 0;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        int rb, ra, ri;
        slot_decode(t, wave * WPX + j * 16 + lrow, rb, ra, ri);
        const int n = img0 + ri;
        const int a = a0 + ra, b = b0 + rb;
        if (t.imgs > 1) {   // n-tiles of a low-resolution tile belong to different images: flush per n-tile
#pragma unroll
            for (int m = 0; m < M_TILES; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[m][r] = 0;
        }
        const bool live = !(a >= AH || b >= AW || n >= p.n || ri >= t.imgs);
        if (live) {
        const int oy = ph.oy0 + a * p.out_step, ox = ph.ox0 + b * p.out_step;
        S nz = 0;
        if (p.noise) nz = p.noise[(int64_t)n * p.noise_stride_n + (int64_t)oy * p.ow + ox] * noise_gain;
#pragma unroll
        for (int m = 0; m < M_TILES; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ch = m0 + m * 16 + lgrp * 4 + r;
                if (ch >= p.co) continue;
                S v = acc[m][j][r];
                if (p.reduce_out) red[m][r] += v * Elem<T>::load(((const T*)p.reduce_x)[(((int64_t)n * p.co + ch) * p.oh + oy) * p.ow + ox]);
                if (p.out_scale) v *= p.out_scale[(int64_t)n * p.co + ch];
                v += nz;
                if (p.bias) v += Elem<T>::load(((const T*)p.bias)[ch]);
                if (p.act == 3) v = (v > 0) ? v : v * p.alpha;
                v *= p.gain;
                if (p.clamp >= 0) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                T* dst = y + (((int64_t)n * p.co + ch) * p.oh + oy) * p.ow + ox;
                if constexpr (sizeof(T) == 4) {
                    if (t.split_k > 1) { atomicAdd(reinterpret_cast<float*>(dst), v); continue; }
                }
                *dst = Elem<T>::store(v);
            }
        }
        }   // live
        if (p.reduce_out && t.imgs > 1) {   // 16 lanes = 16 pixels of one image: shuffle-reduce, one atomic per channel
            const int n_tile = img0 + (wave * WPX + j * 16) / (t.tw * t.th);   // tw * th is a multiple of 16 here
#pragma unroll
            for (int m = 0; m < M_TILES; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    S v = red[m][r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    const int ch = m0 + m * 16 + lgrp * 4 + r;
                    if (lrow == 0 && ch < p.co && n_tile < p.n && v != 0) atomicAdd(p.reduce_out + (int64_t)n_tile * p.co + ch, v);
                }
        }
    }
    if (p.reduce_out && t.imgs == 1) {
        // The whole tile lies in one image: reduce over the wave's lanes, then over the four waves through LDS,
        // and store one partial row per (image, tile, phase) - no atomics, the caller sums the rows.
        __syncthreads();   // every wave has left the MFMA loop; LDS can be reused
        float* s_red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int m = 0; m < M_TILES; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                S v = red[m][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (lrow == 0) s_red[wave * BM + m * 16 + lgrp * 4 + r] = v;
            }
        __syncthreads();
        if (tid < BM && m0 + tid < p.co && img0 < p.n) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < WAVES; wv++) v += s_red[wv * BM + tid];
            const int rows = t.tiles_a * t.tiles_b * p.num_phases;
            const int row = (ta * t.tiles_b + tb) * p.num_phases + phase;
            p.reduce_out[((int64_t)img0 * rows + row) * p.co + m0 + tid] = v;
        }
    }
}

__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ dst, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) dst[i] = 0u;
}

int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) s++;
    return s;
}

// Phase-local output extent (largest phase) decides the pixel tile: up to 32 columns, then rows, then images, 256
// pixel slots in all.  Extents below 32 that are no power of two get a tile of exactly their size (a 9 x 12 phase
// of an up-sampling layer would fill 42% of a 16 x 16 tile), except where the per-image reduction is wanted,
// whose shuffle groups need tiles of whole 16-pixel groups.
Tile make_tile_of(const sg_conv2d_params& p, int px, int* out_max_taps) {
    const bool split = p.dtype == SG_F32;   // LDS rows and weight-chunk rows of the kernel that runs this dtype
    int AH = 0, AW = 0, max_taps = 0, max_nky = 0, max_nkx = 0;
    for (int i = 0; i < p.num_phases; i++) {
        const sg_conv2d_phase& ph = p.phase[i];
        AH = max(AH, (p.oh - ph.oy0 + p.out_step - 1) / p.out_step);
        AW = max(AW, (p.ow - ph.ox0 + p.out_step - 1) / p.out_step);
        max_taps = max(max_taps, ph.nky * ph.nkx);
        max_nky = max(max_nky, ph.nky);
        max_nkx = max(max_nkx, ph.nkx);
    }
    const bool exact = p.reduce_x == nullptr;
    Tile t;
    t.split_k = 1;
    t.vec_epi = 0;
    t.px = px;
    t.tw = 32;
    while (t.tw > 1 && t.tw / 2 >= AW) t.tw /= 2;              // smallest power of two covering AW, at most 32
    if (exact && AW < t.tw) t.tw = AW;
    // Polyphase layers have extents like 33 (the even phase of 32 -> 65): tiles of equal width instead of 32 + 1
    const bool balance = exact && p.num_phases > 1;
    if (balance && AW > t.tw) t.tw = (AW + (AW + t.tw - 1) / t.tw - 1) / ((AW + t.tw - 1) / t.tw);
    int rest = px / t.tw;
    t.th = (p.in_step == 1) ? px / 32 : 4;
    if (exact && t.tw < 32) t.th = 16;                          // low resolution: whole images where they fit
    if (t.th > rest) t.th = rest;
    if (exact && AH < t.th) t.th = AH;
    while (!exact && t.th > 1 && t.th / 2 >= AH) t.th /= 2;
    if (balance && AH > t.th) t.th = (AH + (AH + t.th - 1) / t.th - 1) / ((AH + t.th - 1) / t.th);
    t.lw = (t.tw - 1) * p.in_step + max_nkx;
    if (split) {
        // Rows of 128 bytes: pick the rows x images per tile that fill the most pixel slots within the LDS budget
        // (slots beyond imgs * th * tw idle).  Tiles that feed the per-image reduction keep whole 16-pixel groups.
        // (696 rows beside the two weight buffers: with more, the up-sampling layers get tiles of more images and
        // fewer rows, whose larger share of halo rows costs more than the fuller slots return: 0.75 against 0.55 ms
        // at 512 -> 512, 32^2 -> 64^2, batch 16)
        const size_t budget = 152 * 1024 / 144 - 2 * (size_t)min(max_taps, kSplitTaps) * 64;
        int best_th = 1, best_imgs = 1, best_used = 0;
        for (int th = t.th; th >= 1; th = (th > 1) ? (th + 1) / 2 : 0) {
            if (!exact && (t.tw * th) % 16 != 0) break;
            const size_t lh = (size_t)(th - 1) * p.in_step + max_nky;
            int imgs = min(rest / th, max(p.n, 1));
            while (imgs > 1 && imgs * lh * t.lw > budget) imgs--;
            if (imgs * lh * t.lw > budget) continue;
            if (imgs * th * t.tw > best_used) { best_used = imgs * th * t.tw; best_th = th; best_imgs = imgs; }
        }
        t.th = best_th;
        t.imgs = best_imgs;
        t.lh = (t.th - 1) * p.in_step + max_nky;
    } else {
    for (;; t.th = (t.th + 1) / 2) {                            // whole-image tiles of strided layers can outgrow LDS
        t.imgs = min(rest / t.th, max(p.n, 1));
        t.lh = (t.th - 1) * p.in_step + max_nky;
        if (!exact || t.th == 1 || ((size_t)t.imgs * t.lh * t.lw + (size_t)max_taps * 64) * kPitch <= 144 * 1024) break;
    }
    }
    t.tiles_b = (AW + t.tw - 1) / t.tw;
    t.tiles_a = (AH + t.th - 1) / t.th;
    t.tiles_img = (p.n + t.imgs - 1) / t.imgs;

    *out_max_taps = max_taps;
    return t;
}

// The float kernel takes 512 pixels per workgroup (every staged weight then feeds twice the multiplies, and the
// weights are what its workgroups read most of) where such tiles are filled and still give every CU a workgroup.
// At stride 2 the footprint of 128 output pixels fills LDS: such layers run with 128 pixel slots (one 16-pixel tile
// per wave) instead of multiplying 128 idle slots along.
Tile make_tile(const sg_conv2d_params& p, int* out_max_taps) {
    const bool wide_ok = p.dtype == SG_F32 && p.co >= 64 &&
                         split_bytes((int64_t)p.n * p.h * p.w_in, p.ci) / 16 < (size_t)INT32_MAX;
    const auto blocks = [&](const Tile& t) {
        return (int64_t)t.tiles_a * t.tiles_b * t.tiles_img * p.num_phases * ((p.co + 63) / 64);
    };
    if (wide_ok) {
        const Tile t = make_tile_of(p, kPixelsWide, out_max_taps);
        if (t.imgs * t.th * t.tw >= kPixelsWide * 3 / 4 && blocks(t) >= 256) return t;
    }
    const Tile t = make_tile_of(p, kPixels, out_max_taps);
    if (wide_ok && p.in_step == 2 && t.imgs * t.th * t.tw <= kPixelsHalf && blocks(t) >= 256) {
        const Tile h = make_tile_of(p, kPixelsHalf, out_max_taps);
        if (h.imgs * h.th * h.tw == t.imgs * t.th * t.tw) return h;
    }
    return t;
}

template <class T>
int launch(const sg_conv2d_params& p, hipStream_t stream) {
    int max_taps = 0;
    Tile t = make_tile(p, &max_taps);
    const int64_t tiles = (int64_t)t.tiles_a * t.tiles_b * t.tiles_img;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    // Low-resolution float layers: few pixel tiles and a long K loop.  Without an epilogue the sum over input
    // channels can be divided over workgroups that add into a zeroed output.
    constexpr bool SPLIT = sizeof(T) == 4;
    constexpr int KCH = SPLIT ? 32 : kChunkBytes / (int)sizeof(T);
    constexpr size_t PITCH = SPLIT ? kPitchSplit : kPitch;
    const int chunks = (p.ci + KCH - 1) / KCH;
    const bool plain = !p.in_scale && !p.out_scale && !p.bias && !p.noise && p.act == 1 && p.gain == 1.f && p.clamp < 0 &&
                       !p.reduce_out;
    int max_split = 1;
    if (sizeof(T) == 4 && plain)
        while (max_split < 16 && chunks / (max_split * 2) >= (SPLIT ? 1 : 2)) max_split *= 2;
    // 64 output channels per workgroup (every staged pixel then feeds four times the multiplies), unless that
    // leaves most of the 256 CUs without a workgroup even with the channel sum divided
    const int64_t tiles_px = tiles * p.num_phases;
    const int m_tiles = (t.px != kPixels || (p.co > 16 && tiles_px * ((p.co + 63) / 64) * max_split >= 256)) ? 4 : 1;
    const int BM = 16 * m_tiles;
    size_t lds = ((size_t)t.imgs * t.lh * t.lw + (size_t)max_taps * BM) * PITCH;
    t.max_taps = max_taps;
    t.x_double = 0;
    if constexpr (SPLIT) {   // two weight buffers of kSplitTaps taps, and two input footprints where they fit
        const size_t x_rows = (size_t)t.imgs * t.lh * t.lw, w_rows = (size_t)std::min(max_taps, kSplitTaps) * BM;
        t.x_double = (2 * x_rows + 2 * w_rows) * PITCH <= 160 * 1024;
        lds = ((t.x_double ? 2 : 1) * x_rows + 2 * w_rows) * PITCH;
    }
    SG_REQUIRE(lds <= 160 * 1024, SG_ERR_UNSUPPORTED, "conv2d: tile needs %zu bytes of LDS", lds);
    t.split_k = 1;
    {
        const int64_t blocks = tiles_px * ((p.co + BM - 1) / BM);
        while (t.split_k < max_split && blocks * t.split_k < 512) t.split_k *= 2;
        if (t.split_k > 1) {
            // (a kernel rather than hipMemsetAsync: memset nodes recorded into a HIP graph were seen to act on
            // the first replay only)
            const size_t words = (size_t)p.n * p.co * p.oh * p.ow * sizeof(T) / 4;
            const unsigned zblocks = (unsigned)std::min<size_t>((words + 255) / 256, 4096);
            hipLaunchKernelGGL(zero_words_kernel, dim3(zblocks), dim3(256), 0, stream, (unsigned*)p.y, words);
            if (int rc = check_launch("sg_conv2d (zero)")) return rc;
        }
    }
    // Wide epilogue: whole 8-pixel packets of a row must be addressable with aligned 16-byte accesses.
    t.vec_epi = t.split_k == 1 && p.out_step == 1 && p.num_phases == 1 && p.phase[0].oy0 == 0 && p.phase[0].ox0 == 0 &&
                t.px >= kPixels && p.ow % 8 == 0 && t.tw % 8 == 0 && aligned16(p.y) && (!p.noise || (aligned16(p.noise) && p.noise_stride_n % 4 == 0)) &&
                (!p.reduce_out || aligned16(p.reduce_x));
    if (t.vec_epi) lds = std::max(lds, (size_t)t.px * (BM + 4) * sizeof(float));
    t.amax = t.wmax = nullptr;
    t.xs = t.ws = nullptr;
    t.chunks = 0;
    t.w_rows = 0;
    if constexpr (SPLIT) {
        int64_t w_elems = 0;
        for (int i = 0; i < p.num_phases; i++)
            w_elems = std::max<int64_t>(w_elems, p.phase[i].w_offset + (int64_t)p.phase[i].nky * p.phase[i].nkx * p.co * p.ci);
        const bool prepared = p.w_prepared != nullptr;
        SG_REQUIRE(!prepared || (reinterpret_cast<uintptr_t>(p.w_prepared) & 255u) == 0, SG_ERR_INVALID_ARGUMENT,
                   "conv2d: w_prepared must be 256-byte aligned");
        if (int rc = absmax_measure(stream, (const float*)p.x, (int64_t)p.n * p.ci * p.h * p.w_in, prepared ? nullptr : (const float*)p.w,
                                    w_elems, p.in_scale, (int64_t)p.n * p.ci, nullptr, 0, &t.amax))
            return rc;
        const int64_t pixels = (int64_t)p.h * p.w_in, w_rows = w_elems / p.ci;
        const size_t x_bytes = split_bytes((int64_t)p.n * pixels, p.ci), need = x_bytes + (prepared ? 0 : split_bytes(w_rows, p.ci));
        unsigned char* work = (unsigned char*)p.workspace;
        if (work != nullptr) {
            SG_REQUIRE(p.workspace_bytes >= need && (reinterpret_cast<uintptr_t>(work) & 255u) == 0, SG_ERR_INVALID_ARGUMENT,
                       "conv2d: workspace of %zu bytes (need %zu, 256-byte aligned)", p.workspace_bytes, need);
        } else {
            work = (unsigned char*)split_workspace(stream, need);
            if (work == nullptr) return (int)hipErrorOutOfMemory;
        }
        if (int rc = split_pixels(stream, (const float*)p.x, p.in_scale, p.n, p.ci, pixels, t.amax, 0, 2, work)) return rc;
        t.xs = work;
        if (prepared) {
            t.wmax = (const unsigned*)p.w_prepared;
            t.ws = (const unsigned char*)p.w_prepared + kPreparedHeader;
        } else {
            if (int rc = split_rows(stream, (const float*)p.w, w_rows, p.ci, t.amax, 1, work + x_bytes)) return rc;
            t.wmax = t.amax;
            t.ws = work + x_bytes;
        }
        t.chunks = (int)split_chunks(p.ci);
        t.w_rows = w_rows;
    }
    dim3 grid((unsigned)tiles, (unsigned)((p.co + BM - 1) / BM), (unsigned)(p.num_phases * t.split_k));
    auto kernel = (m_tiles == 4) ? conv2d_mfma_kernel<T, 4> : conv2d_mfma_kernel<T, 1>;
    if constexpr (SPLIT) {
        if (t.px > kPixels) kernel = conv2d_mfma_kernel<T, 4, kPixelsWide>;
        if (t.px < kPixels) kernel = conv2d_mfma_kernel<T, 4, kPixelsHalf>;
    }
    if (lds > 64 * 1024) {   // opt in to large dynamic LDS (idempotent, cheap)
        hipError_t e = allow_large_lds((const void*)kernel);
        if (e != hipSuccess) return fail((int)e, "conv2d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(conv_threads<T>()), lds, stream, p, t);
    return check_launch("sg_conv2d");
}


// ---------------------------------------------------------------------------
// Weight gradient:  dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * g_scale[n,cg]
//                                          * x[n,cx,a*step+dy0+ky,b*step+dx0+kx] * x_scale[n,cx]
// GEMM view per tap: M = cg, N = cx, K = pixels.  Both operands are staged exactly like the
// forward kernel's input, [pixel][channel], so K (pixels) runs down the LDS rows; the
// 16-bit path fetches its fragments with the transposing LDS read (4 pixel rows x 16 channel
// columns per 16 lanes, a lane receives 4 consecutive pixels of one channel), which also
// makes the per-tap pixel shift free.  A workgroup owns one (cg chunk, cx chunk) pair and
// strides over pixel tiles (split K); every wave reduces 64 pixels per tile into taps x
// (chunk/16)^2 accumulator tiles and adds them to dw with float atomics at the end.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;

template <class T> struct WgradFrag;      // fragment fetch + MFMA for one 16-channel tile pair
template <class T16> struct WgradFrag16 {
    // pixel k-step of 32: lane group g takes pixels 8g..8g+7, two transposing reads of 4 pixels
    static constexpr int kPixelsPerStep = 32;
    static __device__ __forceinline__ uint4 fetch(const unsigned char* base, const int (&pix_off)[2], int chan_byte) {
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(base + pix_off[0] + chan_byte));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(base + pix_off[1] + chan_byte));
        uint4 r;
        r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
        r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
        return r;
    }
};

template <class T, int MAX_TAPS>
__global__ void __launch_bounds__(kThreads) conv2d_wgrad_kernel(sg_conv2d_wgrad_params p, Tile t, int tiles_total) {
    constexpr int KCH = kChunkBytes / (int)sizeof(T);   // channels per chunk: 32 (16-bit) or 16 (f32)
    constexpr int NT = KCH / 16;                        // 16-channel tiles per chunk
    constexpr bool HALF = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int cg0 = blockIdx.y * KCH, cx0 = blockIdx.z * KCH;
    const int taps = p.nky * p.nkx;
    const int lw = (t.tw - 1) * p.step + p.nkx, lh = (t.th - 1) * p.step + p.nky;
    unsigned char* s_x = smem;
    unsigned char* s_g = smem + (size_t)t.imgs * lh * lw * kPitch;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ g = (const T*)p.g;

    f32x4_t acc[MAX_TAPS][NT][NT];
#pragma unroll
    for (int tp = 0; tp < MAX_TAPS; tp++)
#pragma unroll
        for (int m = 0; m < NT; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // LDS byte offsets (without the tap shift) of the pixels this lane addresses, per k-step.
    //  16-bit: 2 k-steps x 2 reads; lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3.
    //  f32   : 16 k-steps; lane group g addresses pixel 4*kstep + g, channel lrow.
    constexpr int KSTEPS = HALF ? 2 : 16;
    int gx_off[KSTEPS][2], gg_off[KSTEPS][2];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int slot = HALF ? (wave * 64 + ks * 32 + lgrp * 8 + hf * 4 + (lrow >> 2)) : (wave * 64 + ks * 4 + lgrp);
            int rb, ra, ri;
            slot_decode(t, slot, rb, ra, ri);
            const int chan = HALF ? (lrow & 3) * 8 : lrow * 4;
            gx_off[ks][hf] = ((ri * lh + ra * p.step) * lw + rb * p.step) * kPitch + chan;
            gg_off[ks][hf] = slot * kPitch + chan;
        }

    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        int rest = tile;
        const int tb = rest % t.tiles_b; rest /= t.tiles_b;
        const int ta = rest % t.tiles_a; rest /= t.tiles_a;
        const int img0 = rest * t.imgs;
        const int a0 = ta * t.th, b0 = tb * t.tw;
        __syncthreads();
        stage_tile<T>(s_x, x, p.x_scale, p.n, p.cx, p.h, p.w_in, cx0, img0, t.imgs, a0 * p.step + p.dy0,
                        b0 * p.step + p.dx0, lh, lw);
        stage_tile<T>(s_g, g, p.g_scale, p.n, p.cg, p.gh, p.gw, cg0, img0, t.imgs, a0, b0, t.th, t.tw);
        __syncthreads();

#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            uint4 a[NT];
#pragma unroll
            for (int m = 0; m < NT; m++) {
                if constexpr (HALF) a[m] = WgradFrag16<T>::fetch(s_g, gg_off[ks], m * 32);
                else a[m].x = *reinterpret_cast<const unsigned*>(s_g + gg_off[ks][0]);
            }
#pragma unroll
            for (int tp = 0; tp < MAX_TAPS; tp++) {
                if (tp < taps) {
                    const int ky = tp / p.nkx, kx = tp - ky * p.nkx;
                    const unsigned char* sx = s_x + (ky * lw + kx) * kPitch;
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        uint4 b;
                        if constexpr (HALF) b = WgradFrag16<T>::fetch(sx, gx_off[ks], n * 32);
                        else b.x = *reinterpret_cast<const unsigned*>(sx + gx_off[ks][0]);
#pragma unroll
                        for (int m = 0; m < NT; m++) {
                            if constexpr (HALF) Mma<T>::run(a[m], b, acc[tp][m][n]);
                            else acc[tp][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a[m].x),
                                                                                       __builtin_bit_cast(float, b.x), acc[tp][m][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- reduce into dw[tap][cg][cx] --------------------------------------------------------------
    float* __restrict__ dw = p.dw;
#pragma unroll
    for (int tp = 0; tp < MAX_TAPS; tp++) {
        if (tp < taps) {
#pragma unroll
            for (int m = 0; m < NT; m++)
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int cg = cg0 + m * 16 + lgrp * 4 + r, cx = cx0 + n * 16 + lrow;
                        if (cg < p.cg && cx < p.cx) atomicAdd(dw + ((int64_t)tp * p.cg + cg) * p.cx + cx, acc[tp][m][n][r]);
                    }
        }
    }
}

template <class T>
int launch_wgrad(const sg_conv2d_wgrad_params& p, hipStream_t stream) {
    Tile t{};
    t.tw = 32;
    while (t.tw > 4 && t.tw / 2 >= p.gw) t.tw /= 2;
    int rest = kPixels / t.tw;
    t.th = (p.step == 1) ? 8 : 4;
    if (t.th > rest) t.th = rest;
    while (t.th > 1 && t.th / 2 >= p.gh) t.th /= 2;
    t.imgs = rest / t.th;
    t.lw = (t.tw - 1) * p.step + p.nkx;
    t.lh = (t.th - 1) * p.step + p.nky;
    t.tiles_b = (p.gw + t.tw - 1) / t.tw;
    t.tiles_a = (p.gh + t.th - 1) / t.th;
    t.tiles_img = (p.n + t.imgs - 1) / t.imgs;
    const int64_t tiles = (int64_t)t.tiles_a * t.tiles_b * t.tiles_img;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: too many tiles");
    constexpr int KCH = kChunkBytes / (int)sizeof(T);
    const int chunks_g = (p.cg + KCH - 1) / KCH, chunks_x = (p.cx + KCH - 1) / KCH;
    // split K so that about four workgroups per CU exist; every split ends in taps*KCH*KCH atomics
    int64_t split = (1024 + (int64_t)chunks_g * chunks_x - 1) / ((int64_t)chunks_g * chunks_x);
    if (split > tiles) split = tiles;
    if (split < 1) split = 1;
    const size_t lds = ((size_t)t.imgs * t.lh * t.lw + kPixels) * kPitch;
    SG_REQUIRE(lds <= 160 * 1024, SG_ERR_UNSUPPORTED, "conv2d_wgrad: tile needs %zu bytes of LDS", lds);
    const int taps = p.nky * p.nkx;
    dim3 grid((unsigned)split, (unsigned)chunks_g, (unsigned)chunks_x);
    auto kernel = (taps <= 1) ? conv2d_wgrad_kernel<T, 1> : (taps <= 4) ? conv2d_wgrad_kernel<T, 4> : conv2d_wgrad_kernel<T, 9>;
    if (lds > 64 * 1024) {
        hipError_t e = allow_large_lds((const void*)kernel);
        if (e != hipSuccess) return fail((int)e, "conv2d_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, stream, p, t, (int)tiles);
    return check_launch("sg_conv2d_wgrad");
}

}  // namespace
}  // namespace sg

extern "C" int sg_conv2d(const sg_conv2d_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->n >= 0 && p->ci >= 1 && p->co >= 1 && p->h >= 1 && p->w_in >= 1 && p->oh >= 1 && p->ow >= 1,
               SG_ERR_INVALID_ARGUMENT, "conv2d: bad tensor sizes");
    SG_REQUIRE(p->in_step >= 1 && p->in_step <= 2 && p->out_step >= 1 && p->out_step <= 2, SG_ERR_INVALID_ARGUMENT,
               "conv2d: in_step and out_step must be 1 or 2");
    SG_REQUIRE(p->num_phases >= 1 && p->num_phases <= 4, SG_ERR_INVALID_ARGUMENT, "conv2d: 1..4 phases");
    for (int i = 0; i < p->num_phases; i++) {
        const sg_conv2d_phase& ph = p->phase[i];
        SG_REQUIRE(ph.nky >= 1 && ph.nkx >= 1 && ph.nky * ph.nkx <= 16, SG_ERR_INVALID_ARGUMENT,
                   "conv2d: phase %d has %dx%d taps (at most 16 in total)", i, ph.nky, ph.nkx);
        SG_REQUIRE(ph.oy0 >= 0 && ph.ox0 >= 0 && ph.oy0 < p->out_step && ph.ox0 < p->out_step && ph.w_offset >= 0,
                   SG_ERR_INVALID_ARGUMENT, "conv2d: phase %d has a bad origin", i);
    }
    SG_REQUIRE(p->act == 1 || p->act == 3, SG_ERR_UNSUPPORTED, "conv2d: fused activation must be linear (1) or lrelu (3)");
    SG_REQUIRE((int64_t)p->n * p->ci * p->h * p->w_in <= INT32_MAX && (int64_t)p->n * p->co * p->oh * p->ow <= INT32_MAX,
               SG_ERR_INVALID_ARGUMENT, "conv2d: tensor is too large");
    if (p->n == 0) return SG_OK;
    SG_REQUIRE(p->x && p->w && p->y, SG_ERR_NULL_POINTER, "conv2d: x, w and y must not be NULL");
    if (p->addend != nullptr) {
        if (conv1x1_few_eligible(*p)) return conv1x1_few_launch(*p, (hipStream_t)stream);
        SG_REQUIRE(p->mask_y == nullptr && pointwise_eligible(*p), SG_ERR_UNSUPPORTED,
                   "conv2d: no kernel adds a tensor to the result for this geometry");
        return pointwise_launch(*p, (hipStream_t)stream);
    }
    if (conv1x1_few_eligible(*p)) return conv1x1_few_launch(*p, (hipStream_t)stream);
    if (band3x3_eligible(*p)) return band3x3_launch(*p, (hipStream_t)stream);
    SG_REQUIRE(p->mask_y == nullptr, SG_ERR_UNSUPPORTED, "conv2d: no kernel applies an input mask for this geometry");
    if (conv3x3_eligible(*p)) return conv3x3_launch(*p, (hipStream_t)stream);
    if (conv_up2_eligible(*p)) return conv_up2_launch(*p, (hipStream_t)stream);
    if (pointwise_eligible(*p)) return pointwise_launch(*p, (hipStream_t)stream);
    switch (p->dtype) {
        case SG_F32: return launch<float>(*p, (hipStream_t)stream);
        case SG_F16: return launch<__half>(*p, (hipStream_t)stream);
        case SG_BF16: return launch<__hip_bfloat16>(*p, (hipStream_t)stream);
    }
    return fail(SG_ERR_UNSUPPORTED, "conv2d: dtype %d is not supported (f32, f16, bf16)", p->dtype);
}

extern "C" int sg_conv2d_wgrad(const sg_conv2d_wgrad_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->n >= 0 && p->cx >= 1 && p->cg >= 1 && p->h >= 1 && p->w_in >= 1 && p->gh >= 1 && p->gw >= 1,
               SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: bad tensor sizes");
    SG_REQUIRE(p->step >= 1 && p->step <= 2, SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: step must be 1 or 2");
    SG_REQUIRE(p->nky >= 1 && p->nkx >= 1 && p->nky * p->nkx <= 9, SG_ERR_INVALID_ARGUMENT,
               "conv2d_wgrad: %dx%d taps (at most 9 in total)", p->nky, p->nkx);
    SG_REQUIRE((int64_t)p->n * p->cx * p->h * p->w_in <= INT32_MAX && (int64_t)p->n * p->cg * p->gh * p->gw <= INT32_MAX,
               SG_ERR_INVALID_ARGUMENT, "conv2d_wgrad: tensor is too large");
    if (p->n == 0) return SG_OK;
    SG_REQUIRE(p->x && p->g && p->dw, SG_ERR_NULL_POINTER, "conv2d_wgrad: x, g and dw must not be NULL");
    if (wgrad1x1_few_eligible(*p)) return wgrad1x1_few_launch(*p, (hipStream_t)stream);
    if (wgrad_rows_eligible(*p)) return wgrad_rows_launch(*p, (hipStream_t)stream);
    SG_REQUIRE(p->mask_y == nullptr, SG_ERR_UNSUPPORTED, "conv2d_wgrad: no kernel applies a gradient mask for this geometry");
    if (wgrad3x3_eligible(*p)) return wgrad3x3_launch(*p, (hipStream_t)stream);
    if (wgrad1x1_eligible(*p)) return wgrad1x1_launch(*p, (hipStream_t)stream);
    if (wgrad_f32_eligible(*p)) return wgrad_f32_launch(*p, (hipStream_t)stream);
    switch (p->dtype) {
        case SG_F32: return launch_wgrad<float>(*p, (hipStream_t)stream);
        case SG_F16: return launch_wgrad<__half>(*p, (hipStream_t)stream);
        case SG_BF16: return launch_wgrad<__hip_bfloat16>(*p, (hipStream_t)stream);
    }
    return fail(SG_ERR_UNSUPPORTED, "conv2d_wgrad: dtype %d is not supported (f32, f16, bf16)", p->dtype);
}

extern "C" size_t sg_conv2d_workspace_bytes(const sg_conv2d_params* p) {
    if (p == nullptr || p->dtype != SG_F32 || p->n <= 0 || p->ci <= 0 || p->co <= 0 || p->num_phases < 1 || p->num_phases > 4) return 0;
    int64_t w_elems = 0;
    for (int i = 0; i < p->num_phases; i++)
        w_elems = std::max<int64_t>(w_elems, p->phase[i].w_offset + (int64_t)p->phase[i].nky * p->phase[i].nkx * p->co * p->ci);
    return sg::split_bytes((int64_t)p->n * p->h * p->w_in, p->ci) + (p->w_prepared ? 0 : sg::split_bytes(w_elems / p->ci, p->ci));
}

namespace {
int64_t packed_weight_elems(const sg_conv2d_params& p) {
    int64_t w_elems = 0;
    for (int i = 0; i < p.num_phases; i++)
        w_elems = std::max<int64_t>(w_elems, p.phase[i].w_offset + (int64_t)p.phase[i].nky * p.phase[i].nkx * p.co * p.ci);
    return w_elems;
}
}  // namespace

extern "C" size_t sg_conv2d_prepared_weight_bytes(const sg_conv2d_params* p) {
    if (p == nullptr || p->dtype != SG_F32 || p->ci <= 0 || p->co <= 0 || p->num_phases < 1 || p->num_phases > 4) return 0;
    return sg::kPreparedHeader + sg::split_bytes(packed_weight_elems(*p) / p->ci, p->ci);
}

extern "C" int sg_conv2d_prepare_weights(const sg_conv2d_params* p, void* out, size_t out_bytes, sg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const size_t need = sg_conv2d_prepared_weight_bytes(p);
    SG_REQUIRE(need != 0 && p->w != nullptr, SG_ERR_INVALID_ARGUMENT, "conv2d_prepare_weights: a float32 geometry with weights is needed");
    SG_REQUIRE(out != nullptr && out_bytes >= need && (reinterpret_cast<uintptr_t>(out) & 255u) == 0, SG_ERR_INVALID_ARGUMENT,
               "conv2d_prepare_weights: out of %zu bytes (need %zu, 256-byte aligned)", out_bytes, need);
    const int64_t w_elems = packed_weight_elems(*p);
    unsigned* header = (unsigned*)out;
    if (int rc = sg::absmax_into(stream, (const float*)p->w, w_elems, header)) return rc;
    return sg::split_rows(stream, (const float*)p->w, w_elems / p->ci, p->ci, header, 1, (unsigned char*)out + sg::kPreparedHeader);
}

extern "C" size_t sg_conv2d_wgrad_workspace_bytes(const sg_conv2d_wgrad_params* p) {
    if (p == nullptr || !sg::wgrad_f32_eligible(*p) || p->n <= 0) return 0;
    return sg::split_bytes((int64_t)p->n * p->h * p->w_in, p->cx) + sg::split_bytes((int64_t)p->n * p->gh * p->gw, p->cg);
}

extern "C" int sg_conv2d_reduce_rows(const sg_conv2d_params* p) {
    if (p == nullptr || p->num_phases < 1 || p->num_phases > 4) return -1;
    if (sg::conv1x1_few_eligible(*p)) return sg::conv1x1_few_reduce_rows(*p);
    if (sg::band3x3_eligible(*p)) return sg::band3x3_reduce_rows(*p);
    if (sg::conv3x3_eligible(*p)) return sg::conv3x3_reduce_rows(*p);
    int taps = 0;
    const sg::Tile t = sg::make_tile(*p, &taps);
    return (t.imgs == 1) ? t.tiles_a * t.tiles_b * p->num_phases : 1;
}

extern "C" int sg_conv2d_supports_addend(const sg_conv2d_params* p) {
    if (p == nullptr || p->num_phases < 1 || p->num_phases > 4) return 0;
    sg_conv2d_params q = *p;
    if (q.addend == nullptr) q.addend = q.y;       // (asked before the addend exists: any aligned pointer stands in)
    if (sg::conv1x1_few_eligible(q)) return 1;
    return (q.mask_y == nullptr && sg::pointwise_eligible(q)) ? 1 : 0;
}

extern "C" int sg_conv2d_supports_mask(const sg_conv2d_params* p) {
    if (p == nullptr || p->num_phases < 1 || p->num_phases > 4) return 0;
    sg_conv2d_params q = *p;
    if (q.mask_y == nullptr) q.mask_y = q.x;
    return (sg::conv1x1_few_eligible(q) || sg::band3x3_eligible(q)) ? 1 : 0;
}

extern "C" int sg_conv2d_wgrad_supports_mask(const sg_conv2d_wgrad_params* p) {
    if (p == nullptr) return 0;
    sg_conv2d_wgrad_params q = *p;
    if (q.mask_y == nullptr) q.mask_y = q.g;
    return (sg::wgrad1x1_few_eligible(q) || sg::wgrad_rows_eligible(q)) ? 1 : 0;
}

extern "C" const char* sg_conv2d_kernel(const sg_conv2d_params* p) {
    if (p == nullptr) return "";
    if (sg::conv1x1_few_eligible(*p)) return "conv1x1_few_kernel";
    if (sg::band3x3_eligible(*p)) return "band3x3_kernel";
    if (sg::conv_up2_eligible(*p)) return "conv_up2_kernel";
    if (sg::pointwise_eligible(*p)) return "pointwise_kernel";
    if (!sg::conv3x3_eligible(*p)) return "conv2d_mfma_kernel";
    return sg::conv3x3_resident(*p) ? "conv3x3_resident_kernel" : "conv3x3_kernel";
}

extern "C" const char* sg_conv2d_wgrad_kernel(const sg_conv2d_wgrad_params* p) {
    if (p == nullptr) return "";
    if (sg::wgrad1x1_few_eligible(*p)) return "wgrad1x1_few_kernel";
    if (sg::wgrad_rows_eligible(*p)) return "wgrad_rows_kernel";
    if (sg::wgrad3x3_eligible(*p)) return "wgrad3x3_kernel";
    if (sg::wgrad1x1_eligible(*p)) return "wgrad1x1_kernel";
    return sg::wgrad_f32_eligible(*p) ? "wgrad_f32_kernel" : "conv2d_wgrad_kernel";
}

// ---------------------------------------------------------------------------------------------------------
// sg_conv2d_reduced: sg_conv2d with the style-gradient reduction finished on the device, for callers that may
// not allocate (stream capture, serving loops).  The partial rows that sg_conv2d leaves in a caller-provided
// workspace go to a fixed array in device memory instead, and one more kernel adds them up.
namespace sg {
namespace {

constexpr int64_t kScratchFloats = SG_CONV2D_REDUCED_MAX_PARTIALS;   // 4 MiB on every device the library runs on
static __device__ __align__(256) float scratch[kScratchFloats];

// Device address of `scratch` on the current device (every device holds its own copy of the array); resolved once
// per device.  Returns nullptr and sets the error text on failure.
float* scratch_address() {
    static std::atomic<float*> cached[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        fail((int)(e != hipSuccess ? e : hipErrorInvalidDevice), "conv2d_reduced: no usable current device");
        return nullptr;
    }
    float* ptr = cached[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr) {
            fail((int)(e != hipSuccess ? e : hipErrorInvalidSymbol), "conv2d_reduced: cannot resolve the scratch array: %s",
                 hipGetErrorString(e));
            return nullptr;
        }
        cached[dev].store(ptr, std::memory_order_release);
    }
    return ptr;
}

// out[n, c] = sum over rows of partial[n, row, c]; one lane per (n, c), rows added in order.
__global__ void __launch_bounds__(256) sum_rows_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, int rows, int c) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * c) return;
    const int64_t img = i / c, ch = i - img * c;
    const float* src = partial + img * rows * c + ch;
    float s = 0.f;
    for (int r = 0; r < rows; r++) s += src[(int64_t)r * c];
    out[i] = s;
}

}  // namespace
}  // namespace sg

extern "C" int sg_init(void) {
    // Everything the entry points would otherwise look up on their first call on a device and that is not a stream
    // operation: the address of the scratch array (hipGetSymbolAddress must not run inside a stream capture).
    return sg::scratch_address() != nullptr ? SG_OK : SG_ERR_UNSUPPORTED;
}

extern "C" int sg_conv2d_reduced(const sg_conv2d_params* p, float* reduce_sum, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->reduce_x != nullptr && reduce_sum != nullptr, SG_ERR_NULL_POINTER,
               "conv2d_reduced: reduce_x and reduce_sum must not be NULL");
    SG_REQUIRE(p->reduce_out == nullptr, SG_ERR_INVALID_ARGUMENT,
               "conv2d_reduced: reduce_out must be NULL (the partial rows live in the library's own array)");
    SG_REQUIRE(p->workspace != nullptr || sg_conv2d_workspace_bytes(p) == 0, SG_ERR_INVALID_ARGUMENT,
               "conv2d_reduced: a float32 call needs p->workspace (%zu bytes): this entry point allocates nothing",
               sg_conv2d_workspace_bytes(p));
    SG_REQUIRE(p->n >= 0 && p->co >= 1, SG_ERR_INVALID_ARGUMENT, "conv2d: bad tensor sizes");
    if (p->n == 0) return SG_OK;
    const int rows = sg_conv2d_reduce_rows(p);
    SG_REQUIRE(rows >= 1, SG_ERR_INVALID_ARGUMENT, "conv2d_reduced: bad geometry");
    const int64_t partials = (int64_t)p->n * rows * p->co;
    SG_REQUIRE(partials <= kScratchFloats, SG_ERR_UNSUPPORTED,
               "conv2d_reduced: %lld partial sums (n %d x %d rows x %d channels) exceed the fixed array of %lld; use sg_conv2d "
               "with a workspace", (long long)partials, p->n, rows, p->co, (long long)kScratchFloats);
    float* ws = scratch_address();
    if (ws == nullptr) return SG_ERR_UNSUPPORTED;
    // Cleared by a kernel, not hipMemsetAsync: a memset node recorded into a HIP graph was seen to act on the first
    // replay only, and the kernels that add into the array with atomics need it zero on every replay.
    {
        const size_t words = (size_t)partials;
        const unsigned zblocks = (unsigned)std::min<size_t>((words + 255) / 256, 4096);
        hipLaunchKernelGGL(zero_words_kernel, dim3(zblocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)ws, words);
        if (int rc = check_launch("sg_conv2d_reduced (zero)")) return rc;
    }
    sg_conv2d_params q = *p;
    q.reduce_out = ws;
    const int rc = sg_conv2d(&q, stream);
    if (rc != SG_OK) return rc;
    const int64_t lanes = (int64_t)p->n * p->co;
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, reduce_sum, p->n, rows, p->co);
    return check_launch("sg_conv2d_reduced");
}
