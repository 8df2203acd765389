// upfirdn2d: zero-stuff, pad / crop, FIR-filter and decimate every (sample, channel)
// plane.  Bandwidth-bound: the ideal traffic is one read of x and one write of y.
//
// Two kernels:
//   * tiled   - rows contiguous in memory (stride_w == 1), filter at most 4x4 taps,
//               up/down in {1,2}: one workgroup of 256 lanes produces a TILE_W x TILE_H
//               output tile of one plane.  The input footprint of the tile (with its
//               filter halo) is staged once through LDS with row-contiguous loads, the
//               (flipped) taps sit in LDS too, and every lane then reads only LDS.
//   * generic - any strides (channels-last, slices), any filter size, any integer
//               up/down factors; one lane per output element, taps walked in global
//               memory.
// Tap indexing follows the reference kernels (torch_utils/ops/upfirdn2d.cu:42-89 and
// :141-195): for output o, mid = o * down + up - 1 - pad0, first input sample
// floor(mid / up), first tap (in + 1) * up - mid - 1 of the flipped filter, then every
// up-th tap.
#include "sg_common.h"
#include "sg_upfirdn2d.h"

namespace sg {
namespace {

__device__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// ---------------------------------------------------------------------------
// Generic kernel.
template <class T>
__global__ void __launch_bounds__(256) upfirdn2d_generic_kernel(sg_upfirdn2d_params p, int64_t total) {
    using S = typename Elem<T>::acc_t;
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const bool channels_fastest = (p.in_stride_c == 1 && p.in_c > 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        // Walk the output in the order its memory is laid out, so that stores coalesce.
        int64_t r = idx;
        int c, ox;
        if (channels_fastest) {
            c = (int)(r % p.in_c); r /= p.in_c;
            ox = (int)(r % p.out_w); r /= p.out_w;
        } else {
            ox = (int)(r % p.out_w); r /= p.out_w;
        }
        const int oy = (int)(r % p.out_h); r /= p.out_h;
        if (!channels_fastest) { c = (int)(r % p.in_c); r /= p.in_c; }
        const int n = (int)r;

        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int in_x0 = min(max(floor_div(mid_x, p.up_x), 0), p.in_w);
        const int in_y0 = min(max(floor_div(mid_y, p.up_y), 0), p.in_h);
        const int w = min(max(floor_div(mid_x + p.f_w, p.up_x), 0), p.in_w) - in_x0;
        const int h = min(max(floor_div(mid_y + p.f_h, p.up_y), 0), p.in_h) - in_y0;
        int tap_x0 = mid_x + p.f_w - (in_x0 + 1) * p.up_x;
        int tap_y0 = mid_y + p.f_h - (in_y0 + 1) * p.up_y;
        int step_x = -p.up_x, step_y = -p.up_y;
        if (p.flip) {
            tap_x0 = p.f_w - 1 - tap_x0;
            tap_y0 = p.f_h - 1 - tap_y0;
            step_x = p.up_x;
            step_y = p.up_y;
        }
        const T* xp = x + (int64_t)n * p.in_stride_n + (int64_t)c * p.in_stride_c;
        S acc = (S)0;
        for (int j = 0; j < h; j++) {
            const T* row = xp + (int64_t)(in_y0 + j) * p.in_stride_h + (int64_t)in_x0 * p.in_stride_w;
            const float* frow = p.f + (int64_t)(tap_y0 + j * step_y) * p.f_stride_h;
            for (int i = 0; i < w; i++)
                acc += Elem<T>::load(row[(int64_t)i * p.in_stride_w]) * (S)frow[(int64_t)(tap_x0 + i * step_x) * p.f_stride_w];
        }
        acc *= (S)p.gain;
        y[(int64_t)n * p.out_stride_n + (int64_t)c * p.out_stride_c + (int64_t)oy * p.out_stride_h +
          (int64_t)ox * p.out_stride_w] = Elem<T>::store(acc);
    }
}

// Workgroups of a (tiles, slots) grid are handed to the eight XCDs round-robin in launch order (x fastest), and
// every XCD has its own L2.  With slots a multiple of 8 this renumbers them so that all tiles of a plane slot run
// on one XCD, close in time: the halo that neighbouring tiles share then comes out of that L2, not over the
// fabric a second time.
__device__ __forceinline__ void plane_major_block(int& tile, int& slot) {
    tile = blockIdx.x;
    slot = blockIdx.y;
    if ((gridDim.y & 7) != 0) return;
    const unsigned id = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = id & 7, j = id >> 3;
    tile = j % gridDim.x;
    slot = (j / gridDim.x) * 8 + xcd;
}

// ---------------------------------------------------------------------------
// Tiled kernel.  FW / FH are the tap counts the code is unrolled for (the actual
// filter may be smaller; missing taps are zero) and must be multiples of UPX / UPY.
template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int FW, int FH, int TILE_W, int TILE_H>
__global__ void __launch_bounds__(256) upfirdn2d_tiled_kernel(sg_upfirdn2d_params p, int tiles_x, int tiles_y) {
    using S = float;
    constexpr int IN_W = ((TILE_W - 1) * DOWNX + FW - 1) / UPX + 1;
    constexpr int IN_H = ((TILE_H - 1) * DOWNY + FH - 1) / UPY + 1;
    constexpr int IN_PITCH = IN_W | 1;  // odd pitch: rows start on different banks
    __shared__ S s_f[FH][FW];
    __shared__ S s_x[IN_H * IN_PITCH];

    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const int tid = threadIdx.x;

    // Taps, flipped the way the correlation below consumes them.
    for (int t = tid; t < FW * FH; t += 256) {
        const int fy = t / FW, fx = t % FW;
        S v = 0;
        if (fx < p.f_w && fy < p.f_h) {
            const int sx = p.flip ? fx : p.f_w - 1 - fx;
            const int sy = p.flip ? fy : p.f_h - 1 - fy;
            v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h];
        }
        s_f[fy][fx] = v;
    }

    // blockIdx.x enumerates (tile_x, tile_y); blockIdx.y strides over the n*c planes.
    int tile, slot;
    plane_major_block(tile, slot);
    const int tile_x = tile % tiles_x;
    const int tile_y = tile / tiles_x;
    const int out_x0 = tile_x * TILE_W;
    const int out_y0 = tile_y * TILE_H;
    const int mid_x0 = out_x0 * DOWNX + UPX - 1 - p.pad_x0;
    const int mid_y0 = out_y0 * DOWNY + UPY - 1 - p.pad_y0;
    const int in_x0 = floor_div(mid_x0, UPX);
    const int in_y0 = floor_div(mid_y0, UPY);
    const int planes = p.in_n * p.in_c;

    for (int plane = slot; plane < planes; plane += gridDim.y) {
        const int n = plane / p.in_c;
        const int c = plane - n * p.in_c;
        const T* xp = x + (int64_t)n * p.in_stride_n + (int64_t)c * p.in_stride_c;
        T* yp = y + (int64_t)n * p.out_stride_n + (int64_t)c * p.out_stride_c;

        __syncthreads();  // previous plane's readers are done with s_x (and s_f is written)
        for (int t = tid; t < IN_H * IN_W; t += 256) {
            const int ry = t / IN_W, rx = t - ry * IN_W;
            const int ix = in_x0 + rx, iy = in_y0 + ry;
            S v = 0;
            if (ix >= 0 && iy >= 0 && ix < p.in_w && iy < p.in_h) v = Elem<T>::load(xp[(int64_t)iy * p.in_stride_h + ix]);
            s_x[ry * IN_PITCH + rx] = v;
        }
        __syncthreads();

        for (int t = tid; t < TILE_W * TILE_H; t += 256) {
            const int ry = t / TILE_W, rx = t - ry * TILE_W;
            const int ox = out_x0 + rx, oy = out_y0 + ry;
            const int mid_x = mid_x0 + rx * DOWNX;
            const int mid_y = mid_y0 + ry * DOWNY;
            const int ix = floor_div(mid_x, UPX);
            const int iy = floor_div(mid_y, UPY);
            const int tap_x = (ix + 1) * UPX - mid_x - 1;
            const int tap_y = (iy + 1) * UPY - mid_y - 1;
            const S* sx = s_x + (iy - in_y0) * IN_PITCH + (ix - in_x0);
            S acc = 0;
#pragma unroll
            for (int j = 0; j < FH / UPY; j++)
#pragma unroll
                for (int i = 0; i < FW / UPX; i++) acc += sx[j * IN_PITCH + i] * s_f[tap_y + j * UPY][tap_x + i * UPX];
            if (ox < p.out_w && oy < p.out_h) yp[(int64_t)oy * p.out_stride_h + ox] = Elem<T>::store(acc * p.gain);
        }
    }
}

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int FW, int FH, int TILE_W, int TILE_H>
int launch_tiled(const sg_upfirdn2d_params& p, hipStream_t stream) {
    const int tiles_x = (p.out_w + TILE_W - 1) / TILE_W;
    const int tiles_y = (p.out_h + TILE_H - 1) / TILE_H;
    const int64_t planes = (int64_t)p.in_n * p.in_c;
    const int64_t tiles = (int64_t)tiles_x * tiles_y;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "output has too many tiles");
    // Enough workgroups for every CU (256 CUs x 8), the remaining planes are looped over.
    int64_t gy = planes;
    const int64_t want = 4096;
    if (tiles * gy > want) gy = (want + tiles - 1) / tiles;
    if (gy > 65528) gy = 65528;
    if (gy < 1) gy = 1;
    if (gy > 8) gy = (gy + 7) / 8 * 8;   // whole groups of eight plane slots: see plane_major_block
    hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, FW, FH, TILE_W, TILE_H>),
                       dim3((unsigned)tiles, (unsigned)gy), dim3(256), 0, stream, p, tiles_x, tiles_y);
    return check_launch("sg_upfirdn2d (tiled)");
}


// ---------------------------------------------------------------------------
// Fast path for the most frequent instance: up = down = 1, filter up to 4x4 (the blur after
// every up-sampling convolution and in front of every strided one).  One workgroup produces a
// 64 x 32 output tile: the 67 x 35 input footprint is fetched with aligned 16-byte packets and
// kept in LDS as float; every lane then produces 8 consecutive outputs of one row from four
// 12-wide LDS row segments (three 16-byte reads per row) and writes them with one 16-byte store.
template <class T, int SHIFT>   // SHIFT = (first tap column in the LDS image) mod 4
__global__ void __launch_bounds__(256) upfirdn2d_blur4_kernel(sg_upfirdn2d_params p, int tiles_x, int tiles_y, sg_epilogue ep,
                                                              bool has_ep, int vec_in) {
    // LDS image of a tile: rows of PITCH floats whose column 0 is the aligned 16-byte packet of the input row that
    // holds the tile's first tap (input column in_x0 sits at column `delta`, the same for every tile since tiles are
    // 64 wide).  Packets are therefore written whole, as float4 stores (single-float stores at a stride of 8 land
    // on 4 of the 32 banks), and a lane reads its 11-float window from 4 aligned float4 at the compile-time offset
    // SHIFT = delta mod 4.
    constexpr int TW = 64, TH = 32, FW = 4, FH = 4;
    constexpr int IN_W = TW + FW - 1, IN_H = TH + FH - 1, PITCH = 84;
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) float s_x[IN_H * PITCH];
    __shared__ float s_f[FH * FW];
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const int tid = threadIdx.x;
    if (tid < FH * FW) {
        const int fy = tid / FW, fx = tid % FW;
        float v = 0;
        if (fx < p.f_w && fy < p.f_h) {
            const int sx = p.flip ? fx : p.f_w - 1 - fx, sy = p.flip ? fy : p.f_h - 1 - fy;
            v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h];
        }
        s_f[tid] = v * p.gain;
    }
    int tile, slot;
    plane_major_block(tile, slot);
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int out_x0 = tile_x * TW, out_y0 = tile_y * TH;
    const int in_x0 = out_x0 - p.pad_x0, in_y0 = out_y0 - p.pad_y0;
    // aligned packets that cover [in_x0, in_x0 + IN_W); without vector loads the image starts at in_x0 itself
    const int vx0 = (in_x0 >= 0) ? in_x0 / VEC : -((-in_x0 + VEC - 1) / VEC);
    const int delta = vec_in ? in_x0 - vx0 * VEC : 0;
    const int nv = (delta + IN_W - 1) / VEC + 1;
    const bool vec_out = (p.out_stride_h % VEC == 0) && (p.out_stride_c % VEC == 0) && (p.out_stride_n % VEC == 0) &&
                         ((reinterpret_cast<uintptr_t>(y) & 15u) == 0);
    const int planes = p.in_n * p.in_c;
    const int lx = (tid & 7) * 8, ly = tid >> 3;   // this lane's 8 outputs: row ly, columns lx .. lx+7
    const int win = lx + (delta & ~3);             // first float4 of the lane's window

    // Packet path: the packets of a plane's tile are loaded into registers one plane ahead (during the filter
    // arithmetic and the stores of the plane before it) and converted into the LDS image when their turn comes.
    constexpr int MAX_NV = (VEC - 1 + IN_W - 1) / VEC + 1;
    constexpr int MAX_IT = (IN_H * MAX_NV + 255) / 256;
    uint4 ahead[MAX_IT];
    auto prefetch = [&](int plane) {
        const int n = plane / p.in_c, c = plane - n * p.in_c;
        const T* xp = x + (int64_t)n * p.in_stride_n + (int64_t)c * p.in_stride_c;
#pragma unroll
        for (int k = 0; k < MAX_IT; k++) {
            const int it = tid + k * 256;
            const int ry = it / nv, v = it - ry * nv;
            const int iy = in_y0 + ry, px0 = (vx0 + v) * VEC;
            ahead[k] = make_uint4(0, 0, 0, 0);
            if (it < IN_H * nv && iy >= 0 && iy < p.in_h && px0 >= 0 && px0 < p.in_w)
                ahead[k] = *reinterpret_cast<const uint4*>(xp + (int64_t)iy * p.in_stride_h + px0);
        }
    };
    if (vec_in && slot < planes) prefetch(slot);

    for (int plane = slot; plane < planes; plane += gridDim.y) {
        const int n = plane / p.in_c, c = plane - n * p.in_c;
        const T* xp = x + (int64_t)n * p.in_stride_n + (int64_t)c * p.in_stride_c;
        T* yp = y + (int64_t)n * p.out_stride_n + (int64_t)c * p.out_stride_c;
        __syncthreads();
        if (vec_in) {
#pragma unroll
            for (int k = 0; k < MAX_IT; k++) {
                const int it = tid + k * 256;
                if (it >= IN_H * nv) continue;
                const int ry = it / nv, v = it - ry * nv;
                const Pack<T> pk = __builtin_bit_cast(Pack<T>, ahead[k]);   // (an all-zero packet is zeros in every type)
                float f[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j++) f[j] = (float)Elem<T>::load(pk.v[j]);
                float4* dst = reinterpret_cast<float4*>(s_x + ry * PITCH + v * VEC);
#pragma unroll
                for (int q = 0; q < VEC / 4; q++) dst[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
            }
        } else {
            for (int it = tid; it < IN_H * IN_W; it += 256) {
                const int ry = it / IN_W, rx = it - ry * IN_W;
                const int ix = in_x0 + rx, iy = in_y0 + ry;
                float v = 0;
                if (ix >= 0 && iy >= 0 && ix < p.in_w && iy < p.in_h) v = (float)Elem<T>::load(xp[(int64_t)iy * p.in_stride_h + ix]);
                s_x[ry * PITCH + rx] = v;
            }
        }
        __syncthreads();
        if (vec_in && plane + (int)gridDim.y < planes) prefetch(plane + gridDim.y);

        const int oy = out_y0 + ly, ox = out_x0 + lx;
        // the lane's 8 noise values, requested before the filter arithmetic that hides their latency
        float nzv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) nzv[j] = 0.f;
        if (has_ep && ep.noise && oy < p.out_h && ox < p.out_w) {
            const float* nz = ep.noise + (int64_t)n * ep.noise_stride_n + (int64_t)oy * p.out_w + ox;
            if (ox + 8 <= p.out_w && (reinterpret_cast<uintptr_t>(nz) & 15u) == 0) {
                const float4 a = reinterpret_cast<const float4*>(nz)[0], b = reinterpret_cast<const float4*>(nz)[1];
                nzv[0] = a.x; nzv[1] = a.y; nzv[2] = a.z; nzv[3] = a.w; nzv[4] = b.x; nzv[5] = b.y; nzv[6] = b.z; nzv[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (ox + j < p.out_w) nzv[j] = nz[j];
            }
        }

        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll
        for (int fy = 0; fy < FH; fy++) {
            const float4* row = reinterpret_cast<const float4*>(s_x + (ly + fy) * PITCH + win);
            const float4 r0 = row[0], r1 = row[1], r2 = row[2];
            float in[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, 0.f, 0.f, 0.f, 0.f};
            if constexpr (SHIFT > 1) {   // the window's last floats lie in a fourth packet
                const float4 r3 = row[3];
                in[12] = r3.x; in[13] = r3.y; in[14] = r3.z; in[15] = r3.w;
            }
#pragma unroll
            for (int fx = 0; fx < FW; fx++) {
                const float tap = s_f[fy * FW + fx];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += in[j + fx + SHIFT] * tap;
            }
        }
        if (has_ep && oy < p.out_h && ox < p.out_w) {
            const float sc = ep.scale ? ep.scale[(int64_t)n * p.in_c + c] : 1.f;
            const float bias = ep.bias ? (float)Elem<T>::load(((const T*)ep.bias)[c]) : 0.f;
            const float ng = (ep.noise && ep.noise_gain) ? *ep.noise_gain : 1.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float v = acc[j] * sc + bias;
                if (ep.noise) v += nzv[j] * ng;
                if (ep.act == 3) v = (v > 0) ? v : v * ep.alpha;
                v *= ep.gain;
                if (ep.clamp >= 0) v = fminf(fmaxf(v, -ep.clamp), ep.clamp);
                acc[j] = v;
            }
        }
        if (oy < p.out_h && ox < p.out_w) {
            T* dst = yp + (int64_t)oy * p.out_stride_h + ox;
            if (sizeof(T) == 2 && vec_out && ox + 8 <= p.out_w) {
                Pack<T> out;
#pragma unroll
                for (int j = 0; j < 8; j++) out.v[j] = Elem<T>::store(acc[j]);
                store_pack(dst, out);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (ox + j < p.out_w) dst[j] = Elem<T>::store(acc[j]);
            }
        }
    }
}

// The same filter on small planes (outputs of 4 x 4 up to 32 x 32: the float32 resolutions of the networks), where
// the 64 x 32 tile of the kernel above is mostly empty - a 16 x 16 plane uses an eighth of its lanes, a 4 x 4 plane one
// in 128.  A workgroup takes as many whole planes as give its 256 lanes four outputs of a row each; the planes' input
// (with the padding as zeros) is staged in LDS as float.
constexpr int kSmallBlurFloats = 6144;       // LDS image of a workgroup's planes
__host__ __device__ inline int small_blur_lanes(int out_w, int out_h) { return out_h * ((out_w + 3) / 4); }   // per plane
inline bool small_blur_eligible(const sg_upfirdn2d_params& p) {
    if (p.out_w < 4 || p.out_w > 32 || p.out_h < 4 || p.out_h > 32) return false;
    const int per_wg = 256 / small_blur_lanes(p.out_w, p.out_h);
    return per_wg >= 1 && (int64_t)per_wg * (p.out_h + 3) * (p.out_w + 3) <= kSmallBlurFloats;
}

template <class T>
__global__ void __launch_bounds__(256) blur4_small_kernel(sg_upfirdn2d_params p, sg_epilogue ep, bool has_ep) {
    __shared__ float s_x[kSmallBlurFloats];
    __shared__ float s_f[16];
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const int tid = threadIdx.x;
    if (tid < 16) {
        const int fy = tid / 4, fx = tid % 4;
        float v = 0;
        if (fx < p.f_w && fy < p.f_h) {
            const int sx = p.flip ? fx : p.f_w - 1 - fx, sy = p.flip ? fy : p.f_h - 1 - fy;
            v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h];
        }
        s_f[tid] = v * p.gain;
    }
    const int iw = p.out_w + 3, ih = p.out_h + 3, img = iw * ih;          // a plane's image: input columns -pad_x0 ..
    const int groups = (p.out_w + 3) / 4, lanes = p.out_h * groups, per_wg = 256 / lanes;
    const int planes = p.in_n * p.in_c, plane0 = blockIdx.x * per_wg, here = min(per_wg, planes - plane0);
    for (int e = tid; e < here * img; e += 256) {
        const int pl = e / img, r = e - pl * img, ry = r / iw, rx = r - ry * iw;
        const int plane = plane0 + pl, n = plane / p.in_c, c = plane - n * p.in_c;
        const int ix = rx - p.pad_x0, iy = ry - p.pad_y0;
        float v = 0.f;
        if (ix >= 0 && ix < p.in_w && iy >= 0 && iy < p.in_h)
            v = (float)Elem<T>::load(x[(int64_t)n * p.in_stride_n + (int64_t)c * p.in_stride_c + (int64_t)iy * p.in_stride_h + ix]);
        s_x[e] = v;
    }
    __syncthreads();
    const int pl = tid / lanes, l = tid - pl * lanes;
    if (pl >= here) return;
    const int oy = l / groups, ox = (l - oy * groups) * 4;
    const int plane = plane0 + pl, n = plane / p.in_c, c = plane - n * p.in_c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fy = 0; fy < 4; fy++) {
        const float* row = s_x + pl * img + (oy + fy) * iw + ox;
        float in[7];
#pragma unroll
        for (int j = 0; j < 7; j++) in[j] = (ox + j < iw) ? row[j] : 0.f;
#pragma unroll
        for (int fx = 0; fx < 4; fx++) {
            const float tap = s_f[fy * 4 + fx];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += in[j + fx] * tap;
        }
    }
    if (has_ep) {
        const float sc = ep.scale ? ep.scale[(int64_t)n * p.in_c + c] : 1.f;
        const float bias = ep.bias ? (float)Elem<T>::load(((const T*)ep.bias)[c]) : 0.f;
        const float ng = (ep.noise && ep.noise_gain) ? *ep.noise_gain : 1.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = acc[j] * sc + bias;
            if (ep.noise && ox + j < p.out_w) v += ep.noise[(int64_t)n * ep.noise_stride_n + (int64_t)oy * p.out_w + ox + j] * ng;
            if (ep.act == 3) v = (v > 0) ? v : v * ep.alpha;
            v *= ep.gain;
            if (ep.clamp >= 0) v = fminf(fmaxf(v, -ep.clamp), ep.clamp);
            acc[j] = v;
        }
    }
    T* dst = y + (int64_t)n * p.out_stride_n + (int64_t)c * p.out_stride_c + (int64_t)oy * p.out_stride_h + ox;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (ox + j < p.out_w) dst[j] = Elem<T>::store(acc[j]);
}

template <class T>
int launch_blur4(const sg_upfirdn2d_params& p, hipStream_t stream, const sg_epilogue* ep = nullptr) {
    if (blur_stream_eligible(p, ep)) return blur_stream_launch(p, ep, stream);
    if (small_blur_eligible(p)) {
        const int per_wg = 256 / small_blur_lanes(p.out_w, p.out_h);
        const int64_t planes = (int64_t)p.in_n * p.in_c;
        sg_epilogue e = {};
        if (ep) e = *ep;
        hipLaunchKernelGGL((blur4_small_kernel<T>), dim3((unsigned)((planes + per_wg - 1) / per_wg)), dim3(256), 0, stream, p, e, ep != nullptr);
        return check_launch("sg_upfirdn2d (blur, small planes)");
    }
    const int tiles_x = (p.out_w + 63) / 64, tiles_y = (p.out_h + 31) / 32;
    const int64_t planes = (int64_t)p.in_n * p.in_c, tiles = (int64_t)tiles_x * tiles_y;
    SG_REQUIRE(tiles <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "output has too many tiles");
    int64_t gy = planes;
    if (tiles * gy > 8192) gy = (8192 + tiles - 1) / tiles;
    if (gy > 65528) gy = 65528;
    if (gy < 1) gy = 1;
    if (gy > 8) gy = (gy + 7) / 8 * 8;   // whole groups of eight plane slots: see plane_major_block
    sg_epilogue e = {};
    if (ep) e = *ep;
    // Rows of whole aligned packets are loaded as such; the column of the first tap inside its packet is then the
    // same for every tile (tiles are 64 outputs wide): -pad_x0 modulo the packet length.
    constexpr int VEC = 16 / (int)sizeof(T);
    const int vec_in = (p.in_stride_h % VEC == 0) && (p.in_stride_c % VEC == 0) && (p.in_stride_n % VEC == 0) &&
                       (p.in_w % VEC == 0) && aligned16(p.x);
    const int delta = vec_in ? ((-p.pad_x0) % VEC + VEC) % VEC : 0;
    const dim3 grid((unsigned)tiles, (unsigned)gy);
    switch (delta & 3) {
        case 0: hipLaunchKernelGGL((upfirdn2d_blur4_kernel<T, 0>), grid, dim3(256), 0, stream, p, tiles_x, tiles_y, e, ep != nullptr, vec_in); break;
        case 1: hipLaunchKernelGGL((upfirdn2d_blur4_kernel<T, 1>), grid, dim3(256), 0, stream, p, tiles_x, tiles_y, e, ep != nullptr, vec_in); break;
        case 2: hipLaunchKernelGGL((upfirdn2d_blur4_kernel<T, 2>), grid, dim3(256), 0, stream, p, tiles_x, tiles_y, e, ep != nullptr, vec_in); break;
        default: hipLaunchKernelGGL((upfirdn2d_blur4_kernel<T, 3>), grid, dim3(256), 0, stream, p, tiles_x, tiles_y, e, ep != nullptr, vec_in); break;
    }
    return check_launch("sg_upfirdn2d (blur)");
}

template <class T>
int launch_generic(const sg_upfirdn2d_params& p, hipStream_t stream) {
    const int64_t total = (int64_t)p.out_w * p.out_h * p.in_c * p.in_n;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL((upfirdn2d_generic_kernel<T>), dim3((unsigned)grid), dim3(256), 0, stream, p, total);
    return check_launch("sg_upfirdn2d (generic)");
}

template <class T>
int launch(const sg_upfirdn2d_params& p, hipStream_t stream) {
    const bool rows_contiguous = p.in_stride_w == 1 && p.out_stride_w == 1;
    if (rows_contiguous && sizeof(T) <= 4 && p.f_w <= 4 && p.f_h <= 4) {
        const int ux = p.up_x, uy = p.up_y, dx = p.down_x, dy = p.down_y;
        if (ux == 1 && uy == 1 && dx == 1 && dy == 1) return launch_blur4<T>(p, stream);
        if (ux == 2 && uy == 2 && dx == 1 && dy == 1) {
            if (up2_stream_eligible(p)) return up2_stream_launch(p, nullptr, stream);
            return launch_tiled<T, 2, 2, 1, 1, 4, 4, 64, 16>(p, stream);
        }
        if (ux == 1 && uy == 1 && dx == 2 && dy == 2) {
            if (down2_stream_eligible(p)) return down2_stream_launch(p, stream);
            return launch_tiled<T, 1, 1, 2, 2, 4, 4, 64, 8>(p, stream);
        }
    }
    // One-dimensional passes of a separable filter of up to 12 taps at a factor of two (the wavelet low-pass of the
    // augmentation pipeline's geometric resampler, reference augment.py:352 and :370, and their adjoints).
    if (rows_contiguous && sizeof(T) <= 4 && p.f_w <= 12 && p.f_h == 1 && p.up_y == 1 && p.down_y == 1) {
        if (p.up_x == 2 && p.down_x == 1) return launch_tiled<T, 2, 1, 1, 1, 12, 1, 64, 16>(p, stream);
        if (p.up_x == 1 && p.down_x == 2) return launch_tiled<T, 1, 1, 2, 1, 12, 1, 64, 16>(p, stream);
    }
    if (rows_contiguous && sizeof(T) <= 4 && p.f_h <= 12 && p.f_w == 1 && p.up_x == 1 && p.down_x == 1) {
        if (p.up_y == 2 && p.down_y == 1) return launch_tiled<T, 1, 2, 1, 1, 1, 12, 64, 16>(p, stream);
        if (p.up_y == 1 && p.down_y == 2) return launch_tiled<T, 1, 1, 1, 2, 1, 12, 64, 8>(p, stream);
    }
    return launch_generic<T>(p, stream);
}

template <>
int launch<double>(const sg_upfirdn2d_params& p, hipStream_t stream) {
    return launch_generic<double>(p, stream);
}

}  // namespace
}  // namespace sg

extern "C" int sg_upfirdn2d_out_size(int in_size, int up, int pad0, int pad1, int f_size, int down) {
    if (up < 1 || down < 1 || f_size < 1) return -1;
    const int64_t v = (int64_t)in_size * up + pad0 + pad1 - f_size + down;
    if (v < 0) return 0;  // C division truncates toward zero; a negative numerator means "no output"
    return (int)(v / down);
}

extern "C" int sg_upfirdn2d(const sg_upfirdn2d_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->up_x >= 1 && p->up_y >= 1, SG_ERR_INVALID_ARGUMENT, "upsampling factor must be at least 1");
    SG_REQUIRE(p->down_x >= 1 && p->down_y >= 1, SG_ERR_INVALID_ARGUMENT, "downsampling factor must be at least 1");
    SG_REQUIRE(p->f_w >= 1 && p->f_h >= 1, SG_ERR_INVALID_ARGUMENT, "f must be at least 1x1");
    SG_REQUIRE(p->in_w >= 0 && p->in_h >= 0 && p->in_c >= 0 && p->in_n >= 0, SG_ERR_INVALID_ARGUMENT,
               "negative input size");
    SG_REQUIRE(p->out_w >= 1 && p->out_h >= 1, SG_ERR_INVALID_ARGUMENT, "output must be at least 1x1");
    SG_REQUIRE((int64_t)p->in_w * p->in_h * p->in_c * p->in_n <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "x is too large");
    SG_REQUIRE((int64_t)p->out_w * p->out_h * p->in_c * p->in_n <= INT32_MAX, SG_ERR_INVALID_ARGUMENT,
               "output is too large");
    // The caller's output must not be larger than the filter footprint allows (else reads run past x's padding
    // logic into other planes); out_w * down <= in_w * up + pad0 + pad1 - f_w + down cannot be checked without
    // pad1, so bound it by the widest legal choice pad1 -> the kernels clamp every read to [0, in) anyway.
    if (p->in_c == 0 || p->in_n == 0) return SG_OK;
    SG_REQUIRE(p->x && p->f && p->y, SG_ERR_NULL_POINTER, "x, f and y must not be NULL");
    SG_DISPATCH_DTYPE(p->dtype, true, return launch<T>(*p, (hipStream_t)stream));
    return SG_OK;
}

extern "C" const char* sg_upfirdn2d_kernel(const sg_upfirdn2d_params* p, const sg_epilogue* ep) {
    // mirrors launch() / launch_blur4() above
    using namespace sg;
    if (p == nullptr) return "";
    if (p->dtype == SG_F64) return "upfirdn2d_generic_kernel";
    const bool rows_contiguous = p->in_stride_w == 1 && p->out_stride_w == 1;
    const int ux = p->up_x, uy = p->up_y, dx = p->down_x, dy = p->down_y;
    if (rows_contiguous && p->f_w <= 4 && p->f_h <= 4) {
        if (ux == 1 && uy == 1 && dx == 1 && dy == 1)
            return blur_stream_eligible(*p, ep) ? "blur4_stream_kernel" : small_blur_eligible(*p) ? "blur4_small_kernel" : "upfirdn2d_blur4_kernel";
        if (ux == 2 && uy == 2 && dx == 1 && dy == 1) return up2_stream_eligible(*p) ? "up2_stream_kernel" : "upfirdn2d_tiled_kernel";
        if (ux == 1 && uy == 1 && dx == 2 && dy == 2) return down2_stream_eligible(*p) ? "down2_stream_kernel" : "upfirdn2d_tiled_kernel";
    }
    if (ep != nullptr) return "";
    if (rows_contiguous && p->f_w <= 12 && p->f_h == 1 && uy == 1 && dy == 1 && ((ux == 2 && dx == 1) || (ux == 1 && dx == 2)))
        return "upfirdn2d_tiled_kernel";
    if (rows_contiguous && p->f_h <= 12 && p->f_w == 1 && ux == 1 && dx == 1 && ((uy == 2 && dy == 1) || (uy == 1 && dy == 2)))
        return "upfirdn2d_tiled_kernel";
    return "upfirdn2d_generic_kernel";
}

extern "C" int sg_upfirdn2d_add_supported(const sg_upfirdn2d_params* p) {
    return (p != nullptr && p->in_c > 0 && p->in_n > 0 && p->out_w >= 1 && p->out_h >= 1 && sg::up2_stream_eligible(*p)) ? 1 : 0;
}

extern "C" int sg_upfirdn2d_add(const sg_upfirdn2d_params* p, const void* addend, sg_stream_t stream) {
    using namespace sg;
    if (addend == nullptr) return sg_upfirdn2d(p, stream);
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->in_w >= 0 && p->in_h >= 0 && p->in_c >= 0 && p->in_n >= 0, SG_ERR_INVALID_ARGUMENT, "negative input size");
    SG_REQUIRE(p->out_w >= 1 && p->out_h >= 1, SG_ERR_INVALID_ARGUMENT, "output must be at least 1x1");
    SG_REQUIRE((int64_t)p->in_w * p->in_h * p->in_c * p->in_n <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "x is too large");
    SG_REQUIRE((int64_t)p->out_w * p->out_h * p->in_c * p->in_n <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "output is too large");
    if (p->in_c == 0 || p->in_n == 0) return SG_OK;
    SG_REQUIRE(p->x && p->f && p->y, SG_ERR_NULL_POINTER, "x, f and y must not be NULL");
    SG_REQUIRE(sg_upfirdn2d_add_supported(p) && aligned16(addend), SG_ERR_UNSUPPORTED,
               "upfirdn2d with addend: only the up-sampling stream geometry (16-bit, up 2, down 1, filter up to 4x4, rows of whole "
               "aligned packets, output at least 64 wide with an even number of rows)");
    return up2_stream_launch(*p, addend, (hipStream_t)stream);
}

extern "C" int sg_upfirdn2d_epilogue(const sg_upfirdn2d_params* p, const sg_epilogue* ep, sg_stream_t stream) {
    using namespace sg;
    if (ep == nullptr) return sg_upfirdn2d(p, stream);
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->up_x == 1 && p->up_y == 1 && p->down_x == 1 && p->down_y == 1 && p->f_w >= 1 && p->f_h >= 1 &&
                   p->f_w <= 4 && p->f_h <= 4 && p->in_stride_w == 1 && p->out_stride_w == 1,
               SG_ERR_UNSUPPORTED, "upfirdn2d with epilogue: only up = down = 1, filters up to 4x4, contiguous rows");
    SG_REQUIRE(p->out_w >= 1 && p->out_h >= 1, SG_ERR_INVALID_ARGUMENT, "output must be at least 1x1");
    SG_REQUIRE(ep->act == 1 || ep->act == 3, SG_ERR_UNSUPPORTED, "epilogue activation must be linear (1) or lrelu (3)");
    SG_REQUIRE(!ep->noise || p->out_stride_h == p->out_w, SG_ERR_UNSUPPORTED, "epilogue noise needs a dense output");
    if (p->in_c == 0 || p->in_n == 0) return SG_OK;
    SG_REQUIRE(p->x && p->f && p->y, SG_ERR_NULL_POINTER, "x, f and y must not be NULL");
    switch (p->dtype) {
        case SG_F32: return launch_blur4<float>(*p, (hipStream_t)stream, ep);
        case SG_F16: return launch_blur4<__half>(*p, (hipStream_t)stream, ep);
        case SG_BF16: return launch_blur4<__hip_bfloat16>(*p, (hipStream_t)stream, ep);
    }
    return fail(SG_ERR_UNSUPPORTED, "upfirdn2d with epilogue: dtype %d is not supported", p->dtype);
}
