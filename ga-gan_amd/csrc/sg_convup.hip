// Transposed 3x3 convolution with stride 2 of 16-bit tensors (the up-sampling layers of the
// generator and the data gradient of the discriminator's down-sampling layers) as ONE unit-stride
// contraction: output pixel (2a + py, 2b + px) only depends on the 2 x 2 input window
// {a-1, a} x {b-1, b}, through 4 / 2 / 2 / 1 of the 9 weight taps for the phases (py, px) =
// (0,0) / (0,1) / (1,0) / (1,1).  GEMM view: M = input-grid pixels (a, b), N = phase x output
// channel, K = input channels x window taps - the 9 non-zero (phase, tap) combinations only.
//
// Against the general kernel's polyphase form this stages the input tile once for all four phases
// (not once per phase), and a lane ends up with 4 consecutive b of both px phases of a channel,
// i.e. 8 consecutive output pixels: one 16-byte store instead of 8 strided 2-byte stores.
//
// Workgroup: 4 waves, tile of 8 x 32 input-grid pixels (16 x 64 outputs) x 32 output channels x 4
// phases; K in chunks of 32 channels with the next chunk's packets (input and weights) loaded into
// registers during the multiply phase; swizzled 64-byte LDS rows as in the 3x3 kernel.  Input
// modulation (in_scale) is applied while staging; there is no output epilogue (the layer's epilogue
// runs in the FIR that follows) - sg_conv2d keeps the general kernel for calls that ask for one.
#include "sg_common.h"
#include "sg_conv3x3.h"

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TW = 32, TH = 8;            // input-grid tile
constexpr int KCH = 32;                   // input channels per chunk
constexpr int BCO = 32;                   // output channels per workgroup (x 4 phases = 128 GEMM columns)
constexpr int kRow = 64;
constexpr int LH = TH + 1;                // footprint: rows a0-1 .. a0+7, columns b0-1 .. b0+31
constexpr int NV = 5;                     // packets b0-8 .. b0+31
constexpr int LWP = NV * 8;
constexpr int PX_COL = 7;                 // column of input b0-1 inside the staged row
constexpr int kInBytes = LH * LWP * kRow;         // 23040
constexpr int kCombos = 9;
constexpr int kWBytes = kCombos * BCO * kRow;     // 18432
constexpr int ITEMS = LH * NV;                    // 45 (row, packet) items per channel pair
constexpr int IN_STEPS = (ITEMS + 15) / 16;       // 3

// (phase, tap) combinations: phase = 2 py + px; (ty, tx) index the phase's own tap grid in the packed weights;
// (dy, dx) in {-1, 0} is the input offset of the tap.
struct Combo { int phase, ty, tx, dy, dx; };
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

__device__ __forceinline__ int swz(int row, int byte) {
    return row * kRow + ((((byte >> 4) ^ (row >> 1)) & 3) << 4) + (byte & 15);
}
template <class T> __device__ __forceinline__ float lo16(unsigned w) { return Elem<T>::load(__builtin_bit_cast(T, (unsigned short)(w & 0xffffu))); }
template <class T> __device__ __forceinline__ unsigned pack16(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(a)) |
           ((unsigned)__builtin_bit_cast(unsigned short, Elem<T>::store(b)) << 16);
}

struct ConvUpArgs {
    const void* x; const void* w; void* y;
    const float* in_scale;
    int n, ci, co, h, wid, oh, ow;
    int w_offset0, w_offset1, w_offset2, w_offset3;   // start of each phase's taps in the packed weights (elements)
    int tiles_a, tiles_b, units;
};

// WAVES = 8 (layers with more than 32 output channels): 64 output channels per workgroup - waves 4..7 multiply the
// same staged pixels with the second 32 channels; one such workgroup per CU, two waves per SIMD.
template <class T, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) conv_up2_kernel(ConvUpArgs g) {
    constexpr int BCOT = BCO * (WAVES / 4);                                    // output channels per workgroup
    constexpr int SLOTS = 4 * WAVES, STEPS = (ITEMS + SLOTS - 1) / SLOTS;      // staging: item slots, items per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_in = smem;
    unsigned char* s_w = smem + kInBytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3, half = tid >> 8;   // tile rows 2 wave, 2 wave + 1; channels 32 half ..
    const int lrow = lane & 15, lgrp = lane >> 4;

    // XCD k (workgroups k, k + 8, ...) takes the k-th contiguous eighth of the units: neighbouring tiles share an L2
    const int per_xcd = (g.units + 7) >> 3;
    int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= g.units) return;
    const int tb = unit % g.tiles_b; unit /= g.tiles_b;
    const int ta = unit % g.tiles_a; unit /= g.tiles_a;
    const int n = unit % g.n;
    const int m0 = (unit / g.n) * BCOT;
    const int a0 = ta * TH, b0 = tb * TW;
    const int plane = g.h * g.wid;
    const T* __restrict__ x = (const T*)g.x + (unsigned)(n * g.ci * plane);
    const T* __restrict__ w = (const T*)g.w;

    const int cg = tid & 15, slot0 = tid >> 4;
    uint4 va[STEPS], vb[STEPS];
    float s0 = 1.f, s1 = 1.f;

    auto fetch = [&](int c0) {
        const int c = c0 + 2 * cg;
        const bool c_ok = c < g.ci, two = c + 1 < g.ci;
        const T* src0 = x + (unsigned)(c * plane);
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const int item = slot0 + SLOTS * q;
            const int ry = item / NV, v = item - ry * NV;
            const int iy = a0 - 1 + ry, px0 = b0 - 8 + v * 8;
            va[q] = make_uint4(0, 0, 0, 0);
            vb[q] = make_uint4(0, 0, 0, 0);
            if (c_ok && item < ITEMS && iy >= 0 && iy < g.h && px0 >= 0 && px0 < g.wid) {
                const T* src = src0 + iy * g.wid + px0;
                va[q] = *reinterpret_cast<const uint4*>(src);
                if (two) vb[q] = *reinterpret_cast<const uint4*>(src + plane);
            }
        }
        if (g.in_scale) {
            s0 = c_ok ? g.in_scale[n * g.ci + c] : 0.f;
            s1 = two ? g.in_scale[n * g.ci + c + 1] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const int item = slot0 + SLOTS * q;
            if (item >= ITEMS) continue;
            const unsigned wa[4] = {va[q].x, va[q].y, va[q].z, va[q].w};
            const unsigned wb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
            const int r0 = item * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned ea = wa[j >> 1] >> (16 * (j & 1)), eb = wb[j >> 1] >> (16 * (j & 1));
                unsigned out;
                if (g.in_scale) out = pack16<T>(lo16<T>(ea) * s0, lo16<T>(eb) * s1);
                else out = (ea & 0xffffu) | (eb << 16);
                *reinterpret_cast<unsigned*>(s_in + swz(r0 + j, cg * 4)) = out;
            }
        }
    };
    // weights of a chunk: cache-resident (every workgroup reads them), fetched right before use.  Waves 0-1 take the
    // even combinations, waves 2-3 the odd ones; a lane moves one 16-byte packet of one output channel per step.
    auto stage_weights = [&](int c0) {
        constexpr Combo kc[kCombos + 1] = {{0, 0, 0, -1, -1}, {0, 0, 1, -1, 0}, {0, 1, 0, 0, -1}, {0, 1, 1, 0, 0}, {1, 0, 0, -1, 0},
                                           {1, 1, 0, 0, 0}, {2, 0, 0, 0, -1}, {2, 0, 1, 0, 0}, {3, 0, 0, 0, 0}, {3, 0, 0, 0, 0}};
        constexpr int nkx[4] = {2, 1, 2, 1};
        const bool odd = (tid & 255) >= 128;
        const int o_local = ((tid >> 2) & 31) + 32 * half, slot = tid & 3;
        const int o = m0 + o_local, cw = c0 + slot * 8;
        const bool ok = o < g.co && cw < g.ci;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const int combo = 2 * q + (odd ? 1 : 0);
            const int phase = odd ? kc[2 * q + 1].phase : kc[2 * q].phase;
            const int tap = odd ? kc[2 * q + 1].ty * nkx[kc[2 * q + 1].phase] + kc[2 * q + 1].tx
                                : kc[2 * q].ty * nkx[kc[2 * q].phase] + kc[2 * q].tx;
            const int w_off = phase == 0 ? g.w_offset0 : phase == 1 ? g.w_offset1 : phase == 2 ? g.w_offset2 : g.w_offset3;
            if (combo >= kCombos) continue;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(w + (unsigned)(w_off + (tap * g.co + o) * g.ci + cw));
            *reinterpret_cast<uint4*>(s_w + swz(combo * BCOT + o_local, slot * 16)) = v;
        }
    };

    // n-tile j of wave w: tile row 2w + (j >> 1), columns 16 (j & 1) .. +15; LDS row of input (a + dy, b + dx):
    // (a - a0 + 1 + dy) * LWP + PX_COL + 1 + (b - b0) + dx
    int brow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) brow[j] = (2 * wave + (j >> 1) + 1) * LWP + PX_COL + 1 + (j & 1) * 16 + lrow;
    const int a_lane = lrow * kRow + (((lgrp ^ (lrow >> 1)) & 3) << 4);

    f32x4_t acc[4][2][4];   // [phase][channel tile][pixel tile]: rows = pixels 4 lgrp + r, column = channel lrow
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[p][ct][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&]() {
        // window offset outermost: its 4 input fragments serve every (phase, tap) combination that reads it
        constexpr Combo kc[kCombos] = {{0, 0, 0, -1, -1}, {0, 0, 1, -1, 0}, {0, 1, 0, 0, -1}, {0, 1, 1, 0, 0}, {1, 0, 0, -1, 0},
                                       {1, 1, 0, 0, 0}, {2, 0, 0, 0, -1}, {2, 0, 1, 0, 0}, {3, 0, 0, 0, 0}};
#pragma unroll
        for (int o = 0; o < 4; o++) {   // o = 2 (dy + 1) + (dx + 1)
            uint4 fb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = brow[j] + ((o >> 1) - 1) * LWP + (o & 1) - 1;
                fb[j] = *reinterpret_cast<const uint4*>(s_in + row * kRow + (((lgrp ^ (row >> 1)) & 3) << 4));
            }
#pragma unroll
            for (int combo = 0; combo < kCombos; combo++) {
                if (2 * (kc[combo].dy + 1) + kc[combo].dx + 1 != o) continue;
                const int p = kc[combo].phase;
                uint4 fa[2];
#pragma unroll
                for (int ct = 0; ct < 2; ct++) fa[ct] = *reinterpret_cast<const uint4*>(s_w + (combo * BCOT + 32 * half + ct * 16) * kRow + a_lane);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) Mma16<T>::run(fb[j], fa[ct], acc[p][ct][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    fetch(0);
    for (int c0 = 0; c0 < g.ci; c0 += KCH) {
        __syncthreads();
        stage_weights(c0);
        commit();
        __syncthreads();
        if (c0 + KCH < g.ci) fetch(c0 + KCH);   // in flight during the multiply phase
        multiply();
    }

    // ---- store: both px phases of a channel interleave into 8 consecutive output pixels -----------------------
    T* __restrict__ y = (T*)g.y;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = a0 + 2 * wave + (j >> 1), b = b0 + (j & 1) * 16 + 4 * lgrp;
        const int ox = 2 * b;
        if (ox >= g.ow) continue;
#pragma unroll
        for (int py = 0; py < 2; py++) {
            const int oy = 2 * a + py;
            if (oy >= g.oh) continue;
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int ch = m0 + 32 * half + ct * 16 + lrow;
                if (ch >= g.co) continue;
                const f32x4_t e = acc[2 * py][ct][j], o = acc[2 * py + 1][ct][j];
                const uint4 out = make_uint4(pack16<T>(e[0], o[0]), pack16<T>(e[1], o[1]), pack16<T>(e[2], o[2]), pack16<T>(e[3], o[3]));
                T* dst = y + (unsigned)(((n * g.co + ch) * g.oh + oy) * g.ow + ox);
                if (ox + 8 <= g.ow) *reinterpret_cast<uint4*>(dst) = out;
                else {   // ragged last packet of a row
                    const unsigned words[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (ox + i < g.ow) dst[i] = __builtin_bit_cast(T, (unsigned short)(words[i >> 1] >> (16 * (i & 1))));
                }
            }
        }
    }
}

}  // namespace

bool conv_up2_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 4 || p.in_step != 1 || p.out_step != 2) return false;
    // the four phases of a 3x3, stride-2, unpadded transposed convolution, in (py, px) order
    static const int want[4][6] = {{0, 0, -1, -1, 2, 2}, {0, 1, -1, 0, 2, 1}, {1, 0, 0, -1, 1, 2}, {1, 1, 0, 0, 1, 1}};
    for (int i = 0; i < 4; i++) {
        const sg_conv2d_phase& ph = p.phase[i];
        if (ph.oy0 != want[i][0] || ph.ox0 != want[i][1] || ph.dy0 != want[i][2] || ph.dx0 != want[i][3] ||
            ph.nky != want[i][4] || ph.nkx != want[i][5] || ph.w_offset < 0 || ph.w_offset > INT32_MAX / 2)
            return false;
    }
    if (p.out_scale || p.bias || p.noise || p.reduce_out || p.act != 1 || p.gain != 1.f || p.clamp >= 0.f) return false;
    if (p.w_in % 8 != 0 || p.ow % 8 != 0 || p.ci % 8 != 0 || p.w_in < 16 || p.ci < KCH) return false;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.y)) return false;
    return true;
}

int conv_up2_launch(const sg_conv2d_params& p, hipStream_t stream) {
    ConvUpArgs g;
    g.x = p.x; g.w = p.w; g.y = p.y; g.in_scale = p.in_scale;
    g.n = p.n; g.ci = p.ci; g.co = p.co; g.h = p.h; g.wid = p.w_in; g.oh = p.oh; g.ow = p.ow;
    g.w_offset0 = (int)p.phase[0].w_offset; g.w_offset1 = (int)p.phase[1].w_offset;
    g.w_offset2 = (int)p.phase[2].w_offset; g.w_offset3 = (int)p.phase[3].w_offset;
    g.tiles_a = ((p.oh + 1) / 2 + TH - 1) / TH;
    g.tiles_b = ((p.ow + 1) / 2 + TW - 1) / TW;
    const bool wide = p.co > BCO;
    const int bco = wide ? 2 * BCO : BCO;
    const int64_t units = (int64_t)g.tiles_a * g.tiles_b * p.n * ((p.co + bco - 1) / bco);
    SG_REQUIRE(units <= INT32_MAX - 8, SG_ERR_INVALID_ARGUMENT, "conv2d: too many tiles");
    g.units = (int)units;
    const size_t lds = kInBytes + kWBytes * (wide ? 2 : 1);   // 59904 at most: below the 64 KB that need no opt-in
    const dim3 grid((unsigned)((units + 7) / 8 * 8));
    if (p.dtype == SG_BF16) {
        if (wide) hipLaunchKernelGGL((conv_up2_kernel<__hip_bfloat16, 8>), grid, dim3(512), lds, stream, g);
        else hipLaunchKernelGGL((conv_up2_kernel<__hip_bfloat16, 4>), grid, dim3(256), lds, stream, g);
    } else {
        if (wide) hipLaunchKernelGGL((conv_up2_kernel<__half, 8>), grid, dim3(512), lds, stream, g);
        else hipLaunchKernelGGL((conv_up2_kernel<__half, 4>), grid, dim3(256), lds, stream, g);
    }
    return check_launch("sg_conv2d");
}

}  // namespace sg
