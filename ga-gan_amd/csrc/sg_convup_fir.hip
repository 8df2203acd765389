// Up-sampling layer in ONE pass: transposed 3x3 convolution with stride 2, the 4-tap resampling filter in both
// directions and the layer epilogue (demodulation, noise, bias, leaky ReLU, gain, clamp), with no intermediate
// tensor in memory.  What the reference composes from conv_transpose2d + upfirdn2d (torch_utils/ops/
// conv2d_resample.py:125-142), fma (training/networks.py:647-652) and bias_act (training/networks.py:919-921).
//
// The filter is separable, f = fy (x) fx, and a filter along a row commutes with the channel contraction, so
//   * the HORIZONTAL filter is folded into the weights: the caller convolves every weight row with 2 fx
//     (3 taps -> 6 taps of a stride-2 transposed kernel).  Output column 2 b + px then reads input columns
//     b + 1, b, b - 1 through composed taps d = px - 2, px, px + 2 - three taps per column phase, no halo needed
//     beyond one input column on either side;
//   * the VERTICAL direction keeps the polyphase form of the transposed convolution (row 2 a + py of the
//     intermediate reads input rows a and a - 1 for py = 0, row a for py = 1) and the vertical filter runs on the
//     ACCUMULATORS: a wave owns a strip of 16 input columns for all rows of the tile, so the rows
//     u[2a-1 .. 2a+2] that output rows 2a and 2a+1 need are registers of the same lane.
// 18 (row tap, column tap) combinations per input pixel instead of 9 for the bare transposed convolution: twice
// the matrix work (these layers have little of it), in exchange for never writing and re-reading the
// (2H+1) x (2W+1) intermediate, which is 4/3 of all other traffic of the layer.
//
// GEMM view per chunk of 32 input channels: M = pixels of the strip (16 per matrix instruction), N = output channel,
// K = 32 channels; v_mfma_f32_16x16x32.  Workgroup: 8 waves = 4 strips x 2 tiles of 16 output channels; tile of
// 4 x 64 input pixels (8 x 128 outputs) x 32 output channels.  Accumulators: U[r][py][px] for the 6 intermediate row
// pairs r (the first one: py = 1 only) = 88 registers.  Workgroups are persistent and everything in LDS is double
// buffered: the packets of the next chunk - or of the next tile's first chunk -, pixels and weights, are requested
// before the multiply phase and committed to the other buffer after it, one barrier per chunk.  With at most 64 input
// and 32 output channels the two weight buffers hold the whole composed kernel and are filled once per workgroup.
// Waves 4..7 run a tile's filter / epilogue / stores after the barrier, in front of their next multiplies, so that on
// every SIMD one wave multiplies while its partner does vector arithmetic and stores.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int TW = 64, TH = 4;            // input-grid tile
constexpr int KCH = 32;                   // input channels per chunk
constexpr int BCO = 32;                   // output channels per group of four waves
constexpr int kRow = 64;                  // bytes of a pixel (or weight row) in LDS: 32 channels
constexpr int LH = TH + 2;                // staged input rows a0-1 .. a0+TH
constexpr int NV = TW / 8 + 2;            // staged packets per row: columns b0-8 .. b0+TW+7
constexpr int LWP = NV * 8;
constexpr int kInBytes = LH * LWP * kRow; // 30720
constexpr int kCombos = 18;               // (row tap) x (column tap)
constexpr int ITEMS = LH * NV;            // (row, packet) items per channel pair

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
template <class T> struct Pair;
template <> struct Pair<__hip_bfloat16> {
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const bf16x2_t v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    }
};
template <> struct Pair<__half> {
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        const f16x2_t v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
};

// 64-byte LDS rows whose 16-byte slots are XOR-swizzled with the row index: conflict-free 16-byte reads by 16
// consecutive rows x 4 slots (the access pattern of a matrix operand)
__device__ __forceinline__ int swz(int row, int byte) {
    return row * kRow + ((((byte >> 4) ^ (row >> 1)) & 3) << 4) + (byte & 15);
}

struct UpFirArgs {
    const void* x; const void* w; void* y;
    const float* in_scale; const float* fy;
    const float* scale; const float* noise; const float* noise_gain; const void* bias;
    int noise_stride;
    int n, ci, co, h, wid;
    int tiles_a, tiles_b, cogs, units;
    float slope, gain, hi;        // leaky-ReLU slope (1: linear), gain (> 0), clamp bound (+inf: none)
};

constexpr int kThreads = 512;
constexpr int kWBytes = kCombos * BCO * kRow;                      // 36864: the weights of one chunk
constexpr int STEPS = 2;                                           // loads of a step per lane and channel: 15 wave-slots of 64 for 8 waves
constexpr int WPACKETS = kCombos * BCO * 4, WROUNDS = (WPACKETS + kThreads - 1) / kThreads;   // 16-byte weight packets of a chunk
constexpr int kNoise = 2 * TH * 2 * TW;                            // floats of a tile's noise
constexpr size_t kLds = 2 * kInBytes + 2 * kWBytes + 3 * kNoise * sizeof(float);   // 147456

struct Tile { int n, a0, b0, m0; };
struct Pos { int unit, c, idx; bool ok; Tile t; };    // a step: chunk c of the idx-th tile of this workgroup
struct Stage {                                        // what a step's loads leave in registers until they are committed
    uint4 va[STEPS], vb[STEPS];
    float s0, s1;
    float4 vn;
    float scg;
    unsigned bsb;                                     // the bias, still in its 16 bits
};

// RESIDENT: at most 64 input and 32 output channels - the two weight buffers hold the whole composed kernel and are
// filled once per workgroup; otherwise the weights of the next chunk are fetched during every multiply phase.

template <class T, bool RESIDENT>
__global__ void __launch_bounds__(kThreads, 1) conv_up2_fir_kernel(UpFirArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_in = smem;                    // [2][kInBytes]
    unsigned char* s_w = smem + 2 * kInBytes;      // [2][kWBytes]
    float* s_noise = reinterpret_cast<float*>(smem + 2 * kInBytes + 2 * kWBytes);   // [3][2 TH][2 TW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave & 3, ct = wave >> 2;    // columns 16 strip .. + 15 of the tile; channels 16 ct .. + 15 of the group
    const bool late = wave >= 4;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int plane = g.h * g.wid, oh = 2 * g.h, ow = 2 * g.wid;
    const T* __restrict__ w = (const T*)g.w;

    const float4 taps = *reinterpret_cast<const float4*>(g.fy);
    const float noise_gain = (g.noise && g.noise_gain) ? *g.noise_gain : 1.f;
    const float ngg = noise_gain * g.gain;
    const int chunks = (g.ci + KCH - 1) / KCH;

    // XCD k (workgroups k, k + 8, ...) walks the k-th contiguous eighth of the units: neighbouring tiles share an L2
    const int xcd = blockIdx.x & 7, per_xcd = (g.units + 7) >> 3, stride = gridDim.x >> 3;
    const int unit_end = min(g.units, (xcd + 1) * per_xcd);
    const int unit0 = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (unit0 >= unit_end) return;

    auto decode = [&](int u) __attribute__((always_inline)) {
        Tile t;
        t.m0 = (u % g.cogs) * BCO; u /= g.cogs;
        t.b0 = (u % g.tiles_b) * TW; u /= g.tiles_b;
        t.a0 = (u % g.tiles_a) * TH;
        t.n = u / g.tiles_a;
        return t;
    };
    auto advance = [&](const Pos& p) __attribute__((always_inline)) {
        Pos q = p;
        if (!p.ok) return q;
        if (p.c + 1 < chunks) { q.c = p.c + 1; return q; }
        q.unit = p.unit + stride; q.c = 0; q.idx = p.idx + 1;
        q.ok = q.unit < unit_end;
        if (q.ok) q.t = decode(q.unit);
        return q;
    };

    const int ch = 16 * ct + lrow;                 // this lane's output channel inside the group

    // ---- loads of a step into registers ------------------------------------------------------------------------------
    // A wave-slot ws = wave + 8 q (q = 0, 1) is one group of 64 loads: ws < 12 the 4 packets v = 4 (ws & 1) + (lane & 3)
    // of footprint row ws >> 1 (columns b0 + 8 v ..) for the 16 channel pairs lane >> 2 - four neighbouring lanes read
    // one 64-byte run; ws = 12 .. 14 the halo packets (columns b0 - 8 .. and b0 + 64 ..) of which only the pixel next
    // to the tile is kept: item 4 (ws - 12) + (lane & 3) = 2 row + side.
    const int cg = lane >> 2, v_lo = lane & 3;
    auto slot_geometry = [&](int q, int& ry, int& col, bool& halo, bool& live) __attribute__((always_inline)) {
        const int ws = wave + 8 * q;
        halo = ws >= 12;
        live = ws < 15;
        if (!halo) { ry = ws >> 1; col = 8 + 8 * (4 * (ws & 1) + v_lo); }       // column inside the staged row (LWP pixels)
        else { const int item = 4 * (ws - 12) + v_lo; ry = item >> 1; col = (item & 1) ? 8 + TW : 0; }
    };
    auto issue = [&](const Pos& p, Stage& st) __attribute__((always_inline)) {
        const Tile& t = p.t;
        const int c = p.c * KCH + 2 * cg;
        const bool c_ok = c < g.ci, two = c + 1 < g.ci;
        const T* src0 = (const T*)g.x + ((int64_t)t.n * g.ci + c) * plane;
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            int ry, col; bool halo, live;
            slot_geometry(q, ry, col, halo, live);
            const int iy = t.a0 - 1 + ry, px0 = t.b0 - 8 + col;
            const bool ok = c_ok && live && iy >= 0 && iy < g.h && px0 >= 0 && px0 < g.wid;
            const T* src = src0 + iy * g.wid + px0;
            st.va[q] = load16_or_zero(ok, src);                 // (unconditional loads: sg_common.h)
            st.vb[q] = load16_or_zero(ok && two, src + plane);
        }
        if (!RESIDENT && g.in_scale) {
            st.s0 = c_ok ? g.in_scale[(int64_t)t.n * g.ci + c] : 0.f;
            st.s1 = two ? g.in_scale[(int64_t)t.n * g.ci + c + 1] : 0.f;
        }
        if (p.c == 0) {   // with a tile's first chunk: its noise (one float4 per lane of waves 0..3) and epilogue constants
            if (g.noise && tid < kNoise / 4) {
                const int oy = 2 * t.a0 + (tid >> 5), ox = 2 * t.b0 + 4 * (tid & 31);
                st.vn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < oh && ox < ow) st.vn = *reinterpret_cast<const float4*>(g.noise + (int64_t)t.n * g.noise_stride + (int64_t)oy * ow + ox);
            }
            const bool ch_ok = t.m0 + ch < g.co;
            st.scg = (g.scale && ch_ok) ? g.scale[(int64_t)t.n * g.co + t.m0 + ch] : 1.f;
            st.bsb = load2_or_zero(g.bias && ch_ok, (const T*)g.bias + t.m0 + ch);     // widened when it is committed
        }
    };
    uint4 vw[RESIDENT ? 1 : WROUNDS];
    // `n` >= 0: the weights are multiplied by that sample's modulation (the resident form: pixels are then staged as
    // they are)
    auto load_weights = [&](int m0, int c0, int n, uint4 (&v)[WROUNDS]) __attribute__((always_inline)) {   // cache-resident: every workgroup reads them
        const int cw = c0 + (tid & 3) * 8;         // (kThreads is a multiple of 4: a lane's packets share their channels)
        float sc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) sc[j] = 1.f;
        const bool scaled = RESIDENT && n >= 0 && g.in_scale;
        if (scaled && cw < g.ci) {
            const float4 a = *reinterpret_cast<const float4*>(g.in_scale + (int64_t)n * g.ci + cw);
            const float4 b = *reinterpret_cast<const float4*>(g.in_scale + (int64_t)n * g.ci + cw + 4);
            sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
        }
#pragma unroll
        for (int q = 0; q < WROUNDS; q++) {
            const int id = tid + kThreads * q;
            const int o = m0 + ((id >> 2) % BCO), combo = (id >> 2) / BCO;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (id < WPACKETS && o < g.co && cw < g.ci) u = *reinterpret_cast<const uint4*>(w + ((int64_t)combo * g.co + o) * g.ci + cw);
            if (scaled) {
                u.x = Pair<T>::pack(Pair<T>::lo(u.x) * sc[0], Pair<T>::hi(u.x) * sc[1]);
                u.y = Pair<T>::pack(Pair<T>::lo(u.y) * sc[2], Pair<T>::hi(u.y) * sc[3]);
                u.z = Pair<T>::pack(Pair<T>::lo(u.z) * sc[4], Pair<T>::hi(u.z) * sc[5]);
                u.w = Pair<T>::pack(Pair<T>::lo(u.w) * sc[6], Pair<T>::hi(u.w) * sc[7]);
            }
            v[q] = u;
        }
    };
    auto store_weights = [&](int wbuf, const uint4 (&v)[WROUNDS]) __attribute__((always_inline)) {
        unsigned char* dst = s_w + wbuf * kWBytes;
#pragma unroll
        for (int q = 0; q < WROUNDS; q++) {
            const int id = tid + kThreads * q;
            if (id < WPACKETS) *reinterpret_cast<uint4*>(dst + swz(id >> 2, (id & 3) * 16)) = v[q];
        }
    };

    // ---- registers -> LDS --------------------------------------------------------------------------------------------
    float nscg = 1.f, nbsg = 0.f;      // epilogue constants of the tile whose first chunk was committed last
    auto commit = [&](const Pos& p, const Stage& st, int buf) __attribute__((always_inline)) {
        unsigned char* img = s_in + buf * kInBytes;
        const bool scale = !RESIDENT && g.in_scale;
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            int ry, col; bool halo, live;
            slot_geometry(q, ry, col, halo, live);
            if (!live) continue;
            const unsigned wa[4] = {st.va[q].x, st.va[q].y, st.va[q].z, st.va[q].w};
            const unsigned wb[4] = {st.vb[q].x, st.vb[q].y, st.vb[q].z, st.vb[q].w};
            const int r0 = ry * LWP + col;
            auto put = [&](int j) __attribute__((always_inline)) {
                const unsigned ea = wa[j >> 1] >> (16 * (j & 1)), eb = wb[j >> 1] >> (16 * (j & 1));
                unsigned out;
                if (scale) out = Pair<T>::pack(Pair<T>::lo(ea) * st.s0, Pair<T>::lo(eb) * st.s1);
                else out = (ea & 0xffffu) | (eb << 16);
                *reinterpret_cast<unsigned*>(img + swz(r0 + j, cg * 4)) = out;
            };
            if (!halo) {
#pragma unroll
                for (int j = 0; j < 8; j++) put(j);
            } else if (col == 0) put(7);       // column b0 - 1
            else put(0);                       // column b0 + 64
        }
        if (p.c == 0) {
            if (g.noise && tid < kNoise / 4) *reinterpret_cast<float4*>(s_noise + (p.idx % 3) * kNoise + 4 * tid) = st.vn;
            nscg = st.scg * g.gain;
            nbsg = Pair<T>::lo(st.bsb) * g.gain;
        }
    };

    // U[r][py][px]: intermediate row 2 (a0 - 1 + r) + py, output column phase px; a register holds columns
    // b = b0 + 16 strip + 4 lgrp + e of channel m0 + 16 ct + lrow
    f32x4_t acc[LH][2][2];
    auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < LH; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[r][i >> 1][i & 1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    const int a_lane = ch * kRow + (((lgrp ^ (lrow >> 1)) & 3) << 4);

    auto multiply = [&](int buf, int wbuf) __attribute__((always_inline)) {
        const unsigned char* img = s_in + buf * kInBytes;
        const unsigned char* wts = s_w + wbuf * kWBytes + a_lane;
#pragma unroll
        for (int dxi = 0; dxi < 3; dxi++) {   // input column b + dxi - 1
            uint4 fp[LH];
#pragma unroll
            for (int ry = 0; ry < LH; ry++) {
                const int row = ry * LWP + 8 + 16 * strip + lrow + dxi - 1;
                fp[ry] = *reinterpret_cast<const uint4*>(img + row * kRow + (((lgrp ^ (row >> 1)) & 3) << 4));
            }
#pragma unroll
            for (int rt = 0; rt < 3; rt++)     // row taps: (py 0, input row a), (py 0, a - 1), (py 1, a)
#pragma unroll
                for (int px = 0; px < 2; px++) {
                    const int combo = rt * 6 + 3 * px + (2 - dxi);
                    const int py = rt == 2 ? 1 : 0, dy = rt == 1 ? -1 : 0;
                    const uint4 fa = *reinterpret_cast<const uint4*>(wts + combo * BCO * kRow);
#pragma unroll
                    for (int r = 1 - py; r < LH; r++) Mma16<T>::run(fp[r + dy], fa, acc[r][py][px]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // vertical filter on the accumulators, epilogue, store: 8 consecutive output pixels (4 columns b x 2 phases) per
    // 16-byte store.  No loads from memory in here: the noise tile is in LDS, the constants in registers.
    auto finish = [&](const Tile& t, int idx, float scg, float bsg) __attribute__((always_inline)) {
        T* __restrict__ y = (T*)g.y;
        const int b = t.b0 + 16 * strip + 4 * lgrp, ox = 2 * b;
        if (ox >= ow || t.m0 + ch >= g.co) return;
        T* dst0 = y + ((int64_t)t.n * g.co + t.m0 + ch) * oh * ow + ox;
        const float* nsrc = s_noise + (idx % 3) * kNoise + 32 * strip + 8 * lgrp;
        const float t0 = taps.x * scg, t1 = taps.y * scg, t2 = taps.z * scg, t3 = taps.w * scg;
#pragma unroll
        for (int a = 0; a < TH; a++) {
            const int r = a + 1, oy0 = 2 * (t.a0 + a);
            if (oy0 >= oh) continue;
#pragma unroll
            for (int py = 0; py < 2; py++) {
                float nz[8];
#pragma unroll
                for (int i = 0; i < 8; i++) nz[i] = 0.f;
                if (g.noise) {
                    const float4 lo4 = *reinterpret_cast<const float4*>(nsrc + (2 * a + py) * 2 * TW);
                    const float4 hi4 = *reinterpret_cast<const float4*>(nsrc + (2 * a + py) * 2 * TW + 4);
                    nz[0] = lo4.x; nz[1] = lo4.y; nz[2] = lo4.z; nz[3] = lo4.w;
                    nz[4] = hi4.x; nz[5] = hi4.y; nz[6] = hi4.z; nz[7] = hi4.w;
                }
                // pairs of adjacent columns (adjacent registers of an accumulator) go through packed multiply-adds; the
                // filter taps carry the demodulation coefficient and the gain (t0..t3)
                f32x2_t z[2][2];    // [px][pair]
#pragma unroll
                for (int px = 0; px < 2; px++)
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        const f32x2_t nzp = {nz[4 * h2 + px], nz[4 * h2 + 2 + px]};
                        f32x2_t v = __builtin_elementwise_fma(nzp, f32x2_t{ngg, ngg}, f32x2_t{bsg, bsg});
                        const f32x4_t& u0 = py == 0 ? acc[r - 1][1][px] : acc[r][0][px];
                        const f32x4_t& u1 = py == 0 ? acc[r][0][px] : acc[r][1][px];
                        const f32x4_t& u2 = py == 0 ? acc[r][1][px] : acc[r + 1][0][px];
                        const f32x4_t& u3 = py == 0 ? acc[r + 1][0][px] : acc[r + 1][1][px];
                        v = __builtin_elementwise_fma(f32x2_t{t0, t0}, f32x2_t{u0[2 * h2], u0[2 * h2 + 1]}, v);
                        v = __builtin_elementwise_fma(f32x2_t{t1, t1}, f32x2_t{u1[2 * h2], u1[2 * h2 + 1]}, v);
                        v = __builtin_elementwise_fma(f32x2_t{t2, t2}, f32x2_t{u2[2 * h2], u2[2 * h2 + 1]}, v);
                        v = __builtin_elementwise_fma(f32x2_t{t3, t3}, f32x2_t{u3[2 * h2], u3[2 * h2 + 1]}, v);
                        const f32x2_t sl = v * f32x2_t{g.slope, g.slope};
                        z[px][h2][0] = __builtin_amdgcn_fmed3f(__builtin_fmaxf(v[0], sl[0]), -g.hi, g.hi);
                        z[px][h2][1] = __builtin_amdgcn_fmed3f(__builtin_fmaxf(v[1], sl[1]), -g.hi, g.hi);
                    }
                unsigned words[4];
#pragma unroll
                for (int e = 0; e < 4; e++) words[e] = Pair<T>::pack(z[0][e >> 1][e & 1], z[1][e >> 1][e & 1]);
                *reinterpret_cast<uint4*>(dst0 + (int64_t)(oy0 + py) * ow) = make_uint4(words[0], words[1], words[2], words[3]);
            }
        }
    };

    // ---- the walk over (tile, chunk) steps.  Step s multiplies from buffers s & 1; its pixel packets were requested
    // two steps earlier and committed during the step before it, so 2 steps' worth of reads are in flight all the time --
    Pos p0 = {unit0, 0, 0, true, decode(unit0)};
    Pos p1 = advance(p0), p2 = advance(p1);
    if (chunks == 1) p1.ok = p1.unit < unit_end, p2.ok = p2.unit < unit_end;
    Stage sa, sb;
    sa.vn = sb.vn = make_float4(0.f, 0.f, 0.f, 0.f);
    sa.s0 = sa.s1 = sb.s0 = sb.s1 = 1.f;
    issue(p0, sa);
    auto stage_resident = [&](const Tile& t) __attribute__((always_inline)) {   // the whole composed kernel, modulated for t's sample
        uint4 v[WROUNDS];
        load_weights(t.m0, 0, t.n, v);
        store_weights(0, v);
        if (chunks == 2) {
            load_weights(t.m0, KCH, t.n, v);
            store_weights(1, v);
        }
    };
    if constexpr (RESIDENT) stage_resident(p0.t);
    else {
        uint4 v[WROUNDS];
        load_weights(p0.t.m0, 0, -1, v);
        store_weights(0, v);
    }
    if (p1.ok) issue(p1, sb);
    commit(p0, sa, 0);
    float scg = nscg, bsg = nbsg;
    clear();
    __syncthreads();

    bool pending = false;     // late waves: the previous tile's results are still in the accumulators
    Tile done = p0.t;
    int done_idx = 0;
    int buf = 0;
    // `mine` held this step's packets (committed: free again), `other` holds the next step's
    [[maybe_unused]] int nstep = 0;
    auto step = [&](Stage& mine, Stage& other) __attribute__((always_inline)) {
        if (p2.ok) issue(p2, mine);
        if constexpr (!RESIDENT) {
            if (p1.ok) load_weights(p1.t.m0, p1.c * KCH, -1, vw);
        }
        // late waves: the previous tile's filter / epilogue / stores first, then multiply; early waves: multiply, then
        // (after a tile's last chunk) filter / epilogue / stores
        if (pending) { finish(done, done_idx, scg, bsg); clear(); pending = false; }
        if (p0.c == 0) { scg = nscg; bsg = nbsg; }
        multiply(buf, RESIDENT ? p0.c : buf);
        if (p0.c + 1 == chunks) {
            if (late) { pending = true; done = p0.t; done_idx = p0.idx; }
            else { finish(p0.t, p0.idx, scg, bsg); clear(); }
        }
        if (p1.ok) {
            commit(p1, other, buf ^ 1);
            if constexpr (!RESIDENT) store_weights(buf ^ 1, vw);
        }
        __syncthreads();
        if constexpr (RESIDENT) {
            if (g.in_scale && p1.ok && p1.c == 0 && p1.t.n != p0.t.n) {   // (uniform) the next tile is another sample's
                stage_resident(p1.t);
                __syncthreads();
            }
        }
        nstep++;
        buf ^= 1;
        p0 = p1; p1 = p2; p2 = advance(p2);
    };
    while (true) {
        step(sa, sb);
        if (!p0.ok) break;
        step(sb, sa);
        if (!p0.ok) break;
    }
    if (pending) finish(done, done_idx, scg, bsg);
}

template <class T>
int launch_up_fir(const UpFirArgs& g, hipStream_t stream) {
    static_assert(kLds <= 160 * 1024, "the double buffers do not fit the LDS of a CU");
    const bool resident = g.ci <= 2 * KCH && g.cogs == 1;
    const void* kernel = resident ? (const void*)conv_up2_fir_kernel<T, true> : (const void*)conv_up2_fir_kernel<T, false>;
    hipError_t e = allow_large_lds(kernel);
    if (e != hipSuccess) return fail((int)e, "conv_up2_fir: cannot reserve %zu bytes of LDS: %s", kLds, hipGetErrorString(e));
    const int64_t grid = std::max<int64_t>(8, std::min<int64_t>(((int64_t)g.units + 7) / 8 * 8, compute_units()) / 8 * 8);
    if (resident) hipLaunchKernelGGL((conv_up2_fir_kernel<T, true>), dim3((unsigned)grid), dim3(kThreads), kLds, stream, g);
    else hipLaunchKernelGGL((conv_up2_fir_kernel<T, false>), dim3((unsigned)grid), dim3(kThreads), kLds, stream, g);
    return check_launch("sg_conv_up2_fir");
}

int validate(const sg_conv_up2_fir_params* p) {
    SG_REQUIRE(p, SG_ERR_NULL_POINTER, "conv_up2_fir: params is NULL");
    SG_REQUIRE(p->x && p->w && p->y && p->fy4, SG_ERR_NULL_POINTER, "conv_up2_fir: x, w, y and fy must not be NULL");
    SG_REQUIRE(p->dtype == SG_F16 || p->dtype == SG_BF16, SG_ERR_UNSUPPORTED, "conv_up2_fir: float16 and bfloat16 tensors only");
    SG_REQUIRE(p->n >= 1 && p->ci >= 1 && p->co >= 1 && p->h >= 1 && p->w_in >= 1, SG_ERR_INVALID_ARGUMENT,
               "conv_up2_fir: sizes must be positive");
    SG_REQUIRE(p->ep.act == 1 || p->ep.act == 3, SG_ERR_INVALID_ARGUMENT, "conv_up2_fir: act must be 1 (linear) or 3 (lrelu)");
    SG_REQUIRE((int64_t)p->n * std::max(p->ci, 4 * p->co) * p->h * p->w_in <= INT32_MAX && p->h <= (1 << 14) && p->w_in <= (1 << 14),
               SG_ERR_INVALID_ARGUMENT, "conv_up2_fir: tensors of more than 2^31 - 1 elements are not supported");
    return SG_OK;
}

bool supported(const sg_conv_up2_fir_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.ci % 8 != 0 || p.ci < KCH || p.w_in % 8 != 0 || p.w_in < 32 || p.h < 2) return false;
    if (!(p.ep.gain > 0.f) || (p.ep.act == 3 && !(p.ep.alpha >= 0.f && p.ep.alpha <= 1.f))) return false;   // gain folded in, lrelu as a max
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.y) || !aligned16(p.fy4)) return false;
    if (p.ep.noise && (!aligned16(p.ep.noise) || p.ep.noise_stride_n % 4 != 0 || p.ep.noise_stride_n > INT32_MAX)) return false;
    return true;
}

}  // namespace
}  // namespace sg

extern "C" int sg_conv_up2_fir_supported(const sg_conv_up2_fir_params* p) {
    if (sg::validate(p) != SG_OK) return 0;
    return sg::supported(*p) ? 1 : 0;
}

extern "C" int sg_conv_up2_fir(const sg_conv_up2_fir_params* p, sg_stream_t stream) {
    using namespace sg;
    const int rc = validate(p);
    if (rc != SG_OK) return rc;
    SG_REQUIRE(supported(*p), SG_ERR_UNSUPPORTED,
               "conv_up2_fir: needs 16-byte aligned tensors, ci %% 8 == 0, ci >= 32, w_in %% 8 == 0, w_in >= 32, h >= 2, gain > 0");
    UpFirArgs g;
    g.x = p->x; g.w = p->w; g.y = p->y; g.in_scale = p->in_scale; g.fy = p->fy4;
    g.scale = p->ep.scale; g.noise = p->ep.noise; g.noise_gain = p->ep.noise_gain; g.bias = p->ep.bias;
    g.noise_stride = (int)p->ep.noise_stride_n;
    g.n = p->n; g.ci = p->ci; g.co = p->co; g.h = p->h; g.wid = p->w_in;
    g.tiles_a = (p->h + TH - 1) / TH;
    g.tiles_b = (p->w_in + TW - 1) / TW;
    g.cogs = (p->co + BCO - 1) / BCO;
    const int64_t units = (int64_t)g.tiles_a * g.tiles_b * g.cogs * p->n;
    SG_REQUIRE(units <= INT32_MAX / 2, SG_ERR_INVALID_ARGUMENT, "conv_up2_fir: too many tiles");
    g.units = (int)units;
    g.slope = (p->ep.act == 3) ? p->ep.alpha : 1.f;
    g.gain = p->ep.gain;
    g.hi = (p->ep.clamp >= 0.f) ? p->ep.clamp : __builtin_inff();
    if (p->dtype == SG_BF16) return launch_up_fir<__hip_bfloat16>(g, (hipStream_t)stream);
    return launch_up_fir<__half>(g, (hipStream_t)stream);
}

