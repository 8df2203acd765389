// 3x3, unit-stride, same-size convolution of 16-bit tensors with 32 input channels and 32 or 64 output channels: the
// layers at 1024^2 (and the 32 -> 64 layers at 512^2), whose arithmetic intensity (144 flop/B) puts them at the HBM
// roofline once the contraction runs on the matrix cores.  The design goal is therefore "every input byte is fetched
// from memory once, every output byte stored once, and the memory system is never left idle":
//
//   * A persistent workgroup of 8 waves owns a BAND of 6 output rows of one sample and 32 output channels, and walks
//     it from left to right in steps of 64 pixels.  The 9 right-most input columns of a step stay in LDS for the next
//     one, so each step loads exactly the 64 new columns, never a halo column.  The LOADS run 8 columns ahead of the
//     outputs (step t loads input columns 64 t - 56 .. 64 t + 7 and produces output columns 64 t - 64 .. 64 t - 1;
//     step 0 only loads): the stores then cover whole 128-byte lines within one step, and the misalignment sits on
//     the loads, where the second touch of a line is a cache hit (with the outputs lagging instead, every line left
//     the L2 as three partial requests: 1.5 x the bytes on the memory side).
//   * The weights of the workgroup's channels (with the per-sample modulation folded in: w * s[n, ci]) live in
//     REGISTERS as matrix-instruction fragments for the whole band (72 registers): the multiply phase reads nothing
//     but pixels from LDS, one 16-byte read per 1.5 matrix instructions.
//   * A step's image is 8 rows x 64 pixels x 32 channels = 512 staging units of 4 channels x 8 pixels: one per lane.
//     The unit's four 16-byte packets are loaded TWO steps ahead into one of two register sets (the loads of a step
//     have two steps' time to arrive, and memory requests are in flight all the time), transposed to the
//     [pixel][channel] image with one byte-permute per two elements and written with conflict-free 8-byte LDS stores.
//   * The image buffers alternate, so one barrier per step suffices.  Waves 4..7 run their epilogue half a step late
//     (after the barrier, before their next multiplies): on every SIMD one wave multiplies while its partner does
//     vector arithmetic and stores, instead of both queueing for the matrix pipe and then both for the vector pipe.
//   * v_mfma_f32_32x32x16: 32 pixels x 32 channels per instruction; the pixel rows are permuted so that a lane ends up
//     with 16 consecutive pixels of one channel; results leave through a wave-private LDS tile so that four lanes
//     store one 64-byte run.
//   * As the data gradient of such a layer the kernel applies the activation mask of the layer's saved output while it
//     stages the incoming gradient (mask_y): no separate pass, no intermediate tensor.  (Two tensors per unit: the
//     loads then run one step ahead only.)
//
// GEMM view: M = pixels, N = 32 output channels, K = 32 input channels x 9 taps.  sg_conv2d dispatches here.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>
#include <type_traits>

namespace sg {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Buffer addressing: a resource descriptor bounds every access to one sample's tensor, a byte offset at or beyond the
// bound reads zeros and drops stores.  Out-of-image packets therefore cost one select on the offset, not a branch.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOutside = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 load16(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store16(rsrc_t r, unsigned voff, const uint4& v) {
    const u32x4_t u = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, 0, 0);
}

constexpr int kThreads = 512;
constexpr int R = 6;                     // output rows of a band
constexpr int IH = R + 2;                // image rows: row r holds input row y0 - 1 + r
constexpr int TWX = 64;                  // new input columns per step
constexpr int CARRY = 9;                 // columns kept from the previous step
constexpr int IW = TWX + CARRY;          // image columns: column i of step t holds input column 64 t - 65 + i (odd: see img_off)
constexpr int kPix = 64;                 // bytes per pixel: 32 channels
constexpr int kImg = IH * IW * kPix;     // 37376 bytes per image
constexpr int kNoise = R * TWX;          // floats of a step's noise tile
constexpr int kOutRow = 80;              // a wave's results pass through LDS as runs of 32 pixels (64 bytes) + 16 bytes of padding
constexpr int kOutWave = 2 * 32 * kOutRow;
constexpr size_t kLds = 2 * kImg + 3 * kNoise * sizeof(float) + 8 * 32 * sizeof(float) + 8 * kOutWave;

template <class T> struct Mma32;
template <> struct Mma32<__hip_bfloat16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma32<__half> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// Two 16-bit elements in a dword <-> two floats.
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

struct BandArgs {
    const void* x; const void* mask_y; const void* w; void* y;
    const float* in_scale; const float* out_scale; const void* bias;
    const float* noise; const float* noise_gain; const void* reduce_x; float* reduce_out;
    int n, ci, co, h, wid;
    int bands, cogs, items;       // bands per sample, groups of 32 output channels, work items = n x bands x cogs
    int noise_stride;
    float slope, gain, hi;        // epilogue: leaky-ReLU slope (1: linear), gain, clamp bound (+inf: none)
    float m_slope, m_hi;          // input mask: factor where mask_y <= 0, bound beyond which the gradient is zero
};

// Byte offset of 16-byte slot s of the image pixel (row r, column c).  The XOR spreads the 32 pixels that a matrix
// operand read touches (one row, one slot each) over all bank groups: ds_read_b128 serves lanes in groups of 16 whose
// pixels are 0-3, 12-15, 20-27 (or the complement) of the 32, i.e. four pixels for each value of c mod 4, at c >> 2
// equal to 0, 3, 5, 6 (mod 4: all different) plus a common offset.  It depends on the column only, so the reads of
// the image rows differ by a constant.  The odd row length puts vertically adjacent pixels into different halves of
// the 128-byte window of an 8-byte store: the 16 lanes of a store group (8 channel groups x 2 rows) never collide.
__device__ __forceinline__ int img_off(int r, int c, int s) { return (r * IW + c) * kPix + (((s ^ (c >> 2)) & 3) << 4); }


template <bool MASK> struct Staged {   // the packets of one staging unit (and of its mask), and a piece of the noise tile
    uint4 p[4];
    uint4 m[MASK ? 4 : 1];
    float4 pn;
};

template <class T, bool MASK, bool REDUCE>
__global__ void __launch_bounds__(kThreads, 2) band3x3_kernel(BandArgs g) {
    constexpr int DEPTH = (MASK || REDUCE) ? 1 : 2;   // steps between the loads of an image and their use (registers decide)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_img = smem;                                       // [2][kImg]
    float* s_noise = reinterpret_cast<float*>(smem + 2 * kImg);        // [3][R][TWX]
    float* s_red = s_noise + 3 * kNoise;                               // [8 waves][32]
    unsigned char* s_out = reinterpret_cast<unsigned char*>(s_red + 8 * 32) + (threadIdx.x >> 6) * kOutWave;   // this wave's [2][32][kOutRow]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sx = wave & 1, rg = wave >> 1;          // strip of 32 output columns, output rows 2 rg and 2 rg + 1 (rg = 3: none)
    const bool late = !MASK && wave >= 4;            // these waves finish a step's results after the step's barrier
    const int lc = lane & 31, hh = lane >> 5;         // matrix row / column index, half
    // matrix row rho of the pixel operand holds strip pixel 16 h' + 4 q + j (rho = 8 q + 4 h' + j), so that the 16
    // accumulator registers of a lane are 16 consecutive pixels
    const int pi = 16 * ((lc >> 2) & 1) + 4 * (lc >> 3) + (lc & 3);
    const int c0px = 32 * sx + pi;                    // image column of this lane's pixel for the left tap

    const int NT = (g.wid + TWX - 1) / TWX;           // output steps 1 .. NT; step 0 only loads (columns 0 .. 7)
    const int Q = NT + 1;
    const int64_t plane = (int64_t)g.h * g.wid;
    const unsigned uplane = (unsigned)plane;
    const float noise_gain = (g.noise && g.noise_gain) ? *g.noise_gain : 1.f;

    // staging unit of this lane: 4 channels u_c4, image row u_r, packet u_v of 8 pixels.  Consecutive lanes take the 8
    // channel groups of one pixel (64 contiguous bytes in LDS) and then the next image row (an odd number of pixels
    // further): 16 lanes fill all 32 banks once.
    const int u_c4 = tid & 7, u_v = (tid >> 4) & 7, u_r = 2 * (tid >> 7) + ((tid >> 3) & 1);

    const int xcd = blockIdx.x & 7, per_xcd = (g.items + 7) >> 3, stride = gridDim.x >> 3;
    const int item_end = min(g.items, (xcd + 1) * per_xcd);
    for (int item = xcd * per_xcd + (int)(blockIdx.x >> 3); item < item_end; item += stride) {
        const int cog = item % g.cogs;
        const int band = (item / g.cogs) % g.bands;
        const int n = item / (g.cogs * g.bands);
        const int y0 = band * R;
        const int cout = cog * 32 + lc;
        const unsigned in_bytes = (unsigned)g.ci * uplane * 2u, out_bytes = (unsigned)g.co * uplane * 2u;   // < 2^31: launcher
        const rsrc_t xr = make_rsrc((const T*)g.x + (int64_t)n * g.ci * plane, in_bytes);
        const rsrc_t mr = make_rsrc(MASK ? (const T*)g.mask_y + (int64_t)n * g.ci * plane : (const T*)g.x, in_bytes);
        const rsrc_t rr = make_rsrc(REDUCE ? (const T*)g.reduce_x + (int64_t)n * g.co * plane : (const T*)g.x, REDUCE ? out_bytes : 0u);
        const rsrc_t yr = make_rsrc((T*)g.y + (int64_t)n * g.co * plane, out_bytes);

        // ---- weights of this sample and channel group -> registers: fragment (tap, ks) holds input channels
        // 16 ks + 8 hh .. + 7 of output channel cout ----------------------------------------------------------------
        uint4 wreg[9][2];
        {
            const T* __restrict__ wsrc = (const T*)g.w;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int c0 = 16 * ks + 8 * hh;
                float s[8];
#pragma unroll
                for (int j = 0; j < 8; j++) s[j] = 1.f;
                if (g.in_scale) {
                    const float4 a = *reinterpret_cast<const float4*>(g.in_scale + (int64_t)n * g.ci + c0);
                    const float4 b = *reinterpret_cast<const float4*>(g.in_scale + (int64_t)n * g.ci + c0 + 4);
                    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
                }
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((int64_t)tap * g.co + cout) * g.ci + c0);
                    if (g.in_scale) {
                        v.x = Pair<T>::pack(Pair<T>::lo(v.x) * s[0], Pair<T>::hi(v.x) * s[1]);
                        v.y = Pair<T>::pack(Pair<T>::lo(v.y) * s[2], Pair<T>::hi(v.y) * s[3]);
                        v.z = Pair<T>::pack(Pair<T>::lo(v.z) * s[4], Pair<T>::hi(v.z) * s[5]);
                        v.w = Pair<T>::pack(Pair<T>::lo(v.w) * s[6], Pair<T>::hi(v.w) * s[7]);
                    }
                    wreg[tap][ks] = v;
                }
            }
        }
        // epilogue constants with the (positive) gain folded in: clamp(lrelu(v sc + noise ng + b) gain) =
        // clamp(lrelu(v (sc gain) + noise (ng gain) + b gain)), and lrelu(z) = max(z, slope z) for 0 <= slope <= 1
        const float scg = (g.out_scale ? g.out_scale[(int64_t)n * g.co + cout] : 1.f) * g.gain;
        const float bsg = (g.bias ? Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.bias)[cout])) : 0.f) * g.gain;
        const float ngg = noise_gain * g.gain;

        // ---- left border: the carried columns of the first step's image are zero ----------------------------------
        if (tid < IH * CARRY * 4) {
            const int s = tid & 3, c = (tid >> 2) % CARRY, r = (tid >> 2) / CARRY;
            *reinterpret_cast<uint4*>(s_img + img_off(r, c, s)) = make_uint4(0, 0, 0, 0);
        }

        // ---- staging: loads of a step into registers, later masked / transposed / written to LDS -------------------
        auto issue = [&](int q, Staged<MASK>& st) {
            const int col = (q - 1) * TWX + 8 + 8 * u_v, yy = y0 - 1 + u_r;
            const bool ok = col >= 0 && col < g.wid && yy >= 0 && yy < g.h;
            const unsigned off = ok ? ((unsigned)(4 * u_c4) * uplane + (unsigned)(yy * g.wid + col)) * 2u : kOutside;   // bytes
#pragma unroll
            for (int e = 0; e < 4; e++) {
                st.p[e] = load16(xr, off, e * uplane * 2u);
                if constexpr (MASK) st.m[e] = load16(mr, off, e * uplane * 2u);
            }
            // noise tile of the step (output columns 64 (q - 1) ..): lanes 384..511, one float4 each
            if (g.noise && tid >= 384) {
                const int id = tid - 384, row = y0 + (id >> 4), c = (q - 1) * TWX + 4 * (id & 15);
                st.pn = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((id >> 4) < R && row < g.h && c >= 0 && c < g.wid)
                    st.pn = *reinterpret_cast<const float4*>(g.noise + (int64_t)n * g.noise_stride + (int64_t)row * g.wid + c);
            }
        };
        auto masked = [&](unsigned d, unsigned m) -> unsigned {   // two gradient elements times the mask of two saved outputs
            const float ml = Pair<T>::lo(m), mh = Pair<T>::hi(m);
            float fl = (ml > 0.f) ? 1.f : g.m_slope, fh = (mh > 0.f) ? 1.f : g.m_slope;
            fl = (fabsf(ml) < g.m_hi) ? fl : 0.f;
            fh = (fabsf(mh) < g.m_hi) ? fh : 0.f;
            return Pair<T>::pack(Pair<T>::lo(d) * fl, Pair<T>::hi(d) * fh);
        };
        auto commit = [&](int q, Staged<MASK>& st) {
            unsigned char* img = s_img + (q & 1) * kImg;
            unsigned wd[4][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                wd[e][0] = st.p[e].x; wd[e][1] = st.p[e].y; wd[e][2] = st.p[e].z; wd[e][3] = st.p[e].w;
                if constexpr (MASK) {
                    wd[e][0] = masked(wd[e][0], st.m[e].x); wd[e][1] = masked(wd[e][1], st.m[e].y);
                    wd[e][2] = masked(wd[e][2], st.m[e].z); wd[e][3] = masked(wd[e][3], st.m[e].w);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                uint2 o;
                o.x = __builtin_amdgcn_perm(wd[1][j >> 1], wd[0][j >> 1], sel);   // channels 4 c4, 4 c4 + 1 of pixel j
                o.y = __builtin_amdgcn_perm(wd[3][j >> 1], wd[2][j >> 1], sel);
                *reinterpret_cast<uint2*>(img + img_off(u_r, CARRY + 8 * u_v + j, u_c4 >> 1) + 8 * (u_c4 & 1)) = o;
            }
            if (q >= 1 && tid < IH * CARRY * 4) {   // the 9 right-most columns of the previous image
                const unsigned char* prev = s_img + ((q - 1) & 1) * kImg;
                const int s = tid & 3, c = (tid >> 2) % CARRY, r = (tid >> 2) / CARRY;
                *reinterpret_cast<uint4*>(img + img_off(r, c, s)) = *reinterpret_cast<const uint4*>(prev + img_off(r, TWX + c, s));
            }
            if (g.noise && tid >= 384) *reinterpret_cast<float4*>(s_noise + (q % 3) * kNoise + (tid - 384) * 4) = st.pn;
        };

        // ---- a step's arithmetic ------------------------------------------------------------------------------------
        f32x16_t acc[2];
        uint4 rx[REDUCE ? 2 : 1][2];
        float red = 0.f;
        auto active = [&](int t) { return t >= 1 && rg < 3; };   // the opening step only loads
        auto load_rx = [&](int t) {
            if constexpr (REDUCE) {
#pragma unroll
                for (int o = 0; o < 2; o++)
#pragma unroll
                    for (int gq = 0; gq < 2; gq++) {
                        const int row = y0 + 2 * rg + o, col0 = (t - 1) * TWX + 32 * sx + 16 * hh + 8 * gq;
                        const bool ok = row < g.h && col0 >= 0 && col0 < g.wid;
                        rx[o][gq] = load16(rr, ok ? ((unsigned)cout * uplane + (unsigned)(row * g.wid + col0)) * 2u : kOutside, 0);
                    }
            }
        };
        auto multiply = [&](int t) {
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[o][e] = 0.f;
            const unsigned char* img = s_img + (t & 1) * kImg;
            // 12 (input row, column shift) pairs; the fragments of pair i + 1 are read while pair i multiplies
            uint4 fa[2][2];
            auto read_pair = [&](int idx, uint4 (&f)[2]) {
                const int r = 2 * rg + idx / 3, c = c0px + idx % 3;
                f[0] = *reinterpret_cast<const uint4*>(img + img_off(r, c, hh));
                f[1] = *reinterpret_cast<const uint4*>(img + img_off(r, c, 2 + hh));
            };
            read_pair(0, fa[0]);
#pragma unroll
            for (int idx = 0; idx < 12; idx++) {
                if (idx + 1 < 12) read_pair(idx + 1, fa[(idx + 1) & 1]);
                const int ir = idx / 3, kx = idx % 3;
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int ky = 0; ky < 3; ky++) {
                        const int o = ir - ky;
                        if (o == 0 || o == 1) Mma32<T>::run(fa[idx & 1][ks], wreg[ky * 3 + kx][ks], acc[o]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto epilogue = [&](int t) {
            const float* nz = s_noise + (t % 3) * kNoise;
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = acc[o][8 * gq + e];
                    if constexpr (REDUCE) {   // rx is zero outside the image
                        const unsigned wr[4] = {rx[o][gq].x, rx[o][gq].y, rx[o][gq].z, rx[o][gq].w};
#pragma unroll
                        for (int e = 0; e < 4; e++) red += v[2 * e] * Pair<T>::lo(wr[e]) + v[2 * e + 1] * Pair<T>::hi(wr[e]);
                    }
                    float nzv[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) nzv[e] = 0.f;
                    if (g.noise) {
                        const float4 a = *reinterpret_cast<const float4*>(nz + (2 * rg + o) * TWX + 32 * sx + 16 * hh + 8 * gq);
                        const float4 b = *reinterpret_cast<const float4*>(nz + (2 * rg + o) * TWX + 32 * sx + 16 * hh + 8 * gq + 4);
                        nzv[0] = a.x; nzv[1] = a.y; nzv[2] = a.z; nzv[3] = a.w; nzv[4] = b.x; nzv[5] = b.y; nzv[6] = b.z; nzv[7] = b.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float z = __builtin_fmaf(v[e], scg, __builtin_fmaf(nzv[e], ngg, bsg));
                        z = __builtin_fmaxf(z, z * g.slope);
                        v[e] = __builtin_amdgcn_fmed3f(z, -g.hi, g.hi);
                    }
                    const uint4 out = make_uint4(Pair<T>::pack(v[0], v[1]), Pair<T>::pack(v[2], v[3]),
                                                 Pair<T>::pack(v[4], v[5]), Pair<T>::pack(v[6], v[7]));
                    // results leave through this wave's own LDS tile: [row o][channel][32 pixels]
                    *reinterpret_cast<uint4*>(s_out + (o * 32 + lc) * kOutRow + 32 * hh + 16 * gq) = out;
                }
            // a lane now moves 8 consecutive pixels, four lanes one run of 32 pixels of a (row, channel): 64 contiguous
            // bytes per run instead of 64 scattered 16-byte pieces per store instruction
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int run = 16 * k + (lane >> 2), piece = lane & 3, o = k >> 1, ch = run & 31;
                const uint4 out = *reinterpret_cast<const uint4*>(s_out + run * kOutRow + 16 * piece);
                const int row = y0 + 2 * rg + o, col = (t - 1) * TWX + 32 * sx + 8 * piece;
                const bool ok = row < g.h && col >= 0 && col < g.wid;
                store16(yr, ok ? ((unsigned)(cog * 32 + ch) * uplane + (unsigned)(row * g.wid + col)) * 2u : kOutside, out);
            }
        };

        // ---- the walk ------------------------------------------------------------------------------------------------
        Staged<MASK> s0, s1;
        s0.pn = s1.pn = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();          // the previous item's last reads are done; the zeroed columns are in place
        issue(0, s0);
        if constexpr (DEPTH == 2) issue(1, s1);
        commit(0, s0);
        __syncthreads();

        // step q: `mine` held image q (committed, free again), `other` holds image q + 1 when loads run two steps ahead
        auto step = [&](int q, Staged<MASK>& mine, Staged<MASK>& other) {
            if (q + DEPTH < Q) issue(q + DEPTH, mine);
            if (late && q > 0 && active(q - 1)) epilogue(q - 1);
            if (active(q)) {
                load_rx(q);
                multiply(q);
                if (!late) epilogue(q);
            }
            if (q + 1 < Q) commit(q + 1, DEPTH == 2 ? other : mine);
            __syncthreads();
        };
        for (int q = 0; q < Q; q += 2) {
            step(q, s0, s1);
            if (q + 1 < Q) step(q + 1, s1, s0);
        }
        if (late && active(Q - 1)) epilogue(Q - 1);

        if constexpr (REDUCE) {   // lanes l and l + 32 hold the same channel; then the 8 waves through LDS; one row per band
            red += __shfl_xor(red, 32, 64);
            if (hh == 0) s_red[wave * 32 + lc] = red;
            __syncthreads();
            if (tid < 32) {
                float r = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; wv++) r += s_red[wv * 32 + tid];
                g.reduce_out[((int64_t)n * g.bands + band) * g.co + cog * 32 + tid] = r;
            }
        }
    }
}

template <class T, bool MASK, bool REDUCE>
int launch_band(const BandArgs& g, hipStream_t stream) {
    static_assert(kLds <= 160 * 1024, "band images do not fit the LDS of a CU");
    auto kernel = band3x3_kernel<T, MASK, REDUCE>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d: cannot reserve %zu bytes of LDS: %s", kLds, hipGetErrorString(e));
    const int64_t grid = std::max<int64_t>(8, std::min<int64_t>(((int64_t)g.items + 7) / 8 * 8, compute_units()) / 8 * 8);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kThreads), kLds, stream, g);
    return check_launch("sg_conv2d");
}

template <class T>
int launch_band_for(const BandArgs& g, hipStream_t stream) {
    const bool red = g.reduce_out != nullptr;
    if (g.mask_y) return red ? launch_band<T, true, true>(g, stream) : launch_band<T, true, false>(g, stream);
    return red ? launch_band<T, false, true>(g, stream) : launch_band<T, false, false>(g, stream);
}

}  // namespace

bool band3x3_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 1 || p.in_step != 1 || p.out_step != 1) return false;
    const sg_conv2d_phase& ph = p.phase[0];
    if (ph.nky != 3 || ph.nkx != 3 || ph.oy0 != 0 || ph.ox0 != 0 || ph.w_offset != 0 || ph.dy0 != -1 || ph.dx0 != -1) return false;
    if (p.oh != p.h || p.ow != p.w_in) return false;
    if (p.ci != 32 || (p.co != 32 && p.co != 64)) return false;
    if (!(p.gain > 0.f) || (p.act == 3 && !(p.alpha >= 0.f && p.alpha <= 1.f))) return false;   // the epilogue folds the gain in and takes lrelu as a max   // the weights of 32 x 32 channels fill 72 registers
    if (p.w_in % 8 != 0 || p.w_in < 64 || p.h < 8) return false;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.y) || (p.mask_y && !aligned16(p.mask_y))) return false;
    if (!aligned16(p.reduce_x)) return false;   // (NULL passes)
    if (p.in_scale && !aligned16(p.in_scale)) return false;
    if (p.noise && (!aligned16(p.noise) || p.noise_stride_n % 4 != 0)) return false;
    if ((int64_t)std::max(p.ci, p.co) * p.h * p.w_in * 2 >= (int64_t)kOutside) return false;   // byte offsets inside a sample below 2^31
    const int64_t items = (int64_t)p.n * ((p.h + R - 1) / R) * (p.co / 32);
    return items <= INT32_MAX / 2;
}

int band3x3_reduce_rows(const sg_conv2d_params& p) { return (p.h + R - 1) / R; }   // one row per band

int band3x3_launch(const sg_conv2d_params& p, hipStream_t stream) {
    BandArgs g;
    g.x = p.x; g.mask_y = p.mask_y; g.w = p.w; g.y = p.y;
    g.in_scale = p.in_scale; g.out_scale = p.out_scale; g.bias = p.bias;
    g.noise = p.noise; g.noise_gain = p.noise_gain; g.reduce_x = p.reduce_out ? p.reduce_x : nullptr; g.reduce_out = p.reduce_out;
    g.n = p.n; g.ci = p.ci; g.co = p.co; g.h = p.h; g.wid = p.w_in;
    g.bands = (p.h + R - 1) / R;
    g.cogs = p.co / 32;
    g.items = p.n * g.bands * g.cogs;
    g.noise_stride = (int)p.noise_stride_n;
    g.slope = (p.act == 3) ? p.alpha : 1.f;
    g.gain = p.gain;
    g.hi = (p.clamp >= 0.f) ? p.clamp : __builtin_inff();
    g.m_slope = p.mask_alpha;
    g.m_hi = (p.mask_clamp >= 0.f) ? p.mask_clamp : __builtin_inff();
    if (p.dtype == SG_BF16) return launch_band_for<__hip_bfloat16>(g, stream);
    return launch_band_for<__half>(g, stream);
}

}  // namespace sg

