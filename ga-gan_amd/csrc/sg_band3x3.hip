// 3x3, unit-stride, same-size convolution of 16-bit tensors with 32 or 64 channels on either side: the layers at
// 256^2 ... 1024^2, whose arithmetic intensity (144 - 288 flop/B) puts them at the HBM roofline once the contraction
// runs on the matrix cores.  The design goal is therefore "every input byte is fetched from memory once, every output
// byte stored once, and nothing else stands in the way":
//
//   * A persistent workgroup of 8 waves owns a BAND of 8 output rows of one sample and 32 output channels, and walks
//     it from left to right in steps of 64 pixels.  The 9 right-most input columns of a step stay in LDS for the next
//     one, so each step loads exactly the 64 new columns (whole 128-byte lines), never a halo column.  Outputs lag
//     the loads by 8 columns (step t produces columns 64 t - 8 .. 64 t + 55), which keeps the stores 16-byte aligned.
//   * The weights of the workgroup's channels (with the per-sample modulation folded in: w * s[n, ci]) live in
//     REGISTERS as matrix-instruction fragments for the whole band (72 registers per 32 input channels): the multiply
//     phase reads nothing but pixels from LDS, one 16-byte read per three matrix instructions.
//   * NCHW rows are transposed to the [pixel][channel] image the matrix instruction wants with one byte-permute per
//     two elements (4 channels x 8 pixels per lane) and 8-byte LDS stores that are bank-conflict free.
//   * Loads for the next phase are issued before the current phase's multiplies and committed after them; the image
//     buffers rotate, so one barrier per phase suffices.
//   * v_mfma_f32_32x32x16: 32 pixels x 32 channels per instruction; the pixel rows are permuted so that a lane ends up
//     with 16 consecutive pixels of one channel (two 16-byte stores), no transposition on the way out.
//   * As the data gradient of such a layer the kernel applies the activation mask of the layer's saved output while it
//     stages the incoming gradient (mask_y): no separate pass, no intermediate tensor.
//
// GEMM view: M = pixels, N = 32 output channels, K = input channels x 9 taps.  sg_conv2d dispatches here.
#include "sg_common.h"
#include "sg_conv3x3.h"

#include <algorithm>

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
constexpr int TWX = 64;                  // new input columns per step
constexpr int CARRY = 9;                 // columns kept from the previous step
constexpr int IW = TWX + CARRY;          // image columns: column i holds input column 64 t - 9 + i (odd: see img_off)
constexpr int kPix = 64;                 // bytes per pixel: 32 channels
// A band has R = 8 output rows (10 image rows: 640 staging units of 4 channels x 8 pixels, lanes 0..127 take two) or,
// where registers are short (masked input: a lane holds the packets of two tensors; reduction operand), R = 6 (8 image
// rows: one unit per lane).
constexpr int band_rows(bool lean) { return lean ? 6 : 8; }
constexpr int kNoise = 8 * TWX * 4;      // noise tile of a step, float32

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

template <class T, int KC, bool MASK, bool REDUCE>
__global__ void __launch_bounds__(kThreads, 2) band3x3_kernel(BandArgs g) {
    constexpr int NBUF = KC + 1;
    constexpr int R = band_rows(MASK || REDUCE), IH = R + 2, kImg = IH * IW * kPix;
    constexpr bool TWO_UNITS = IH > 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_img = smem;                                       // [NBUF][kImg]
    float* s_noise = reinterpret_cast<float*>(smem + NBUF * kImg);     // [2][8][TWX]
    float* s_red = s_noise + 2 * 8 * TWX;                              // [8 waves][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sx = wave & 1, rg = wave >> 1;          // strip of 32 output columns, output rows 2 rg and 2 rg + 1
    const int lc = lane & 31, hh = lane >> 5;         // matrix row / column index, half
    // matrix row rho of the pixel operand holds strip pixel 16 h' + 4 q + j (rho = 8 q + 4 h' + j), so that the 16
    // accumulator registers of a lane are 16 consecutive pixels
    const int pi = 16 * ((lc >> 2) & 1) + 4 * (lc >> 3) + (lc & 3);
    const int c0px = 32 * sx + pi;                    // image column of this lane's pixel for the left tap

    const int NT = (g.wid + TWX - 1) / TWX;           // loading steps; step NT only finishes the last 8 columns
    const int Q = (NT + 1) * KC;
    const int64_t plane = (int64_t)g.h * g.wid;
    const unsigned uplane = (unsigned)plane;
    const float noise_gain = (g.noise && g.noise_gain) ? *g.noise_gain : 1.f;

    // staging unit u: 4 channels c4, image row r, packet v of 8 pixels.  Consecutive lanes take the 8 channel groups
    // of one pixel (64 contiguous bytes in LDS) and then the next image row (an odd number of pixels further): 16
    // lanes fill all 32 banks once.
    const int u_c4 = tid & 7, u_v = (tid >> 4) & 7;
    const int u_rA = 2 * (tid >> 7) + ((tid >> 3) & 1);     // rows 0..7
    const int u_rB = 8 + ((tid >> 3) & 1);                  // rows 8, 9: lanes 0..127 only
    const bool has_B = TWO_UNITS && tid < 128;

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

        // ---- weights of this sample and channel group -> registers: fragment (kc, tap, ks) holds input channels
        // 32 kc + 16 ks + 8 hh .. + 7 of output channel cout ---------------------------------------------------------
        uint4 wreg[KC][9][2];
        {
            const T* __restrict__ wsrc = (const T*)g.w;
#pragma unroll
            for (int kc = 0; kc < KC; kc++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int c0 = 32 * kc + 16 * ks + 8 * hh;
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
                        wreg[kc][tap][ks] = v;
                    }
                }
        }
        const float sc = g.out_scale ? g.out_scale[(int64_t)n * g.co + cout] : 1.f;
        const float bs = g.bias ? Pair<T>::lo((unsigned)__builtin_bit_cast(unsigned short, ((const T*)g.bias)[cout])) : 0.f;

        // ---- left border: the carried columns of the first step's images are zero --------------------------------
        for (int e = tid; e < KC * IH * CARRY * 4; e += kThreads) {
            const int s = e & 3, c = (e >> 2) % CARRY, r = ((e >> 2) / CARRY) % IH, b = (e >> 2) / (CARRY * IH);
            *reinterpret_cast<uint4*>(s_img + b * kImg + img_off(r, c, s)) = make_uint4(0, 0, 0, 0);
        }

        // ---- staging: loads of a phase into registers, later masked / transposed / written to LDS -----------------
        uint4 pa[4], pb[TWO_UNITS ? 4 : 1], ma[MASK ? 4 : 1], mb[MASK && TWO_UNITS ? 4 : 1];
        float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
        auto issue = [&](int q) {
            const int t = q / KC, kc = q - t * KC;
            const int col = t * TWX + 8 * u_v;
            const bool col_ok = t < NT && col < g.wid;
            const int ya = y0 - 1 + u_rA, yb = y0 - 1 + u_rB;
            const bool ok_a = col_ok && ya >= 0 && ya < g.h, ok_b = has_B && col_ok && yb >= 0 && yb < g.h;
            const unsigned base = ((unsigned)(32 * kc + 4 * u_c4) * uplane + (unsigned)(ya * g.wid + col)) * 2u;   // bytes
            const unsigned off_a = ok_a ? base : kOutside;
            const unsigned off_b = ok_b ? base + (unsigned)((u_rB - u_rA) * g.wid) * 2u : kOutside;
#pragma unroll
            for (int e = 0; e < 4; e++) {
#ifdef BAND_ABL_NOLOAD
                pa[e] = make_uint4(q, e, tid, 1);
                if constexpr (TWO_UNITS) pb[e] = make_uint4(q, e, tid, 2);
                continue;
#endif
                pa[e] = load16(xr, off_a, e * uplane * 2u);
                if constexpr (TWO_UNITS) pb[e] = load16(xr, off_b, e * uplane * 2u);
                if constexpr (MASK) {
                    ma[e] = load16(mr, off_a, e * uplane * 2u);
                    if constexpr (TWO_UNITS) mb[e] = load16(mr, off_b, e * uplane * 2u);
                }
            }
            // noise tile of the step this phase opens (output columns 64 t - 8 ..): lanes 384..511, one float4 each
            if (g.noise && kc == 0 && tid >= 384) {
                const int id = tid - 384, row = y0 + (id >> 4), c = t * TWX - 8 + 4 * (id & 15);
                pn = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((id >> 4) < R && row < g.h && c >= 0 && c < g.wid)
                    pn = *reinterpret_cast<const float4*>(g.noise + (int64_t)n * g.noise_stride + (int64_t)row * g.wid + c);
            }
        };
        auto masked = [&](unsigned d, unsigned m) -> unsigned {   // two gradient elements times the mask of two saved outputs
            const float ml = Pair<T>::lo(m), mh = Pair<T>::hi(m);
            float fl = (ml > 0.f) ? 1.f : g.m_slope, fh = (mh > 0.f) ? 1.f : g.m_slope;
            fl = (fabsf(ml) < g.m_hi) ? fl : 0.f;
            fh = (fabsf(mh) < g.m_hi) ? fh : 0.f;
            return Pair<T>::pack(Pair<T>::lo(d) * fl, Pair<T>::hi(d) * fh);
        };
        auto commit_unit = [&](unsigned char* img, int r, auto& p, auto& m) {
            unsigned wd[4][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                wd[e][0] = p[e].x; wd[e][1] = p[e].y; wd[e][2] = p[e].z; wd[e][3] = p[e].w;
                if constexpr (MASK) {
                    wd[e][0] = masked(wd[e][0], m[e].x); wd[e][1] = masked(wd[e][1], m[e].y);
                    wd[e][2] = masked(wd[e][2], m[e].z); wd[e][3] = masked(wd[e][3], m[e].w);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                uint2 o;
                o.x = __builtin_amdgcn_perm(wd[1][j >> 1], wd[0][j >> 1], sel);   // channels 4 c4, 4 c4 + 1 of pixel j
                o.y = __builtin_amdgcn_perm(wd[3][j >> 1], wd[2][j >> 1], sel);
#ifdef BAND_ABL_NOCOMMIT
                asm volatile("" :: "v"(o.x), "v"(o.y));
#else
                *reinterpret_cast<uint2*>(img + img_off(r, CARRY + 8 * u_v + j, u_c4 >> 1) + 8 * (u_c4 & 1)) = o;
#endif
            }
        };
        auto commit = [&](int q) {
            const int t = q / KC, kc = q - t * KC;
            unsigned char* img = s_img + (q % NBUF) * kImg;
            commit_unit(img, u_rA, pa, ma);
            if constexpr (TWO_UNITS) {
                if (has_B) commit_unit(img, u_rB, pb, mb);
            }
            if (q >= KC) {   // the 9 right-most columns of the same chunk's previous image
                const unsigned char* prev = s_img + ((q - KC) % NBUF) * kImg;
                if (tid < IH * CARRY * 4) {
                    const int s = tid & 3, c = (tid >> 2) % CARRY, r = (tid >> 2) / CARRY;
                    *reinterpret_cast<uint4*>(img + img_off(r, c, s)) = *reinterpret_cast<const uint4*>(prev + img_off(r, TWX + c, s));
                }
            }
            if (g.noise && kc == 0 && tid >= 384)
                *reinterpret_cast<float4*>(s_noise + (t & 1) * 8 * TWX + (tid - 384) * 4) = pn;
        };

        float red = 0.f;
        __syncthreads();          // the previous item's last reads are done; the zeroed columns are in place
        issue(0);
        commit(0);
        __syncthreads();

        f32x16_t acc[2];
        for (int q = 0; q < Q; q++) {
            const int t = q / KC, kc = q - t * KC;
            if (q + 1 < Q) issue(q + 1);
            const bool active = (t < NT || sx == 0) && 2 * rg < R;   // the closing step only has the 8 left-most columns left
            uint4 rx[REDUCE ? 2 : 1][2];
            if (REDUCE && kc == KC - 1 && active) {
#pragma unroll
                for (int o = 0; o < 2; o++)
#pragma unroll
                    for (int gq = 0; gq < 2; gq++) {
                        const int row = y0 + 2 * rg + o, col0 = t * TWX - 8 + 32 * sx + 16 * hh + 8 * gq;
                        const bool ok = row < g.h && col0 >= 0 && col0 < g.wid;
                        rx[o][gq] = load16(rr, ok ? ((unsigned)cout * uplane + (unsigned)(row * g.wid + col0)) * 2u : kOutside, 0);
                    }
            }
            if (active) {
                if (kc == 0) {
#pragma unroll
                    for (int o = 0; o < 2; o++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[o][e] = 0.f;
                }
                const unsigned char* img = s_img + (q % NBUF) * kImg;
#pragma unroll
                for (int kcc = 0; kcc < KC; kcc++) {
                    if (kcc != kc) continue;     // static register indices: the chunk loop is unrolled
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
#ifdef BAND_ABL_NOMFMA
                                if (o == 0 || o == 1) { acc[o][0] += __builtin_bit_cast(float, fa[idx & 1][ks].x ^ wreg[kcc][ky * 3 + kx][ks].y); }
#else
                                if (o == 0 || o == 1) Mma32<T>::run(fa[idx & 1][ks], wreg[kcc][ky * 3 + kx][ks], acc[o]);
#endif
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }

                // ---- epilogue of the step: after its last chunk ---------------------------------------------------
                if (kc == KC - 1) {
                    const float* nz = s_noise + (t & 1) * 8 * TWX;
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const int row = y0 + 2 * rg + o;
#pragma unroll
                        for (int gq = 0; gq < 2; gq++) {
                            const int col0 = t * TWX - 8 + 32 * sx + 16 * hh + 8 * gq;
                            const bool ok = row < g.h && col0 >= 0 && col0 < g.wid;   // rx is zero otherwise
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; e++) v[e] = acc[o][8 * gq + e];
                            if constexpr (REDUCE) {
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
#ifdef BAND_ABL_NOEPI
                                continue;
#endif
                                float z = v[e] * sc + nzv[e] * noise_gain + bs;
                                z = (z > 0.f) ? z : z * g.slope;
                                v[e] = fminf(fmaxf(z * g.gain, -g.hi), g.hi);
                            }
                            const uint4 out = make_uint4(Pair<T>::pack(v[0], v[1]), Pair<T>::pack(v[2], v[3]),
                                                         Pair<T>::pack(v[4], v[5]), Pair<T>::pack(v[6], v[7]));
#ifdef BAND_ABL_NOSTORE
                            asm volatile("" :: "v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
#else
                            store16(yr, ok ? ((unsigned)cout * uplane + (unsigned)(row * g.wid + col0)) * 2u : kOutside, out);
#endif
                        }
                    }
                }
            }
            if (q + 1 < Q) commit(q + 1);
            __syncthreads();
        }

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

template <class T, int KC, bool MASK, bool REDUCE>
int launch_band(const BandArgs& g, hipStream_t stream) {
    constexpr size_t lds = (size_t)(KC + 1) * (band_rows(MASK || REDUCE) + 2) * IW * kPix + 2 * kNoise + 8 * 32 * sizeof(float);
    static_assert(lds <= 160 * 1024, "band images do not fit the LDS of a CU");
    auto kernel = band3x3_kernel<T, KC, MASK, REDUCE>;
    hipError_t e = allow_large_lds((const void*)kernel);
    if (e != hipSuccess) return fail((int)e, "conv2d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    const int64_t grid = std::max<int64_t>(8, std::min<int64_t>(((int64_t)g.items + 7) / 8 * 8, compute_units()) / 8 * 8);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kThreads), lds, stream, g);
    return check_launch("sg_conv2d");
}

template <class T>
int launch_band_for(const BandArgs& g, hipStream_t stream) {
    const bool red = g.reduce_out != nullptr;
    if (g.mask_y) return red ? launch_band<T, 1, true, true>(g, stream) : launch_band<T, 1, true, false>(g, stream);
    return red ? launch_band<T, 1, false, true>(g, stream) : launch_band<T, 1, false, false>(g, stream);
}

}  // namespace

bool band3x3_eligible(const sg_conv2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.num_phases != 1 || p.in_step != 1 || p.out_step != 1) return false;
    const sg_conv2d_phase& ph = p.phase[0];
    if (ph.nky != 3 || ph.nkx != 3 || ph.oy0 != 0 || ph.ox0 != 0 || ph.w_offset != 0 || ph.dy0 != -1 || ph.dx0 != -1) return false;
    if (p.oh != p.h || p.ow != p.w_in) return false;
    if (p.ci != 32 || (p.co != 32 && p.co != 64)) return false;   // the weights of 32 x 32 channels fill 72 registers
    if (p.w_in % 8 != 0 || p.w_in < 64 || p.h < 8) return false;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.y) || (p.mask_y && !aligned16(p.mask_y))) return false;
    if (p.reduce_out && !aligned16(p.reduce_x)) return false;
    if (p.in_scale && !aligned16(p.in_scale)) return false;
    if (p.noise && (!aligned16(p.noise) || p.noise_stride_n % 4 != 0)) return false;
    if ((int64_t)std::max(p.ci, p.co) * p.h * p.w_in * 2 >= (int64_t)kOutside) return false;   // byte offsets inside a sample below 2^31
    const int R = band_rows(p.mask_y != nullptr || p.reduce_out != nullptr);
    const int64_t items = (int64_t)p.n * ((p.h + R - 1) / R) * (p.co / 32);
    return items <= INT32_MAX / 2;
}

int band3x3_reduce_rows(const sg_conv2d_params& p) { const int R = band_rows(true); return (p.h + R - 1) / R; }   // one row per band of a reducing launch

int band3x3_launch(const sg_conv2d_params& p, hipStream_t stream) {
    BandArgs g;
    g.x = p.x; g.mask_y = p.mask_y; g.w = p.w; g.y = p.y;
    g.in_scale = p.in_scale; g.out_scale = p.out_scale; g.bias = p.bias;
    g.noise = p.noise; g.noise_gain = p.noise_gain; g.reduce_x = p.reduce_out ? p.reduce_x : nullptr; g.reduce_out = p.reduce_out;
    g.n = p.n; g.ci = p.ci; g.co = p.co; g.h = p.h; g.wid = p.w_in;
    const int R = band_rows(p.mask_y != nullptr || p.reduce_out != nullptr);
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
