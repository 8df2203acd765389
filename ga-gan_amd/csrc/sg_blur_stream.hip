// upfirdn2d with up = down = 1 and a filter of at most 4 x 4 taps on 16-bit tensors whose rows are whole 16-byte
// packets (the blur in front of every strided convolution and behind every transposed one, and their adjoints;
// reference torch_utils/ops/upfirdn2d.cu:42-89 for this case), as a pure streaming kernel: no LDS, no barrier.
//
//   * A lane owns 8 output columns of TWO planes and walks down a band of rows.  Every input row it reads (one
//     16-byte packet per plane plus 8 bytes on either side for the filter's reach) is scattered into the four output
//     rows it contributes to: acc[row][column] += f[i][t] * in[column + t].  The four partial rows live in
//     registers and rotate; the row that has received its fourth contribution is finished and stored.
//   * The two planes sit in the halves of 64-bit register pairs, so all multiply-adds are packed (v_pk_fma_f32 with
//     the tap broadcast): 8 vector instructions per output for a full 4 x 4 filter, separable or not.
//   * A wave's 64 lanes cover 512 consecutive columns (narrower planes: whole rows of several bands): every load and
//     store instruction moves contiguous runs of up to a kilobyte per plane.  The loads of the row after next are requested before a row's arithmetic.
//   * The optional layer epilogue (demodulation, noise, bias, leaky ReLU, gain, clamp) runs on the finished row.
#include "sg_common.h"
#include "sg_upfirdn2d.h"

namespace sg {
namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

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

// (Pieces outside the plane are read from a packet of zeros - load16_or_zero, load8_or_zero - so that every load is
// unconditional: with `ok ? *src : zero` the loads sat in branches of their own, the conversion of a narrow piece
// moved in beside its load, and the wave waited there for everything it had outstanding - the rows just requested
// included: two memory round trips per row, hidden by nothing but the other waves of the CU.)
constexpr int kBand = 32;       // output rows of a wave's band
constexpr int kCols = 512;      // output columns of a wave: 8 per lane

struct Row {                    // the 16 input elements in[-4 .. 11] around a lane's 8 columns, for both planes
    uint2 l[2];
    uint4 m[2];
    uint2 r[2];
};

template <class T, int PADX, bool EP>
__global__ void __launch_bounds__(256) blur4_stream_kernel(sg_upfirdn2d_params p, sg_epilogue ep, int col0, int col_end, int lpr, int cblocks, int bands, int units) {
    // A row of the plane takes lpr = out_w / 8 lanes (at most 64: wider planes are cut into column blocks); the other
    // lanes of a wave take further bands: lane -> (sub-unit lane / lpr, columns 8 (lane % lpr) ..)
    const int lane = threadIdx.x & 63;
    const int wave_unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int sub = lane / lpr, unit = wave_unit * (64 / lpr) + sub;
    const bool live = sub < 64 / lpr && unit < units;
    const int cb = unit % cblocks, bd = (unit / cblocks) % bands, pp = live ? unit / (cblocks * bands) : 0;
    const int planes = p.in_n * p.in_c;
    const int pl0 = 2 * pp, pl1 = min(2 * pp + 1, planes - 1);     // an odd plane count repeats the last plane (stored once)
    const bool second = 2 * pp + 1 < planes;
    const int n0 = pl0 / p.in_c, c0 = pl0 - n0 * p.in_c, n1 = pl1 / p.in_c, c1 = pl1 - n1 * p.in_c;
    const T* __restrict__ x0 = (const T*)p.x + (int64_t)n0 * p.in_stride_n + (int64_t)c0 * p.in_stride_c;
    const T* __restrict__ x1 = (const T*)p.x + (int64_t)n1 * p.in_stride_n + (int64_t)c1 * p.in_stride_c;
    T* __restrict__ y0 = (T*)p.y + (int64_t)n0 * p.out_stride_n + (int64_t)c0 * p.out_stride_c;
    T* __restrict__ y1 = (T*)p.y + (int64_t)n1 * p.out_stride_n + (int64_t)c1 * p.out_stride_c;
    const int ox0 = col0 + cb * kCols + (lane - sub * lpr) * 8, oy0 = bd * kBand;
    const int rows = live ? min(kBand, p.out_h - oy0) : -3;        // (a dead lane loads and stores nothing)

    // flipped taps with the gain folded in (uniform: scalar loads); taps beyond the filter's size are zero
    float f[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float v = 0.f;
            if (t < p.f_w && i < p.f_h) {
                const int sx = p.flip ? t : p.f_w - 1 - t, sy = p.flip ? i : p.f_h - 1 - i;
                v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h] * p.gain;
            }
            f[i][t] = v;
        }

    // element e of the lane's window is input column ox0 - pad_x0 + e = in[e - PADX] of the loaded pieces
    const int ix_main = ox0 - p.pad_x0 + PADX;      // == ox0: the main packet is the aligned one
    auto load_row = [&](int k, Row& row) __attribute__((always_inline)) {
        const int iy = oy0 - p.pad_y0 + k;
        const bool row_ok = iy >= 0 && iy < p.in_h && k < rows + 3;
        const bool m_ok = row_ok && ix_main < p.in_w, l_ok = row_ok && ix_main >= 8 && ix_main - 4 < p.in_w,
                   r_ok = row_ok && ix_main + 8 < p.in_w;
        const int64_t off = (int64_t)iy * p.in_stride_h + ix_main;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const T* src = (q ? x1 : x0) + off;
            row.m[q] = load16_or_zero(m_ok, src);
            row.l[q] = load8_or_zero(l_ok, src - 4);
            row.r[q] = load8_or_zero(r_ok, src + 8);
        }
    };

    // epilogue constants of the two planes
    f32x2_t sc = {1.f, 1.f}, bs = {0.f, 0.f};
    float ngg = 0.f;
    if constexpr (EP) {
        if (ep.scale) sc = f32x2_t{ep.scale[(int64_t)n0 * p.in_c + c0], ep.scale[(int64_t)n1 * p.in_c + c1]};
        if (ep.bias) bs = f32x2_t{(float)Elem<T>::load(((const T*)ep.bias)[c0]), (float)Elem<T>::load(((const T*)ep.bias)[c1])};
        sc *= ep.gain; bs *= ep.gain;
        ngg = ((ep.noise && ep.noise_gain) ? *ep.noise_gain : ep.noise ? 1.f : 0.f) * ep.gain;
    }
    const float slope = (EP && ep.act == 3) ? ep.alpha : 1.f;
    const float hi = (EP && ep.clamp >= 0.f) ? ep.clamp * 1.f : __builtin_inff();
    const bool col_ok = ox0 < col_end;

    f32x2_t acc[4][8];
    float4 nz[2][2];
    auto load_noise = [&](int o, float4 (&v)[2]) __attribute__((always_inline)) {
        if constexpr (EP) {
            v[0] = v[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ep.noise && o >= 0 && o < rows && col_ok) {
                const float* src = ep.noise + (int64_t)n0 * ep.noise_stride_n + (int64_t)(oy0 + o) * p.out_w + ox0;
                v[0] = reinterpret_cast<const float4*>(src)[0];
                v[1] = reinterpret_cast<const float4*>(src)[1];
            }
        }
    };

    // one input row k: unpack, scatter into the four partial output rows k - i (slot (k - i) & 3), finish row k - 3
    auto consume = [&](int k, int s, const Row& row, const float4 (&nzv)[2]) __attribute__((always_inline)) {
        f32x2_t in[16];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned a = j ? row.l[0].y : row.l[0].x, b = j ? row.l[1].y : row.l[1].x;
            in[2 * j] = f32x2_t{Pair<T>::lo(a), Pair<T>::lo(b)};
            in[2 * j + 1] = f32x2_t{Pair<T>::hi(a), Pair<T>::hi(b)};
            const unsigned c = j ? row.r[0].y : row.r[0].x, d = j ? row.r[1].y : row.r[1].x;
            in[12 + 2 * j] = f32x2_t{Pair<T>::lo(c), Pair<T>::lo(d)};
            in[12 + 2 * j + 1] = f32x2_t{Pair<T>::hi(c), Pair<T>::hi(d)};
        }
        const unsigned ma[4] = {row.m[0].x, row.m[0].y, row.m[0].z, row.m[0].w};
        const unsigned mb[4] = {row.m[1].x, row.m[1].y, row.m[1].z, row.m[1].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            in[4 + 2 * j] = f32x2_t{Pair<T>::lo(ma[j]), Pair<T>::lo(mb[j])};
            in[4 + 2 * j + 1] = f32x2_t{Pair<T>::hi(ma[j]), Pair<T>::hi(mb[j])};
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int slot = (s - i) & 3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                f32x2_t v = i == 0 ? f32x2_t{0.f, 0.f} : acc[slot][j];
#pragma unroll
                for (int t = 0; t < 4; t++) v = __builtin_elementwise_fma(f32x2_t{f[i][t], f[i][t]}, in[4 - PADX + j + t], v);
                acc[slot][j] = v;
            }
        }
        const int o = k - 3, slot = (s - 3) & 3;
        if (o >= 0 && o < rows && col_ok) {
            f32x2_t out[8];
#pragma unroll
            for (int j = 0; j < 8; j++) out[j] = acc[slot][j];
            if constexpr (EP) {
                const float nzs[8] = {nzv[0].x, nzv[0].y, nzv[0].z, nzv[0].w, nzv[1].x, nzv[1].y, nzv[1].z, nzv[1].w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float nb = nzs[j] * ngg;
                    f32x2_t z = __builtin_elementwise_fma(out[j], sc, bs + f32x2_t{nb, nb});
                    const f32x2_t zs = z * f32x2_t{slope, slope};
                    z[0] = __builtin_amdgcn_fmed3f(__builtin_fmaxf(z[0], zs[0]), -hi, hi);
                    z[1] = __builtin_amdgcn_fmed3f(__builtin_fmaxf(z[1], zs[1]), -hi, hi);
                    out[j] = z;
                }
            }
            const int64_t off = (int64_t)(oy0 + o) * p.out_stride_h + ox0;
            *reinterpret_cast<uint4*>(y0 + off) = make_uint4(Pair<T>::pack(out[0][0], out[1][0]), Pair<T>::pack(out[2][0], out[3][0]),
                                                             Pair<T>::pack(out[4][0], out[5][0]), Pair<T>::pack(out[6][0], out[7][0]));
            if (second)
                *reinterpret_cast<uint4*>(y1 + off) = make_uint4(Pair<T>::pack(out[0][1], out[1][1]), Pair<T>::pack(out[2][1], out[3][1]),
                                                                 Pair<T>::pack(out[4][1], out[5][1]), Pair<T>::pack(out[6][1], out[7][1]));
        }
    };

    // ---- the walk: rows k = 0 .. rows + 2, four per trip so that the rotating slots are compile-time constants;
    // the loads of row k + 2 are requested before row k's arithmetic ---------------------------------------------------
    Row ra, rb;
    load_row(0, ra);
    load_row(1, rb);
    load_noise(-3, nz[0]);
    const int total = min(kBand, p.out_h) + 3;      // (uniform; a lane's own row count only gates its loads and stores)
    for (int k0 = 0; k0 < total; k0 += 4) {
        // rows k0, k0 + 2 travel in ra; k0 + 1, k0 + 3 in rb.  A row is copied out of its registers before they are reloaded.
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = k0 + s;
            Row cur = (s & 1) ? rb : ra;
            load_row(k + 2, (s & 1) ? rb : ra);
            load_noise(k - 2, nz[(s + 1) & 1]);      // finished by the next row
            consume(k, s, cur, nz[s & 1]);
        }
    }
}

// The same walk for down-sampling by two (up = 1, down = 2: the filter in front of the discriminator's 1x1 skip
// convolutions): a lane's 8 output columns read 16 input columns (two packets per plane plus 8 bytes on either side),
// an input row feeds two output rows (its filter rows i with 2 o + i - pad = row), and two partial rows rotate.
struct Row2 {
    uint2 l[2];
    uint4 m[2][2];
    uint2 r[2];
};

template <class T, int PADX>
__global__ void __launch_bounds__(256) down2_stream_kernel(sg_upfirdn2d_params p, int col0, int col_end, int lpr, int cblocks, int bands, int units) {
    const int lane = threadIdx.x & 63;
    const int wave_unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int sub = lane / lpr, unit = wave_unit * (64 / lpr) + sub;
    const bool live = sub < 64 / lpr && unit < units;
    const int cb = unit % cblocks, bd = (unit / cblocks) % bands, pp = live ? unit / (cblocks * bands) : 0;
    const int planes = p.in_n * p.in_c;
    const int pl0 = 2 * pp, pl1 = min(2 * pp + 1, planes - 1);
    const bool second = 2 * pp + 1 < planes;
    const int n0 = pl0 / p.in_c, c0 = pl0 - n0 * p.in_c, n1 = pl1 / p.in_c, c1 = pl1 - n1 * p.in_c;
    const T* __restrict__ x0 = (const T*)p.x + (int64_t)n0 * p.in_stride_n + (int64_t)c0 * p.in_stride_c;
    const T* __restrict__ x1 = (const T*)p.x + (int64_t)n1 * p.in_stride_n + (int64_t)c1 * p.in_stride_c;
    T* __restrict__ y0 = (T*)p.y + (int64_t)n0 * p.out_stride_n + (int64_t)c0 * p.out_stride_c;
    T* __restrict__ y1 = (T*)p.y + (int64_t)n1 * p.out_stride_n + (int64_t)c1 * p.out_stride_c;
    const int ox0 = col0 + cb * kCols + (lane - sub * lpr) * 8, oy0 = bd * kBand;
    const int rows = live ? min(kBand, p.out_h - oy0) : -1;        // (a dead lane loads and stores nothing)
    const bool col_ok = ox0 < col_end;

    float f[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float v = 0.f;
            if (t < p.f_w && i < p.f_h) {
                const int sx = p.flip ? t : p.f_w - 1 - t, sy = p.flip ? i : p.f_h - 1 - i;
                v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h] * p.gain;
            }
            f[i][t] = v;
        }

    // in[e] is input column 2 ox0 - 4 + e; output column j, tap t reads column 2 (ox0 + j) - pad_x0 + t
    const int ix_main = 2 * ox0;
    auto load_row = [&](int k, Row2& row) __attribute__((always_inline)) {
        const int iy = 2 * oy0 - p.pad_y0 + k;
        const bool row_ok = iy >= 0 && iy < p.in_h && k < 2 * rows + 2 && col_ok;
        const bool m0_ok = row_ok && ix_main < p.in_w, m1_ok = row_ok && ix_main + 8 < p.in_w,
                   l_ok = row_ok && ix_main >= 8 && ix_main - 4 < p.in_w, r_ok = row_ok && ix_main + 16 < p.in_w;
        const int64_t off = (int64_t)iy * p.in_stride_h + ix_main;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const T* src = (q ? x1 : x0) + off;
            row.m[q][0] = load16_or_zero(m0_ok, src);
            row.m[q][1] = load16_or_zero(m1_ok, src + 8);
            row.l[q] = load8_or_zero(l_ok, src - 4);
            row.r[q] = load8_or_zero(r_ok, src + 16);
        }
    };

    f32x2_t acc[2][8];
    // input row k (k & 3 == s): filter rows i = (k & 1), (k & 1) + 2 feed output rows (k - i) / 2; slot = output row & 1
    auto consume = [&](int k, int s, const Row2& row) __attribute__((always_inline)) {
        f32x2_t in[24];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned a = j ? row.l[0].y : row.l[0].x, b = j ? row.l[1].y : row.l[1].x;
            in[2 * j] = f32x2_t{Pair<T>::lo(a), Pair<T>::lo(b)};
            in[2 * j + 1] = f32x2_t{Pair<T>::hi(a), Pair<T>::hi(b)};
            const unsigned c = j ? row.r[0].y : row.r[0].x, d = j ? row.r[1].y : row.r[1].x;
            in[20 + 2 * j] = f32x2_t{Pair<T>::lo(c), Pair<T>::lo(d)};
            in[20 + 2 * j + 1] = f32x2_t{Pair<T>::hi(c), Pair<T>::hi(d)};
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned ma[4] = {row.m[0][h].x, row.m[0][h].y, row.m[0][h].z, row.m[0][h].w};
            const unsigned mb[4] = {row.m[1][h].x, row.m[1][h].y, row.m[1][h].z, row.m[1][h].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                in[4 + 8 * h + 2 * j] = f32x2_t{Pair<T>::lo(ma[j]), Pair<T>::lo(mb[j])};
                in[4 + 8 * h + 2 * j + 1] = f32x2_t{Pair<T>::hi(ma[j]), Pair<T>::hi(mb[j])};
            }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = (s & 1) + 2 * ii;
            const int slot = ((s - i) >> 1) & 1;       // (k - i) / 2 mod 2 with k = 4 m + s
#pragma unroll
            for (int j = 0; j < 8; j++) {
                f32x2_t v = i == 0 ? f32x2_t{0.f, 0.f} : acc[slot][j];
#pragma unroll
                for (int t = 0; t < 4; t++) v = __builtin_elementwise_fma(f32x2_t{f[i][t], f[i][t]}, in[4 - PADX + 2 * j + t], v);
                acc[slot][j] = v;
            }
        }
        if (s & 1) {                                   // filter row 3 completes output row (k - 3) / 2
            const int o = (k - 3) >> 1, slot = ((s - 3) >> 1) & 1;
            if (o >= 0 && o < rows && col_ok) {
                const int64_t off = (int64_t)(oy0 + o) * p.out_stride_h + ox0;
                *reinterpret_cast<uint4*>(y0 + off) = make_uint4(Pair<T>::pack(acc[slot][0][0], acc[slot][1][0]), Pair<T>::pack(acc[slot][2][0], acc[slot][3][0]),
                                                                 Pair<T>::pack(acc[slot][4][0], acc[slot][5][0]), Pair<T>::pack(acc[slot][6][0], acc[slot][7][0]));
                if (second)
                    *reinterpret_cast<uint4*>(y1 + off) = make_uint4(Pair<T>::pack(acc[slot][0][1], acc[slot][1][1]), Pair<T>::pack(acc[slot][2][1], acc[slot][3][1]),
                                                                     Pair<T>::pack(acc[slot][4][1], acc[slot][5][1]), Pair<T>::pack(acc[slot][6][1], acc[slot][7][1]));
            }
        }
    };

    Row2 ra, rb;
    load_row(0, ra);
    load_row(1, rb);
    const int total = 2 * min(kBand, p.out_h) + 2;
    for (int k0 = 0; k0 < total; k0 += 4) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = k0 + s;
            Row2 cur = (s & 1) ? rb : ra;
            load_row(k + 2, (s & 1) ? rb : ra);
            consume(k, s, cur);
        }
    }
}

template <class T, int PADX>
int launch_down2_columns(const sg_upfirdn2d_params& p, int col0, int col_end, hipStream_t stream) {
    const int width = col_end - col0;
    const int cblocks = (width + kCols - 1) / kCols, bands = (p.out_h + kBand - 1) / kBand;
    const int lpr = cblocks > 1 ? 64 : width / 8;
    const int64_t pairs = ((int64_t)p.in_n * p.in_c + 1) / 2;
    const int64_t units = pairs * cblocks * bands;
    SG_REQUIRE(units <= INT32_MAX / 2, SG_ERR_INVALID_ARGUMENT, "output has too many tiles");
    const int64_t waves = (units + 64 / lpr - 1) / (64 / lpr);
    hipLaunchKernelGGL((down2_stream_kernel<T, PADX>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p, col0, col_end, lpr,
                       cblocks, bands, (int)units);
    return check_launch("sg_upfirdn2d (down-sampling stream)");
}

template <class T, int PADX>
int launch_down2(const sg_upfirdn2d_params& p, hipStream_t stream) {
    const int packets = p.out_w / 8;
    const int head = packets > 64 ? packets / 64 * 64 : packets;
    int rc = launch_down2_columns<T, PADX>(p, 0, head * 8, stream);
    if (rc == SG_OK && head < packets) rc = launch_down2_columns<T, PADX>(p, head * 8, p.out_w, stream);
    return rc;
}

template <class T>
int launch_down2_pad(const sg_upfirdn2d_params& p, hipStream_t stream) {
    switch (p.pad_x0) {
        case 0: return launch_down2<T, 0>(p, stream);
        case 1: return launch_down2<T, 1>(p, stream);
        case 2: return launch_down2<T, 2>(p, stream);
        default: return launch_down2<T, 3>(p, stream);
    }
}

// And for up-sampling by two (up = 2, down = 1: the adjoint of the filter above, and the image up-sampling of 16-bit
// tensors).  Only every other row and column of the zero-stuffed plane is non-zero, so an output reads 2 x 2 taps: a
// lane's 16 output columns read one packet per plane plus one element on either side (loaded as 8 bytes), and input
// row r completes the output rows e = 2r - 3, 2r - 2 (its filter rows 3 and 2 on top of what row r - 1 left) and
// starts the rows 2r - 1, 2r (filter rows 1 and 0), with e = output row - pad_y0.  No rotation, 2 packed
// multiply-adds per output.
struct AddendRows { uint4 v[2][2][2]; };     // [output row of the pair an input row completes][plane][half of the lane's 16 columns]

template <class T, int PADX, bool ADD>
__global__ void __launch_bounds__(256) up2_stream_kernel(sg_upfirdn2d_params p, const void* addend, int lpr, int cblocks, int bands, int units) {
    constexpr int kColsUp = 1024;                 // 16 output columns per lane
    const int lane = threadIdx.x & 63;
    const int wave_unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int sub = lane / lpr, unit = wave_unit * (64 / lpr) + sub;
    const bool live = sub < 64 / lpr && unit < units;
    const int cb = unit % cblocks, bd = (unit / cblocks) % bands, pp = live ? unit / (cblocks * bands) : 0;
    const int planes = p.in_n * p.in_c;
    const int pl0 = 2 * pp, pl1 = min(2 * pp + 1, planes - 1);
    const bool second = 2 * pp + 1 < planes;
    const int n0 = pl0 / p.in_c, c0 = pl0 - n0 * p.in_c, n1 = pl1 / p.in_c, c1 = pl1 - n1 * p.in_c;
    const T* __restrict__ x0 = (const T*)p.x + (int64_t)n0 * p.in_stride_n + (int64_t)c0 * p.in_stride_c;
    const T* __restrict__ x1 = (const T*)p.x + (int64_t)n1 * p.in_stride_n + (int64_t)c1 * p.in_stride_c;
    T* __restrict__ y0 = (T*)p.y + (int64_t)n0 * p.out_stride_n + (int64_t)c0 * p.out_stride_c;
    T* __restrict__ y1 = (T*)p.y + (int64_t)n1 * p.out_stride_n + (int64_t)c1 * p.out_stride_c;
    // optional addend with the output's strides: y = round(round(filtered) + addend), the arithmetic of a separate add
    // (ADD: its packets are requested one input row ahead of the output rows they are added to, unconditionally - read
    // where they were used, inside `if (addend)`, each was waited for on the spot together with the rows in flight:
    // 64 x 64 planes of 128^2 -> 256^2 took 0.49 ms with an addend and 0.19 without)
    const T* __restrict__ a0 = ADD ? (const T*)addend + (int64_t)n0 * p.out_stride_n + (int64_t)c0 * p.out_stride_c : nullptr;
    const T* __restrict__ a1 = ADD ? (const T*)addend + (int64_t)n1 * p.out_stride_n + (int64_t)c1 * p.out_stride_c : nullptr;
    const int ox0 = cb * kColsUp + (lane - sub * lpr) * 16, oy0 = bd * kBand;
    const int rows = live ? min(kBand, p.out_h - oy0) : 0;
    const bool col_ok = live && ox0 < p.out_w;

    float f[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float v = 0.f;
            if (t < p.f_w && i < p.f_h) {
                const int sx = p.flip ? t : p.f_w - 1 - t, sy = p.flip ? i : p.f_h - 1 - i;
                v = p.f[(int64_t)sx * p.f_stride_w + (int64_t)sy * p.f_stride_h] * p.gain;
            }
            f[i][t] = v;
        }

    const int e0 = oy0 - p.pad_y0;
    const int r_first = (e0 >= 0) ? e0 / 2 : -((1 - e0) / 2);      // floor(e0 / 2)
    const int ix_main = ox0 / 2;                                   // in[e] is input column ix_main - 4 + e
    auto load_row = [&](int k, Row& row) __attribute__((always_inline)) {
        const int r = r_first + k;
        const bool row_ok = r >= 0 && r < p.in_h && col_ok && k < rows / 2 + 3;
        const bool m_ok = row_ok && ix_main < p.in_w, l_ok = row_ok && ix_main >= 8 && ix_main - 4 < p.in_w,
                   r_ok = row_ok && ix_main + 8 < p.in_w;
        const int64_t off = (int64_t)r * p.in_stride_h + ix_main;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const T* src = (q ? x1 : x0) + off;
            row.m[q] = load16_or_zero(m_ok, src);
            row.l[q] = load8_or_zero(l_ok, src - 4);
            row.r[q] = load8_or_zero(r_ok, src + 8);
        }
    };

    f32x2_t pend[2][16];          // [0]: output row e = 2r - 2 so far (filter row 0 of input row r - 1), [1]: e = 2r - 3 (filter row 1)
#pragma unroll
    for (int j = 0; j < 32; j++) pend[j >> 4][j & 15] = f32x2_t{0.f, 0.f};

    auto load_addend = [&](int k, AddendRows& dst) __attribute__((always_inline)) {   // for the rows input row k completes
        if constexpr (ADD) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int oy = 2 * (r_first + k) - 3 + rr + p.pad_y0;
                const bool ok = oy >= oy0 && oy < oy0 + rows && col_ok;
                const int64_t off = (int64_t)oy * p.out_stride_h + ox0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    dst.v[rr][0][h] = load16_or_zero(ok, a0 + off + 8 * h);
                    dst.v[rr][1][h] = load16_or_zero(ok && second, a1 + off + 8 * h);
                }
            }
        }
    };
    auto store_row = [&](int e, const f32x2_t (&v)[16], const uint4 (&av)[2][2]) __attribute__((always_inline)) {
        const int oy = e + p.pad_y0;
        if (oy < oy0 || oy >= oy0 + rows || !col_ok) return;
        const int64_t off = (int64_t)oy * p.out_stride_h + ox0;
        auto plus = [&](const uint4& r, const uint4& q) __attribute__((always_inline)) {   // packed results + 8 addend elements
            const unsigned rw[4] = {r.x, r.y, r.z, r.w}, qw[4] = {q.x, q.y, q.z, q.w};
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                o[j] = Pair<T>::pack(Pair<T>::lo(rw[j]) + Pair<T>::lo(qw[j]), Pair<T>::hi(rw[j]) + Pair<T>::hi(qw[j]));
            return make_uint4(o[0], o[1], o[2], o[3]);
        };
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint4 r0 = make_uint4(Pair<T>::pack(v[8 * h][0], v[8 * h + 1][0]), Pair<T>::pack(v[8 * h + 2][0], v[8 * h + 3][0]),
                                  Pair<T>::pack(v[8 * h + 4][0], v[8 * h + 5][0]), Pair<T>::pack(v[8 * h + 6][0], v[8 * h + 7][0]));
            if constexpr (ADD) r0 = plus(r0, av[0][h]);
            *reinterpret_cast<uint4*>(y0 + off + 8 * h) = r0;
            if (second) {
                uint4 r1 = make_uint4(Pair<T>::pack(v[8 * h][1], v[8 * h + 1][1]), Pair<T>::pack(v[8 * h + 2][1], v[8 * h + 3][1]),
                                      Pair<T>::pack(v[8 * h + 4][1], v[8 * h + 5][1]), Pair<T>::pack(v[8 * h + 6][1], v[8 * h + 7][1]));
                if constexpr (ADD) r1 = plus(r1, av[1][h]);
                *reinterpret_cast<uint4*>(y1 + off + 8 * h) = r1;
            }
        }
    };
    auto consume = [&](int k, const Row& row, const AddendRows& ar) __attribute__((always_inline)) {
        f32x2_t in[16];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned a = j ? row.l[0].y : row.l[0].x, b = j ? row.l[1].y : row.l[1].x;
            in[2 * j] = f32x2_t{Pair<T>::lo(a), Pair<T>::lo(b)};
            in[2 * j + 1] = f32x2_t{Pair<T>::hi(a), Pair<T>::hi(b)};
            const unsigned c = j ? row.r[0].y : row.r[0].x, d = j ? row.r[1].y : row.r[1].x;
            in[12 + 2 * j] = f32x2_t{Pair<T>::lo(c), Pair<T>::lo(d)};
            in[12 + 2 * j + 1] = f32x2_t{Pair<T>::hi(c), Pair<T>::hi(d)};
        }
        const unsigned ma[4] = {row.m[0].x, row.m[0].y, row.m[0].z, row.m[0].w};
        const unsigned mb[4] = {row.m[1].x, row.m[1].y, row.m[1].z, row.m[1].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            in[4 + 2 * j] = f32x2_t{Pair<T>::lo(ma[j]), Pair<T>::lo(mb[j])};
            in[4 + 2 * j + 1] = f32x2_t{Pair<T>::hi(ma[j]), Pair<T>::hi(mb[j])};
        }
        // horizontal: output column ox0 + j reads the taps t of its parity, input column ix_main + (j - PADX + t) / 2
        auto hrow = [&](int i, f32x2_t (&v)[16], bool accumulate) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                f32x2_t a = accumulate ? v[j] : f32x2_t{0.f, 0.f};
#pragma unroll
                for (int t = (j + PADX) & 1; t < 4; t += 2) {
                    const int num = j - PADX + t;                      // even
                    const int e = 4 + (num >= 0 ? num / 2 : -((-num) / 2));
                    a = __builtin_elementwise_fma(f32x2_t{f[i][t], f[i][t]}, in[e], a);
                }
                v[j] = a;
            }
        };
        const int r = r_first + k;
        hrow(3, pend[1], true);
        store_row(2 * r - 3, pend[1], ar.v[0]);
        hrow(2, pend[0], true);
        store_row(2 * r - 2, pend[0], ar.v[1]);
        hrow(1, pend[1], false);
        hrow(0, pend[0], false);
    };

    Row ra, rb;
    AddendRows aa, ab;
    load_row(0, ra);
    load_row(1, rb);
    load_addend(0, aa);
    const int total = min(kBand, p.out_h) / 2 + 3;
    for (int k0 = 0; k0 < total; k0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int k = k0 + s;
            Row cur = s ? rb : ra;
            load_addend(k + 1, s ? aa : ab);
            load_row(k + 2, s ? rb : ra);
            consume(k, cur, s ? ab : aa);
        }
    }
}

template <class T, int PADX>
int launch_up2(const sg_upfirdn2d_params& p, const void* addend, hipStream_t stream) {
    const int cblocks = (p.out_w + 1023) / 1024, bands = (p.out_h + kBand - 1) / kBand;
    const int lpr = cblocks > 1 ? 64 : p.out_w / 16;
    const int64_t pairs = ((int64_t)p.in_n * p.in_c + 1) / 2;
    const int64_t units = pairs * cblocks * bands;
    SG_REQUIRE(units <= INT32_MAX / 2, SG_ERR_INVALID_ARGUMENT, "output has too many tiles");
    const int64_t waves = (units + 64 / lpr - 1) / (64 / lpr);
    auto kernel = addend ? up2_stream_kernel<T, PADX, true> : up2_stream_kernel<T, PADX, false>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p, addend, lpr, cblocks, bands, (int)units);
    return check_launch("sg_upfirdn2d (up-sampling stream)");
}

template <class T>
int launch_up2_pad(const sg_upfirdn2d_params& p, const void* addend, hipStream_t stream) {
    switch (p.pad_x0) {
        case 0: return launch_up2<T, 0>(p, addend, stream);
        case 1: return launch_up2<T, 1>(p, addend, stream);
        case 2: return launch_up2<T, 2>(p, addend, stream);
        default: return launch_up2<T, 3>(p, addend, stream);
    }
}

// columns [col0, col_end) of every plane: col_end - col0 is a multiple of 8 and, beyond 512, of 512
template <class T, int PADX>
int launch_columns(const sg_upfirdn2d_params& p, const sg_epilogue* ep, int col0, int col_end, hipStream_t stream) {
    const int width = col_end - col0;
    const int cblocks = (width + kCols - 1) / kCols, bands = (p.out_h + kBand - 1) / kBand;
    const int lpr = cblocks > 1 ? 64 : width / 8;
    const int64_t pairs = ((int64_t)p.in_n * p.in_c + 1) / 2;
    const int64_t units = pairs * cblocks * bands;
    SG_REQUIRE(units <= INT32_MAX / 2, SG_ERR_INVALID_ARGUMENT, "output has too many tiles");
    const int64_t waves = (units + 64 / lpr - 1) / (64 / lpr);
    const dim3 grid((unsigned)((waves + 3) / 4));
    sg_epilogue e = {};
    if (ep) {
        e = *ep;
        hipLaunchKernelGGL((blur4_stream_kernel<T, PADX, true>), grid, dim3(256), 0, stream, p, e, col0, col_end, lpr, cblocks, bands, (int)units);
    } else {
        hipLaunchKernelGGL((blur4_stream_kernel<T, PADX, false>), grid, dim3(256), 0, stream, p, e, col0, col_end, lpr, cblocks, bands, (int)units);
    }
    return check_launch("sg_upfirdn2d (stream)");
}

// Rows of more than 64 packets are cut so that waves are full: whole blocks of 64 packets, then the remaining columns
// in a launch of their own, where a wave's lanes take that column range of as many bands as fit (1024 -> 1025 x 1032,
// 16 x 32 planes: 0.72 -> 0.64 ms; for rows of 33 packets the second launch costs more than the idle lanes).
template <class T, int PADX>
int launch_stream(const sg_upfirdn2d_params& p, const sg_epilogue* ep, hipStream_t stream) {
    const int packets = p.out_w / 8;
    int head = packets;
    if (packets > 64) head = packets / 64 * 64;
    int rc = launch_columns<T, PADX>(p, ep, 0, head * 8, stream);
    if (rc == SG_OK && head < packets) rc = launch_columns<T, PADX>(p, ep, head * 8, p.out_w, stream);
    return rc;
}

template <class T>
int launch_stream_pad(const sg_upfirdn2d_params& p, const sg_epilogue* ep, hipStream_t stream) {
    switch (p.pad_x0) {
        case 0: return launch_stream<T, 0>(p, ep, stream);
        case 1: return launch_stream<T, 1>(p, ep, stream);
        case 2: return launch_stream<T, 2>(p, ep, stream);
        default: return launch_stream<T, 3>(p, ep, stream);
    }
}

}  // namespace

bool blur_stream_eligible(const sg_upfirdn2d_params& p, const sg_epilogue* ep) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.up_x != 1 || p.up_y != 1 || p.down_x != 1 || p.down_y != 1 || p.f_w > 4 || p.f_h > 4) return false;
    if (p.in_stride_w != 1 || p.out_stride_w != 1 || p.pad_x0 < 0 || p.pad_x0 > 3) return false;
    if (p.in_w % 8 != 0 || p.out_w % 8 != 0 || p.out_w < 32) return false;      // narrower planes leave most lanes of a wave idle
    if (p.in_stride_h % 8 != 0 || p.in_stride_c % 8 != 0 || p.in_stride_n % 8 != 0) return false;
    if (p.out_stride_h % 8 != 0 || p.out_stride_c % 8 != 0 || p.out_stride_n % 8 != 0) return false;
    if (!aligned16(p.x) || !aligned16(p.y)) return false;
    if (ep) {
        if (ep->noise && (!aligned16(ep->noise) || ep->noise_stride_n % 4 != 0 || p.in_c % 2 != 0)) return false;   // both planes of a lane share the noise
        if (!(ep->gain > 0.f) || (ep->act == 3 && !(ep->alpha >= 0.f && ep->alpha <= 1.f))) return false;        // gain folded in, lrelu as a max
    }
    return true;
}

bool down2_stream_eligible(const sg_upfirdn2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.up_x != 1 || p.up_y != 1 || p.down_x != 2 || p.down_y != 2 || p.f_w > 4 || p.f_h > 4) return false;
    if (p.in_stride_w != 1 || p.out_stride_w != 1 || p.pad_x0 < 0 || p.pad_x0 > 3) return false;
    if (p.in_w % 8 != 0 || p.out_w % 8 != 0 || p.out_w < 32) return false;
    if (p.in_stride_h % 8 != 0 || p.in_stride_c % 8 != 0 || p.in_stride_n % 8 != 0) return false;
    if (p.out_stride_h % 8 != 0 || p.out_stride_c % 8 != 0 || p.out_stride_n % 8 != 0) return false;
    return aligned16(p.x) && aligned16(p.y);
}

int down2_stream_launch(const sg_upfirdn2d_params& p, hipStream_t stream) {
    if (p.dtype == SG_BF16) return launch_down2_pad<__hip_bfloat16>(p, stream);
    return launch_down2_pad<__half>(p, stream);
}

bool up2_stream_eligible(const sg_upfirdn2d_params& p) {
    if (p.dtype != SG_F16 && p.dtype != SG_BF16) return false;
    if (p.up_x != 2 || p.up_y != 2 || p.down_x != 1 || p.down_y != 1 || p.f_w > 4 || p.f_h > 4) return false;
    if (p.in_stride_w != 1 || p.out_stride_w != 1 || p.pad_x0 < 0 || p.pad_x0 > 3) return false;
    if (p.in_w % 8 != 0 || p.out_w % 16 != 0 || p.out_w < 64 || p.out_h % 2 != 0) return false;
    if (p.in_stride_h % 8 != 0 || p.in_stride_c % 8 != 0 || p.in_stride_n % 8 != 0) return false;
    if (p.out_stride_h % 8 != 0 || p.out_stride_c % 8 != 0 || p.out_stride_n % 8 != 0) return false;
    return aligned16(p.x) && aligned16(p.y);
}

int up2_stream_launch(const sg_upfirdn2d_params& p, const void* addend, hipStream_t stream) {
    if (p.dtype == SG_BF16) return launch_up2_pad<__hip_bfloat16>(p, addend, stream);
    return launch_up2_pad<__half>(p, addend, stream);
}

int blur_stream_launch(const sg_upfirdn2d_params& p, const sg_epilogue* ep, hipStream_t stream) {
    if (p.dtype == SG_BF16) return launch_stream_pad<__hip_bfloat16>(p, ep, stream);
    return launch_stream_pad<__half>(p, ep, stream);
}

}  // namespace sg
