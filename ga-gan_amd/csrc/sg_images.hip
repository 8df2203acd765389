// A batch of training images out of a uint8 image set that stays in device memory: gather by index, optional
// left-right mirror, and the conversion to the networks' input range, in one pass (1 byte read, one element
// written per pixel; the op is bound by HBM traffic).  Arithmetic: float32(byte) / 127.5f - 1.0f with a correctly
// rounded division, as training/training_loop.py:441 of the reference evaluates it; the mirror is the one of
// training/dataset.py:90-92.
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float to_range(unsigned byte) { return __fdiv_rn((float)byte, 127.5f) - 1.0f; }

// The 256 possible results, computed once per workgroup with the exact division.
__device__ __forceinline__ void fill_table(float* table) {
    for (int i = threadIdx.x; i < 256; i += kThreads) table[i] = to_range((unsigned)i);
    __syncthreads();
}

// One lane per packet of P = 16 / sizeof(T) pixels of one image row (rows are whole numbers of packets): every lane
// stores exactly one 16-byte packet, so a wavefront writes 1 KiB of consecutive addresses per store instruction and
// reads P * 64 consecutive bytes.  kBatch packets per lane are in flight at a time (all loads first).
constexpr int kBatch = 4;

template <class T>
__global__ void __launch_bounds__(kThreads) images_packet_kernel(sg_images_params p, int64_t packets, int packets_per_row, int64_t packets_per_sample) {
    constexpr int P = Pack<T>::N;          // 4 (float32) or 8 (16-bit types)
    constexpr int WORDS = (P + 3) / 4;     // (the 8-byte type is never launched: see launch())
    __shared__ float table[256];
    fill_table(table);
    const uint8_t* __restrict__ images = p.images;
    T* __restrict__ out = (T*)p.out;
    const int64_t sample_bytes = packets_per_sample * P;
    for (int64_t first = (int64_t)blockIdx.x * (kThreads * kBatch) + threadIdx.x; first < packets; first += (int64_t)gridDim.x * (kThreads * kBatch)) {
        unsigned words[kBatch][WORDS];
        int64_t dst[kBatch];
        bool valid[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const int64_t pk = first + u * kThreads;
            dst[u] = -1;
            valid[u] = false;
            if (pk >= packets) continue;
            const int64_t b = pk / packets_per_sample;
            const int64_t r = pk - b * packets_per_sample;
            const int64_t row = r / packets_per_row;
            const int px = (int)(r - row * packets_per_row);
            const int64_t idx = p.index ? p.index[b] : b;
            const bool mirror = p.flip && p.flip[b];
            dst[u] = b * sample_bytes + (row * packets_per_row + px) * P;
            valid[u] = idx >= 0 && idx < p.count;
            if (!valid[u]) continue;
            const int src_px = mirror ? packets_per_row - 1 - px : px;
            const unsigned* src = reinterpret_cast<const unsigned*>(images + idx * sample_bytes + (row * packets_per_row + src_px) * P);
#pragma unroll
            for (int k = 0; k < WORDS; k++) words[u][k] = mirror ? __builtin_bswap32(src[WORDS - 1 - k]) : src[k];
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (dst[u] < 0) continue;
            Pack<T> v;
#pragma unroll
            for (int j = 0; j < P; j++)
                v.v[j] = Elem<T>::store(valid[u] ? table[(words[u][j / 4] >> (8 * (j % 4))) & 0xffu] : __builtin_nanf(""));
            store_pack(out + dst[u], v);
        }
    }
}

// Any width or alignment: one lane per pixel.
template <class T>
__global__ void __launch_bounds__(kThreads) images_pixel_kernel(sg_images_params p, int64_t total, int64_t sample_elems) {
    const uint8_t* __restrict__ images = p.images;
    T* __restrict__ out = (T*)p.out;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t b = e / sample_elems;
        const int64_t r = e - b * sample_elems;
        const int64_t row = r / p.w;
        const int x = (int)(r - row * p.w);
        const int64_t idx = p.index ? p.index[b] : b;
        const bool mirror = p.flip && p.flip[b];
        float v = __builtin_nanf("");
        if (idx >= 0 && idx < p.count) v = to_range(images[idx * sample_elems + row * p.w + (mirror ? p.w - 1 - x : x)]);
        out[e] = Elem<T>::store(v);
    }
}

template <class T>
int launch(const sg_images_params& p, hipStream_t stream) {
    const int64_t sample_elems = (int64_t)p.c * p.h * p.w;
    const int64_t total = sample_elems * p.n;
    const int64_t max_blocks = 256 * 16;   // every CU several times over; the rest is looped over
    constexpr int P = Pack<T>::N;
    if (sizeof(T) <= 4 && p.w % P == 0 && (reinterpret_cast<uintptr_t>(p.images) & 3u) == 0 && aligned16(p.out)) {
        const int64_t packets = total / P;
        int64_t blocks = (packets + kThreads * kBatch - 1) / (kThreads * kBatch);
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL((images_packet_kernel<T>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, packets, p.w / P,
                           sample_elems / P);
    } else {
        int64_t blocks = (total + kThreads - 1) / kThreads;
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL((images_pixel_kernel<T>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, total, sample_elems);
    }
    return check_launch("sg_images_to_float");
}

}  // namespace
}  // namespace sg

extern "C" int sg_images_to_float(const sg_images_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->n >= 0 && p->c >= 1 && p->h >= 1 && p->w >= 1, SG_ERR_INVALID_ARGUMENT, "n must be >= 0, c, h and w positive");
    SG_REQUIRE(p->count >= 1, SG_ERR_INVALID_ARGUMENT, "the image set is empty");
    SG_REQUIRE(p->index != nullptr || p->n <= p->count, SG_ERR_INVALID_ARGUMENT,
               "without an index the batch (%d) cannot exceed the image set (%lld)", p->n, (long long)p->count);
    SG_REQUIRE(p->dtype == SG_F32 || p->dtype == SG_F16 || p->dtype == SG_BF16, SG_ERR_UNSUPPORTED, "output must be float32, float16 or bfloat16");
    if (p->n == 0) return SG_OK;
    SG_REQUIRE(p->images && p->out, SG_ERR_NULL_POINTER, "images and out must not be NULL");
    SG_DISPATCH_DTYPE(p->dtype, false, return launch<T>(*p, (hipStream_t)stream));
    return SG_OK;
}
