// Largest magnitudes of up to four float tensors per call, into a ring of slots in device memory (sg_absmax.h).
#include "sg_absmax.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "sg_common.h"

namespace sg {
namespace {

constexpr int kSlots = 512;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned amax_ring[kSlots * 4];

struct FourTensors {
    const float* ptr[4];
    int64_t count[4];
};

__global__ void __launch_bounds__(64) absmax_reset_kernel(unsigned* slot) {
    if (threadIdx.x < 4) slot[threadIdx.x] = 0u;
}

// blockIdx.y: which tensor; the blocks of a row stride over the tensor's 16-byte packets (and its ragged ends).
// `clear`: the slot of the stream's next call, zeroed here so that that call needs no launch for it.
__global__ void __launch_bounds__(256) absmax_kernel(FourTensors t, unsigned* slot, unsigned* clear) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) clear[threadIdx.x] = 0u;
    const int which = blockIdx.y;
    const float* __restrict__ x = t.ptr[which];
    const int64_t n = t.count[which];
    if ((int64_t)blockIdx.x * 1024 >= n) return;   // (uniform per workgroup: an absent tensor, or a short one's surplus workgroups)
    // magnitudes compare like their bit patterns; a NaN pattern (above infinity's) wins and is kept
    unsigned best = 0u;
    const int64_t head = min<int64_t>(n, (int64_t)((16 - (reinterpret_cast<uintptr_t>(x) & 15u)) & 15u) / 4);
    const int64_t packets = (n - head) / 4;
    const uint4* __restrict__ xp = reinterpret_cast<const uint4*>(x + head);
    auto fold = [&](const uint4& v) {
        best = max(max(best, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < packets; i += 4 * stride) {   // four packets in flight per lane
        const uint4 v0 = xp[i], v1 = xp[i + stride], v2 = xp[i + 2 * stride], v3 = xp[i + 3 * stride];
        fold(v0); fold(v1); fold(v2); fold(v3);
    }
    for (; i < packets; i += stride) fold(xp[i]);
    if (blockIdx.x == 0) {
        const int64_t tail0 = head + packets * 4;
        for (int64_t i = threadIdx.x; i < head + (n - tail0); i += blockDim.x) {
            const int64_t e = i < head ? i : tail0 + (i - head);
            best = max(best, __builtin_bit_cast(unsigned, x[e]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, off));
    // one atomic per workgroup: thousands of them on one word would take longer than the reads
    __shared__ unsigned s_best[4];
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(s_best[0], s_best[1]), max(s_best[2], s_best[3]));
        if (best != 0u) atomicMax(slot + which, best);
    }
}

unsigned* ring_of_current_device() {
    static std::atomic<unsigned*> ring[kMaxDevices];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    unsigned* ptr = ring[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {
        // (the array's name is a host-side handle; its address on this device comes from the runtime)
        if (hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(amax_ring)) != hipSuccess || ptr == nullptr) return nullptr;
        ring[dev].store(ptr, std::memory_order_release);
    }
    return ptr;
}

// One float4 of scaled values -> the two half pieces of each, as two 8-byte packets.
__device__ __forceinline__ void split4(const float (&v)[4], uint2& p0, uint2& p1) {
    unsigned short h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const _Float16 a = (_Float16)v[j];
        h0[j] = __builtin_bit_cast(unsigned short, a);
        h1[j] = __builtin_bit_cast(unsigned short, (_Float16)(v[j] - (float)a));
    }
    p0 = make_uint2(h0[0] | ((unsigned)h0[1] << 16), h0[2] | ((unsigned)h0[3] << 16));
    p1 = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
}

// x[n][c][pixels] -> split rows.  A workgroup transposes 32 channels x 64 pixels through LDS: reads run along
// pixels (256 bytes per channel), writes along channels (a pixel's chunk is 128 contiguous bytes).
__global__ void __launch_bounds__(256) split_pixels_kernel(const float* __restrict__ x, const float* __restrict__ scale, int c,
                                                           int64_t pixels, int chunks, const unsigned* slot, int tensor_word,
                                                           int scale_word, unsigned char* __restrict__ out) {
    __shared__ float s[32][65];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int cb = blockIdx.y, n = blockIdx.z;
    float mul, inv;
    half_range_scale(slot, tensor_word, scale_word, scale != nullptr, mul, inv);
    const float* xn = x + (int64_t)n * c * pixels;
    const bool vec = (pixels % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    for (int it = threadIdx.x; it < 32 * 16; it += 256) {
        const int ch = it >> 4, q = it & 15;
        const int cc = cb * 32 + ch;
        const int64_t p = p0 + q * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (cc < c) {
            const float f = mul * (scale ? scale[(int64_t)n * c + cc] : 1.f);
            const float* src = xn + (int64_t)cc * pixels + p;
            if (vec && p + 3 < pixels) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                v[0] = t.x * f; v[1] = t.y * f; v[2] = t.z * f; v[3] = t.w * f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (p + j < pixels) v[j] = src[j] * f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s[ch][q * 4 + j] = v[j];
    }
    __syncthreads();
    const int px = threadIdx.x >> 2, q = threadIdx.x & 3;   // pixel of the tile, 8 channels of the chunk
    if (p0 + px >= pixels) return;
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { a[j] = s[q * 8 + j][px]; b[j] = s[q * 8 + 4 + j][px]; }
    uint2 a0, a1, b0, b1;
    split4(a, a0, a1);
    split4(b, b0, b1);
    unsigned char* dst = out + (((int64_t)n * chunks + cb) * pixels + p0 + px) * 128 + q * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, b0.x, b0.y);
    *reinterpret_cast<uint4*>(dst + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
}

// w[rows][c] -> split rows: a lane makes 8 channels of one row.
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ w, int64_t rows, int c, int chunks,
                                                         const unsigned* slot, int tensor_word, unsigned char* __restrict__ out) {
    float mul, inv;
    half_range_scale(slot, tensor_word, 0, false, mul, inv);
    const int64_t items = rows * chunks * 4;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(it & 3);
        const int64_t rc = it >> 2;
        const int cb = (int)(rc % chunks);
        const int64_t r = rc / chunks;
        const float* src = w + r * c + cb * 32 + q * 8;
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            a[j] = (cb * 32 + q * 8 + j < c) ? src[j] * mul : 0.f;
            b[j] = (cb * 32 + q * 8 + 4 + j < c) ? src[4 + j] * mul : 0.f;
        }
        uint2 a0, a1, b0, b1;
        split4(a, a0, a1);
        split4(b, b0, b1);
        unsigned char* dst = out + ((int64_t)cb * rows + r) * 128 + q * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, b0.x, b0.y);
        *reinterpret_cast<uint4*>(dst + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    }
}

}  // namespace

void* split_workspace(hipStream_t stream, size_t bytes) {
    struct Buffer { void* ptr = nullptr; size_t bytes = 0; };
    static std::mutex mutex;
    static std::map<std::pair<int, hipStream_t>, Buffer> buffers;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        fail((int)hipErrorInvalidDevice, "split workspace: no usable current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mutex);
    Buffer& b = buffers[{dev, stream}];
    if (b.bytes < bytes) {
        // (work queued on the old buffer has its own reference to nothing: wait for the stream before freeing)
        if (b.ptr) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(b.ptr);
            b = Buffer{};
        }
        const size_t want = bytes + bytes / 4;
        void* ptr = nullptr;
        const hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) {
            fail((int)e, "split workspace: cannot allocate %zu bytes: %s", want, hipGetErrorString(e));
            return nullptr;
        }
        b.ptr = ptr;
        b.bytes = want;
    }
    return b.ptr;
}

int split_pixels(hipStream_t stream, const float* x, const float* scale, int n, int c, int64_t pixels, const unsigned* slot,
                 int tensor_word, int scale_word, void* out) {
    if (n == 0 || pixels == 0) return SG_OK;
    const int chunks = (int)split_chunks(c);
    const int64_t tiles = (pixels + 63) / 64;
    SG_REQUIRE(tiles <= INT32_MAX && chunks <= 65535 && n <= 65535, SG_ERR_INVALID_ARGUMENT, "split: tensor is too large");
    hipLaunchKernelGGL(split_pixels_kernel, dim3((unsigned)tiles, (unsigned)chunks, (unsigned)n), dim3(256), 0, stream, x, scale, c,
                       pixels, chunks, slot, tensor_word, scale_word, (unsigned char*)out);
    return check_launch("split_pixels");
}

int split_rows(hipStream_t stream, const float* w, int64_t rows, int c, const unsigned* slot, int tensor_word, void* out) {
    if (rows == 0) return SG_OK;
    const int chunks = (int)split_chunks(c);
    const int64_t items = rows * chunks * 4;
    const unsigned blocks = (unsigned)std::min<int64_t>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(split_rows_kernel, dim3(blocks), dim3(256), 0, stream, w, rows, c, chunks, slot, tensor_word, (unsigned char*)out);
    return check_launch("split_rows");
}

int absmax_measure(hipStream_t stream, const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc,
                   const float* d, int64_t nd, const unsigned** slot) {
    // A call's measuring kernel zeroes the slot that the next call on the same stream will take, so a run of calls
    // on one stream launches no kernel to reset slots.  The reset kernel remains for a call whose slot was not
    // prepared that way: the first on a device, one on another stream than the call before it, and any call under
    // stream capture (a replayed graph finds the slots as its previous replay left them).
    struct Prepared { hipStream_t stream; unsigned slot; bool valid; };
    static std::mutex mutex;
    static Prepared prepared[kMaxDevices];
    static unsigned next = 0;
    unsigned* ring = ring_of_current_device();
    if (ring == nullptr) return fail((int)hipErrorInvalidSymbol, "absmax: cannot resolve the slot ring on the current device");
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone;
    unsigned idx, following;
    bool reset;
    {
        std::lock_guard<std::mutex> lock(mutex);
        idx = next++ % kSlots;
        following = (idx + 1) % kSlots;
        Prepared& p = prepared[dev];
        reset = capturing || !p.valid || p.stream != stream || p.slot != idx;
        p = Prepared{stream, following, !capturing};
    }
    unsigned* s = ring + (size_t)idx * 4;
    FourTensors t{{a, b, c, d}, {a ? na : 0, b ? nb : 0, c ? nc : 0, d ? nd : 0}};
    int64_t longest = 0;
    for (int i = 0; i < 4; i++) longest = std::max(longest, t.count[i]);
    // (16 packets per lane, at most 1024 workgroups per tensor: every workgroup ends in an atomic on the tensor's one
    // word, and with 4 packets per lane a tensor of a few megabytes spent 15 us on 600 atomics and 4 us on its reads)
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>((longest / 4 + 4095) / 4096, 1), 1024);
    if (reset) hipLaunchKernelGGL(absmax_reset_kernel, dim3(1), dim3(64), 0, stream, s);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks, 4), dim3(256), 0, stream, t, s, ring + (size_t)following * 4);
    *slot = s;
    return check_launch("absmax");
}

int absmax_into(hipStream_t stream, const float* a, int64_t na, unsigned* dst) {
    FourTensors t{{nullptr, a, nullptr, nullptr}, {0, a ? na : 0, 0, 0}};
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>((na / 4 + 4095) / 4096, 1), 1024);
    hipLaunchKernelGGL(absmax_reset_kernel, dim3(1), dim3(64), 0, stream, dst);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks, 4), dim3(256), 0, stream, t, dst, dst + 4);
    return check_launch("absmax");
}

}  // namespace sg
