// Largest magnitudes of up to four float tensors per call, into a ring of slots in device memory (sg_absmax.h).
#include "sg_absmax.h"

#include <algorithm>
#include <atomic>

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
__global__ void __launch_bounds__(256) absmax_kernel(FourTensors t, unsigned* slot) {
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

}  // namespace

int absmax_measure(hipStream_t stream, const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc,
                   const float* d, int64_t nd, const unsigned** slot) {
    static std::atomic<unsigned> next{0};
    unsigned* ring = ring_of_current_device();
    if (ring == nullptr) return fail((int)hipErrorInvalidSymbol, "absmax: cannot resolve the slot ring on the current device");
    unsigned* s = ring + (size_t)(next.fetch_add(1, std::memory_order_relaxed) % kSlots) * 4;
    FourTensors t{{a, b, c, d}, {a ? na : 0, b ? nb : 0, c ? nc : 0, d ? nd : 0}};
    int64_t longest = 0;
    for (int i = 0; i < 4; i++) longest = std::max(longest, t.count[i]);
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>((longest / 4 + 1023) / 1024, 1), 1024);
    hipLaunchKernelGGL(absmax_reset_kernel, dim3(1), dim3(64), 0, stream, s);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks, 4), dim3(256), 0, stream, t, s);
    *slot = s;
    return check_launch("absmax");
}

}  // namespace sg
