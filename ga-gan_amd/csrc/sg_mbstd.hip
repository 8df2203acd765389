// Minibatch standard deviation: forward (statistic + concat) and first-order backward.
//
// Tiny op (32 KiB per sample at 512 x 4 x 4): the point of the kernel is to replace
// seven tensor-op launches by one.  One workgroup owns one (group, channel-set) pair:
// it reads the g samples of its group once, copies them through to the output, and
// reduces sqrt(var + 1e-8) over (channels of the set, pixels) with wave shuffles and
// one LDS step.  Arithmetic follows training/networks.py:1291-1300 of the reference.
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kMaxGroup = 64;
constexpr int kThreads = 1024;   // few workgroups exist (groups x channel sets): each is as wide as the hardware allows

template <class S>
__device__ __forceinline__ S block_sum(S v, S* scratch) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    S total = 0;
    for (int i = 0; i < (int)(blockDim.x / kWave); i++) total += scratch[i];
    return total;
}

// grid = (n / g groups, f channel sets); element e of a set runs over (c / f) * hw.
template <class T, bool BACKWARD>
__global__ void __launch_bounds__(kThreads) mbstd_kernel(sg_mbstd_params p) {
    using S = typename Elem<T>::acc_t;
    __shared__ S scratch[kThreads / kWave];
    const int groups = p.n / p.g;
    const int m = blockIdx.x, f = blockIdx.y;
    const int cs = p.c / p.f;                 // channels per set
    const int64_t set_elems = (int64_t)cs * p.hw;
    const int64_t x_sample = (int64_t)p.c * p.hw;
    const int64_t y_sample = (int64_t)(p.c + p.f) * p.hw;
    const T* __restrict__ x = (const T*)p.x + (int64_t)f * set_elems;
    const S inv_g = (S)1 / (S)p.g;

    if (!BACKWARD) {
        T* __restrict__ y = (T*)p.out;
        S partial = 0;
        for (int64_t e = threadIdx.x; e < set_elems; e += blockDim.x) {
            S mean = 0, sq = 0, vals[1];
            (void)vals;
            for (int k = 0; k < p.g; k++) mean += Elem<T>::load(x[(int64_t)(k * groups + m) * x_sample + e]);
            mean *= inv_g;
            for (int k = 0; k < p.g; k++) {
                const T raw = x[(int64_t)(k * groups + m) * x_sample + e];
                const S d = Elem<T>::load(raw) - mean;
                sq += d * d;
                y[(int64_t)(k * groups + m) * y_sample + (int64_t)f * set_elems + e] = raw;
            }
            partial += sqrt(sq * inv_g + (S)1e-8);
        }
        const S stat = block_sum(partial, scratch) / (S)set_elems;
        const T out = Elem<T>::store(stat);
        for (int t = threadIdx.x; t < p.g * p.hw; t += blockDim.x) {
            const int k = t / p.hw, px = t - k * p.hw;
            y[(int64_t)(k * groups + m) * y_sample + (int64_t)(p.c + f) * p.hw + px] = out;
        }
    } else {
        const T* __restrict__ dy = (const T*)p.dy;
        T* __restrict__ dx = (T*)p.out + (int64_t)f * set_elems;
        S partial = 0;
        for (int t = threadIdx.x; t < p.g * p.hw; t += blockDim.x) {
            const int k = t / p.hw, px = t - k * p.hw;
            partial += Elem<T>::load(dy[(int64_t)(k * groups + m) * y_sample + (int64_t)(p.c + f) * p.hw + px]);
        }
        const S dstat = block_sum(partial, scratch) / ((S)set_elems * (S)p.g);
        for (int64_t e = threadIdx.x; e < set_elems; e += blockDim.x) {
            S mean = 0, sq = 0;
            for (int k = 0; k < p.g; k++) mean += Elem<T>::load(x[(int64_t)(k * groups + m) * x_sample + e]);
            mean *= inv_g;
            for (int k = 0; k < p.g; k++) {
                const S d = Elem<T>::load(x[(int64_t)(k * groups + m) * x_sample + e]) - mean;
                sq += d * d;
            }
            const S scale = dstat / sqrt(sq * inv_g + (S)1e-8);
            for (int k = 0; k < p.g; k++) {
                const int64_t s = (int64_t)(k * groups + m);
                const S d = Elem<T>::load(x[s * x_sample + e]) - mean;
                const S through = Elem<T>::load(dy[s * y_sample + (int64_t)f * set_elems + e]);
                dx[s * x_sample + e] = Elem<T>::store(through + d * scale);
            }
        }
    }
}

template <class T>
int launch(const sg_mbstd_params& p, hipStream_t stream, bool backward) {
    dim3 grid((unsigned)(p.n / p.g), (unsigned)p.f);
    if (backward)
        hipLaunchKernelGGL((mbstd_kernel<T, true>), grid, dim3(kThreads), 0, stream, p);
    else
        hipLaunchKernelGGL((mbstd_kernel<T, false>), grid, dim3(kThreads), 0, stream, p);
    return check_launch(backward ? "sg_mbstd_backward" : "sg_mbstd_forward");
}

int validate(const sg_mbstd_params* p, bool backward) {
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->n >= 1 && p->c >= 1 && p->hw >= 1, SG_ERR_INVALID_ARGUMENT, "n, c and hw must be positive");
    SG_REQUIRE(p->g >= 1 && p->g <= kMaxGroup && p->n % p->g == 0, SG_ERR_INVALID_ARGUMENT,
               "batch %d is not divisible by group size %d (max %d)", p->n, p->g, kMaxGroup);
    SG_REQUIRE(p->f >= 1 && p->c % p->f == 0, SG_ERR_INVALID_ARGUMENT, "channels %d not divisible by %d", p->c, p->f);
    SG_REQUIRE((int64_t)p->n * (p->c + p->f) * p->hw <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "tensor is too large");
    SG_REQUIRE(p->x && p->out && (!backward || p->dy), SG_ERR_NULL_POINTER, "x, out (and dy) must not be NULL");
    return SG_OK;
}

}  // namespace
}  // namespace sg

extern "C" int sg_mbstd_forward(const sg_mbstd_params* p, sg_stream_t stream) {
    using namespace sg;
    if (int rc = validate(p, false)) return rc;
    SG_DISPATCH_DTYPE(p->dtype, true, return launch<T>(*p, (hipStream_t)stream, false));
    return SG_OK;
}

extern "C" int sg_mbstd_backward(const sg_mbstd_params* p, sg_stream_t stream) {
    using namespace sg;
    if (int rc = validate(p, true)) return rc;
    SG_DISPATCH_DTYPE(p->dtype, true, return launch<T>(*p, (hipStream_t)stream, true));
    return SG_OK;
}
