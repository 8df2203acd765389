// Fused bias + activation + gain + clamp, and its first / second order backward.
//
// Pure streaming op: every operand is read once and the result written once, so
// the kernel is bound by HBM bandwidth.  Each lane moves 16-byte packets
// (4 x f32 / 8 x f16|bf16 / 2 x f64); a wave covers 1 KiB per access.  The grid
// is capped and strides over the array so that large tensors do not pay for
// block scheduling.  Semantics follow torch_utils/ops/bias_act.cu:38-146 of the
// reference: the backward modes recover the pre-gain activation from the saved
// output (yref / gain), the clamp mask is taken from yref, swish uses xref.
#include "sg_common.h"

namespace sg {
namespace {

template <class S> __device__ __forceinline__ S sg_exp(S v);
template <> __device__ __forceinline__ float sg_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double sg_exp<double>(double v) { return exp(v); }
template <class S> __device__ __forceinline__ S sg_log(S v);
template <> __device__ __forceinline__ float sg_log<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double sg_log<double>(double v) { return log(v); }

struct Scalars {
    float alpha, gain, clamp;
};

// One element.  A = activation index (1..9), G = 0 forward / 1 first-order / 2 second-order.
template <class S, int A>
__device__ __forceinline__ S bias_act_one(int G, S x, S b, S xref, S yref, S dy, S alpha, S gain, S clamp) {
    const S one = (S)1, two = (S)2, exp_range = (S)80, half_exp_range = (S)40;
    const S selu_scale = (S)1.0507009873554804934193349852946;
    const S selu_alpha = (S)1.6732632423543772848170429916717;
    const S yy = (gain != (S)0) ? yref / gain : (S)0;
    S y = (S)0;
    if (G == 0) x += b; else xref += b;

    if (A == 1) {
        if (G == 0 || G == 1) y = x;
    } else if (A == 2) {
        if (G == 0) y = (x > 0) ? x : (S)0;
        if (G == 1) y = (yy > 0) ? x : (S)0;
    } else if (A == 3) {
        if (G == 0) y = (x > 0) ? x : x * alpha;
        if (G == 1) y = (yy > 0) ? x : x * alpha;
    } else if (A == 4) {
        if (G == 0) { S c = sg_exp(x); S d = one / c; y = (x < -exp_range) ? -one : (x > exp_range) ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    } else if (A == 5) {
        if (G == 0) y = (x < -exp_range) ? (S)0 : one / (sg_exp(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    } else if (A == 6) {
        if (G == 0) y = (x >= 0) ? x : sg_exp(x) - one;
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0) ? (S)0 : x * (yy + one);
    } else if (A == 7) {
        if (G == 0) y = (x >= 0) ? selu_scale * x : (selu_scale * selu_alpha) * (sg_exp(x) - one);
        if (G == 1) y = (yy >= 0) ? x * selu_scale : x * (yy + selu_scale * selu_alpha);
        if (G == 2) y = (yy >= 0) ? (S)0 : x * (yy + selu_scale * selu_alpha);
    } else if (A == 8) {
        if (G == 0) y = (x > exp_range) ? x : sg_log(sg_exp(x) + one);
        if (G == 1) y = x * (one - sg_exp(-yy));
        if (G == 2) { S c = sg_exp(-yy); y = x * c * (one - c); }
    } else if (A == 9) {
        if (G == 0) {
            y = (x < -exp_range) ? (S)0 : x / (sg_exp(-x) + one);
        } else {
            S c = sg_exp(xref);
            S d = c + one;
            if (G == 1) y = (xref > half_exp_range) ? x : x * c * (xref + d) / (d * d);
            else y = (xref > half_exp_range) ? (S)0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -exp_range) ? (S)0 : xref / (sg_exp(-xref) + one) * gain;
        }
    }

    y *= gain * dy;
    if (clamp >= 0) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0) ? clamp : -clamp;
        else y = (yref > -clamp && yref < clamp) ? y : (S)0;
    }
    return y;
}

// PACKED: every pointer is 16-byte aligned and step_b is a multiple of the packet
// length, so one packet never straddles two bias entries; the tail (size_x not a
// multiple of the packet) is handled element-wise by the last lanes.
template <class T, int A, bool PACKED>
__global__ void __launch_bounds__(256) bias_act_kernel(sg_bias_act_params p) {
    using S = typename Elem<T>::acc_t;
    constexpr int N = Pack<T>::N;
    const int G = p.grad;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clamp = (S)p.clamp;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ b = (const T*)p.b;
    const T* __restrict__ xref = (const T*)p.xref;
    const T* __restrict__ yref = (const T*)p.yref;
    const T* __restrict__ dy = (const T*)p.dy;
    T* __restrict__ y = (T*)p.y;
    const unsigned size_b = (unsigned)p.size_b;
    const unsigned step_b = p.step_b > 0xffffffffLL ? 0xffffffffu : (unsigned)p.step_b;  // size_x < 2^31
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;

    if (PACKED) {
        const int64_t packets = p.size_x / N;
        const unsigned step_p = (unsigned)(p.step_b / N);  // packets per bias entry
        for (int64_t k = tid; k < packets; k += stride) {
            const int64_t i = k * N;
            Pack<T> vx = load_pack(x + i), vxr, vyr, vdy, out;
            if (xref) vxr = load_pack(xref + i);
            if (yref) vyr = load_pack(yref + i);
            if (dy) vdy = load_pack(dy + i);
            S bias = (S)0;
            if (b) bias = Elem<T>::load(b[((unsigned)k / step_p) % size_b]);
#pragma unroll
            for (int j = 0; j < N; j++) {
                S r = bias_act_one<S, A>(G, Elem<T>::load(vx.v[j]), bias, xref ? Elem<T>::load(vxr.v[j]) : (S)0,
                                         yref ? Elem<T>::load(vyr.v[j]) : (S)0, dy ? Elem<T>::load(vdy.v[j]) : (S)1,
                                         alpha, gain, clamp);
                out.v[j] = Elem<T>::store(r);
            }
            store_pack(y + i, out);
        }
    }
    // Element-wise path: everything when not PACKED, else only the tail.
    const int64_t first = PACKED ? (p.size_x / N) * N : 0;
    for (int64_t i = first + tid; i < p.size_x; i += stride) {
        S bias = (S)0;
        if (b) bias = Elem<T>::load(b[((unsigned)i / step_b) % size_b]);
        S r = bias_act_one<S, A>(G, Elem<T>::load(x[i]), bias, xref ? Elem<T>::load(xref[i]) : (S)0,
                                 yref ? Elem<T>::load(yref[i]) : (S)0, dy ? Elem<T>::load(dy[i]) : (S)1, alpha, gain,
                                 clamp);
        y[i] = Elem<T>::store(r);
    }
}

template <class T, int A>
int launch_act(const sg_bias_act_params& p, hipStream_t stream) {
    constexpr int N = Pack<T>::N;
    const bool packed = aligned16(p.x) && aligned16(p.y) && (!p.xref || aligned16(p.xref)) &&
                        (!p.yref || aligned16(p.yref)) && (!p.dy || aligned16(p.dy)) &&
                        (!p.b || p.step_b % N == 0) && p.size_x >= N;
    const int block = 256;
    const int64_t work = packed ? (p.size_x + N - 1) / N : p.size_x;
    int64_t grid = (work + block - 1) / block;
    const int64_t cap = 256 * 8;  // 8 blocks per CU keeps every SIMD busy; the rest is grid-strided
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    if (packed)
        hipLaunchKernelGGL((bias_act_kernel<T, A, true>), dim3((unsigned)grid), dim3(block), 0, stream, p);
    else
        hipLaunchKernelGGL((bias_act_kernel<T, A, false>), dim3((unsigned)grid), dim3(block), 0, stream, p);
    return check_launch("sg_bias_act");
}

template <class T>
int launch_dtype(const sg_bias_act_params& p, hipStream_t stream) {
    switch (p.act) {
        case 1: return launch_act<T, 1>(p, stream);
        case 2: return launch_act<T, 2>(p, stream);
        case 3: return launch_act<T, 3>(p, stream);
        case 4: return launch_act<T, 4>(p, stream);
        case 5: return launch_act<T, 5>(p, stream);
        case 6: return launch_act<T, 6>(p, stream);
        case 7: return launch_act<T, 7>(p, stream);
        case 8: return launch_act<T, 8>(p, stream);
        case 9: return launch_act<T, 9>(p, stream);
    }
    return fail(SG_ERR_INVALID_ARGUMENT, "no kernel for activation index %d", p.act);
}

}  // namespace
}  // namespace sg

extern "C" int sg_bias_act(const sg_bias_act_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->size_x >= 0, SG_ERR_INVALID_ARGUMENT, "size_x must be non-negative");
    SG_REQUIRE(p->size_x <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "x is too large");
    SG_REQUIRE(p->grad >= 0 && p->grad <= 2, SG_ERR_INVALID_ARGUMENT, "grad must be 0, 1 or 2");
    SG_REQUIRE(p->act >= 1 && p->act <= 9, SG_ERR_INVALID_ARGUMENT, "no kernel for activation index %d", p->act);
    if (p->size_x == 0) return SG_OK;
    SG_REQUIRE(p->x && p->y, SG_ERR_NULL_POINTER, "x and y must not be NULL");
    SG_REQUIRE(!p->b || (p->size_b >= 1 && p->step_b >= 1), SG_ERR_INVALID_ARGUMENT,
               "bias needs size_b >= 1 and step_b >= 1");
    SG_DISPATCH_DTYPE(p->dtype, true, return launch_dtype<T>(*p, (hipStream_t)stream));
    return SG_OK;
}
