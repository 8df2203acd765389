// Optimizer step of the training loop over a list of tensors in one launch: gradient
// sanitising (NaN -> 0, +-Inf -> +-grad_limit; reference training_loop.py:509,
// misc.nan_to_num) followed by the Adam update the reference's optimizer applies
// (torch.optim.Adam, no weight decay, no amsgrad; constructed at training_loop.py:303-312),
// and the exponential moving average of the generator weights (training_loop.py:516-523)
// as a second list operation.  Streaming kernels: every element is read and written once.
//
// The tensor table travels by value in the kernel arguments (no device-side table to
// allocate or copy): up to kMaxTensors tensors per launch, cut into blocks of
// kBlockElems elements; block_start[i] is the first block of tensor i.
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kMaxTensors = SG_OPTIM_MAX_TENSORS;
constexpr int kThreads = 256;
constexpr int kBlockElems = kThreads * 16;

struct AdamTable {
    float* param[kMaxTensors];
    float* grad[kMaxTensors];
    float* exp_avg[kMaxTensors];
    float* exp_avg_sq[kMaxTensors];
    float step_size[kMaxTensors];        // lr / (1 - beta1^t)
    float sqrt_bc2[kMaxTensors];         // sqrt(1 - beta2^t)
    int64_t numel[kMaxTensors];
    int block_start[kMaxTensors + 1];
    int count;
    float beta1, beta2, eps, grad_limit;
};

struct LerpTable {
    float* dst[kMaxTensors];
    const float* src[kMaxTensors];
    int64_t numel[kMaxTensors];
    int block_start[kMaxTensors + 1];
    int count;
    float weight;                        // dst = lerp(src, dst, weight)
};

// start + weight * (end - start), in the form that is exact at the end point the weight is nearer to (the
// library's lerp; with beta1 = 0 the first moment is then the gradient itself, bit for bit)
__device__ __forceinline__ float lerp(float start, float end, float weight) {
    return (weight < 0.5f) ? start + weight * (end - start) : end - (end - start) * (1.f - weight);
}

template <class Table>
__device__ __forceinline__ int find_tensor(const Table& t, int block) {
    int lo = 0, hi = t.count - 1;        // last i with block_start[i] <= block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.block_start[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(kThreads) adam_kernel(const AdamTable t) {
    const int ti = find_tensor(t, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - t.block_start[ti]) * kBlockElems;
    const int64_t n = t.numel[ti];
    float* __restrict__ p = t.param[ti];
    float* __restrict__ g = t.grad[ti];
    float* __restrict__ m = t.exp_avg[ti];
    float* __restrict__ v = t.exp_avg_sq[ti];
    const float step_size = t.step_size[ti], sqrt_bc2 = t.sqrt_bc2[ti];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    auto update = [&](float& pv, float& gv, float& mv, float& vv) {
        float gs = gv;
        if (gs != gs) gs = 0.f;
        else if (gs == INFINITY) gs = t.grad_limit;
        else if (gs == -INFINITY) gs = -t.grad_limit;
        gv = gs;                                            // the sanitised gradient is what stays in .grad
        mv = lerp(mv, gs, 1.f - t.beta1);
        vv = vv * t.beta2 + gs * gs * (1.f - t.beta2);
        const float denom = sqrtf(vv) / sqrt_bc2 + t.eps;
        pv = pv - step_size * (mv / denom);
    };
#pragma unroll
    for (int k = 0; k < kBlockElems / (kThreads * 4); k++) {
        const int64_t i = base + ((int64_t)k * kThreads + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            update(pv.x, gv.x, mv.x, vv.x);
            update(pv.y, gv.y, mv.y, vv.y);
            update(pv.z, gv.z, mv.z, vv.z);
            update(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(g + i) = gv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int64_t j = i; j < n && j < i + 4; j++) update(p[j], g[j], m[j], v[j]);
        }
    }
}

__global__ void __launch_bounds__(kThreads) lerp_kernel(const LerpTable t) {
    const int ti = find_tensor(t, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - t.block_start[ti]) * kBlockElems;
    const int64_t n = t.numel[ti];
    float* __restrict__ d = t.dst[ti];
    const float* __restrict__ s = t.src[ti];
    const bool vec = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15u) == 0;
    const float w = t.weight;
#pragma unroll
    for (int k = 0; k < kBlockElems / (kThreads * 4); k++) {
        const int64_t i = base + ((int64_t)k * kThreads + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 dv = *reinterpret_cast<float4*>(d + i);
            const float4 sv = *reinterpret_cast<const float4*>(s + i);
            dv.x = lerp(sv.x, dv.x, w); dv.y = lerp(sv.y, dv.y, w); dv.z = lerp(sv.z, dv.z, w); dv.w = lerp(sv.w, dv.w, w);
            *reinterpret_cast<float4*>(d + i) = dv;
        } else {
            for (int64_t j = i; j < n && j < i + 4; j++) d[j] = lerp(s[j], d[j], w);
        }
    }
}

}  // namespace
}  // namespace sg

extern "C" int sg_adam_step(const sg_adam_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->count >= 0 && p->count <= kMaxTensors, SG_ERR_INVALID_ARGUMENT, "adam_step: %d tensors (at most %d per call)",
               p->count, kMaxTensors);
    SG_REQUIRE(p->beta1 >= 0.f && p->beta1 < 1.f && p->beta2 >= 0.f && p->beta2 < 1.f && p->eps >= 0.f && p->grad_limit > 0.f,
               SG_ERR_INVALID_ARGUMENT, "adam_step: betas must lie in [0, 1), eps >= 0, grad_limit > 0");
    if (p->count == 0) return SG_OK;
    SG_REQUIRE(p->tensors != nullptr, SG_ERR_NULL_POINTER, "adam_step: tensors is NULL");
    AdamTable t;
    t.count = 0;
    t.beta1 = p->beta1; t.beta2 = p->beta2; t.eps = p->eps; t.grad_limit = p->grad_limit;
    int64_t blocks = 0;
    for (int i = 0; i < p->count; i++) {
        const sg_adam_tensor& e = p->tensors[i];
        SG_REQUIRE(e.numel >= 0 && e.step >= 1, SG_ERR_INVALID_ARGUMENT, "adam_step: tensor %d: numel %lld, step %d", i,
                   (long long)e.numel, e.step);
        if (e.numel == 0) continue;
        SG_REQUIRE(e.param && e.grad && e.exp_avg && e.exp_avg_sq, SG_ERR_NULL_POINTER, "adam_step: tensor %d has a NULL pointer", i);
        const int k = t.count++;
        t.param[k] = e.param; t.grad[k] = e.grad; t.exp_avg[k] = e.exp_avg; t.exp_avg_sq[k] = e.exp_avg_sq;
        // bias corrections in double, as the library computes them on the host
        const double bc1 = 1.0 - pow((double)p->beta1, (double)e.step), bc2 = 1.0 - pow((double)p->beta2, (double)e.step);
        t.step_size[k] = (float)((double)p->lr / bc1);
        t.sqrt_bc2[k] = (float)sqrt(bc2);
        t.numel[k] = e.numel;
        t.block_start[k] = (int)blocks;
        blocks += (e.numel + kBlockElems - 1) / kBlockElems;
        SG_REQUIRE(blocks <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "adam_step: too many elements");
    }
    if (t.count == 0) return SG_OK;
    t.block_start[t.count] = (int)blocks;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, t);
    return check_launch("sg_adam_step");
}

extern "C" int sg_lerp_multi(const sg_lerp_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->count >= 0 && p->count <= kMaxTensors, SG_ERR_INVALID_ARGUMENT, "lerp_multi: %d tensors (at most %d per call)",
               p->count, kMaxTensors);
    if (p->count == 0) return SG_OK;
    SG_REQUIRE(p->tensors != nullptr, SG_ERR_NULL_POINTER, "lerp_multi: tensors is NULL");
    LerpTable t;
    t.count = 0;
    t.weight = p->beta;
    int64_t blocks = 0;
    for (int i = 0; i < p->count; i++) {
        const sg_lerp_tensor& e = p->tensors[i];
        SG_REQUIRE(e.numel >= 0, SG_ERR_INVALID_ARGUMENT, "lerp_multi: tensor %d: numel %lld", i, (long long)e.numel);
        if (e.numel == 0) continue;
        SG_REQUIRE(e.dst && e.src, SG_ERR_NULL_POINTER, "lerp_multi: tensor %d has a NULL pointer", i);
        const int k = t.count++;
        t.dst[k] = e.dst; t.src[k] = e.src; t.numel[k] = e.numel;
        t.block_start[k] = (int)blocks;
        blocks += (e.numel + kBlockElems - 1) / kBlockElems;
        SG_REQUIRE(blocks <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "lerp_multi: too many elements");
    }
    if (t.count == 0) return SG_OK;
    t.block_start[t.count] = (int)blocks;
    hipLaunchKernelGGL(lerp_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, t);
    return check_launch("sg_lerp_multi");
}
