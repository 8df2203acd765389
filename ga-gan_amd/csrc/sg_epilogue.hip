// Backward of the layer epilogue  y = clamp(act(v * scale[n,c] + noise * noise_gain + bias[c]) * gain)
// in one pass over dy and y: the gradient w.r.t. v, and the reductions that give the gradients of
// the bias, of the per-sample scale (demodulation coefficients) and of the noise strength.  The
// pre-activation value is recovered from the saved output (leaky ReLU is invertible; where the
// output was clamped the gradient is zero and the value does not matter), so nothing but y is
// kept from the forward pass.  One workgroup per segment of kSegment elements of a (sample, channel)
// plane - or, for planes of at most kSmallPlane elements (the 4^2 ... 64^2 layers, where a workgroup per
// plane leaves most of its lanes without a packet and ends in a reduction through LDS for a few hundred
// elements), one wave per plane, four planes per workgroup; HBM-bound: reads dy and y once, writes dz
// once (or not at all: dz == NULL leaves only the reductions, for callers whose convolution kernels
// apply the mask themselves).
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kSegment = 16384;   // elements of a plane per workgroup
constexpr int kSmallPlane = 4096; // planes up to this size: one wave each

// GROUP: the threads that share a plane segment - the workgroup (256) or a wave (64).
template <class T, int GROUP>
__global__ void __launch_bounds__(256) epilogue_bwd_kernel(sg_epilogue_bwd_params p) {
    constexpr int VEC = Pack<T>::N;
    static_assert(GROUP == 256 || GROUP == kWave, "a plane segment belongs to the workgroup or to one wave");
    __shared__ float scratch[3][256 / kWave];
    const int tid = (GROUP == 256) ? threadIdx.x : threadIdx.x % GROUP;     // within the group
    const int plane = (GROUP == 256) ? blockIdx.x : blockIdx.x * (256 / GROUP) + threadIdx.x / GROUP;
    if (GROUP != 256 && plane >= p.n * p.c) return;                          // (no barrier on this path)
    const int n = plane / p.c, c = plane - n * p.c;
    const int hw = p.h * p.w;
    const sg_epilogue& ep = p.ep;
    const float scale = ep.scale ? ep.scale[plane] : 1.f;
    const float bias = ep.bias ? (float)Elem<T>::load(((const T*)ep.bias)[c]) : 0.f;
    const float ng = (ep.noise && ep.noise_gain) ? *ep.noise_gain : 1.f;
    const float* __restrict__ nz = ep.noise ? ep.noise + (int64_t)n * ep.noise_stride_n : nullptr;
    const T* __restrict__ dy = (const T*)p.dy + (int64_t)plane * hw;
    const T* __restrict__ y = (const T*)p.y + (int64_t)plane * hw;
    T* __restrict__ dz = p.dz ? (T*)p.dz + (int64_t)plane * hw : nullptr;
    const int seg0 = blockIdx.y * kSegment, seg1 = min(hw, seg0 + kSegment);
    const float inv_gain = (ep.gain != 0.f) ? 1.f / ep.gain : 0.f;
    const float inv_alpha = (ep.alpha != 0.f) ? 1.f / ep.alpha : 0.f;
    float sum_g = 0.f, sum_gv = 0.f, sum_gn = 0.f;

    auto one = [&](float yv, float dyv, float noise) -> float {
        float g = dyv * ep.gain;
        if (ep.act == 3 && !(yv > 0.f)) g *= ep.alpha;
        if (ep.clamp >= 0.f && !(yv > -ep.clamp && yv < ep.clamp)) g = 0.f;
        sum_g += g;
        if (p.dscale) {
            float t = yv * inv_gain;                       // act(pre-activation)
            if (ep.act == 3 && t < 0.f) t *= inv_alpha;    // pre-activation
            sum_gv += g * (t - noise * ng - bias);         // = g * v * scale
        }
        sum_gn += g * noise;
        return g * scale;
    };

    const bool vec = (hw % VEC == 0) && aligned16(p.dy) && aligned16(p.y) && aligned16(p.dz);   // kSegment is a multiple of VEC
    if (vec) {
        // two packets per lane and step: both operands (and the noise row, as whole packets where its rows allow)
        // are requested before either packet is worked on
        const bool nz_vec = nz && (hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(nz) & 15u) == 0) && VEC % 4 == 0;
        auto noise_of = [&](int k, float (&out)[VEC]) {
            if (nz_vec) {
#pragma unroll
                for (int q = 0; q < VEC / 4; q++) {
                    const float4 v = *reinterpret_cast<const float4*>(nz + k * VEC + q * 4);
                    out[q * 4] = v.x; out[q * 4 + 1] = v.y; out[q * 4 + 2] = v.z; out[q * 4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; j++) out[j] = nz ? nz[k * VEC + j] : 0.f;
            }
        };
        auto packet = [&](int k, const Pack<T>& vy, const Pack<T>& vd, const float (&nv)[VEC]) {
            Pack<T> out;
#pragma unroll
            for (int j = 0; j < VEC; j++)
                out.v[j] = Elem<T>::store(one((float)Elem<T>::load(vy.v[j]), (float)Elem<T>::load(vd.v[j]), nv[j]));
            if (dz) store_pack(dz + k * VEC, out);
        };
        const int k1 = seg1 / VEC, step = GROUP;
        int k = seg0 / VEC + tid;
        for (; k + step < k1; k += 2 * step) {
            const Pack<T> vy0 = load_pack(y + k * VEC), vd0 = load_pack(dy + k * VEC);
            const Pack<T> vy1 = load_pack(y + (k + step) * VEC), vd1 = load_pack(dy + (k + step) * VEC);
            float n0[VEC], n1[VEC];
            noise_of(k, n0);
            noise_of(k + step, n1);
            packet(k, vy0, vd0, n0);
            packet(k + step, vy1, vd1, n1);
        }
        for (; k < k1; k += step) {
            const Pack<T> vy = load_pack(y + k * VEC), vd = load_pack(dy + k * VEC);
            float nv[VEC];
            noise_of(k, nv);
            packet(k, vy, vd, nv);
        }
    } else {
        for (int i = seg0 + tid; i < seg1; i += GROUP) {
            const T out = Elem<T>::store(one((float)Elem<T>::load(y[i]), (float)Elem<T>::load(dy[i]), nz ? nz[i] : 0.f));
            if (dz) dz[i] = out;
        }
    }

    float tot[3] = {sum_g, sum_gv, sum_gn};
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) tot[q] += __shfl_down(tot[q], off, kWave);
    }
    if constexpr (GROUP == 256) {
        if ((threadIdx.x & (kWave - 1)) == 0)
            for (int q = 0; q < 3; q++) scratch[q][threadIdx.x / kWave] = tot[q];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int q = 0; q < 3; q++) {
                tot[q] = 0.f;
                for (int wv = 0; wv < 256 / kWave; wv++) tot[q] += scratch[q][wv];
            }
    }
    if (tid == 0) {
        if (p.dbias) atomicAdd(p.dbias + c, tot[0]);
        if (p.dscale) atomicAdd(p.dscale + plane, tot[1] / scale);
        if (p.dnoise_gain && nz) atomicAdd(p.dnoise_gain + (plane + blockIdx.y * gridDim.x) % (unsigned)max(p.dnoise_slots, 1), tot[2]);
    }
}

template <class T>
void launch_epilogue_bwd(const sg_epilogue_bwd_params& p, hipStream_t stream) {
    const int64_t hw = (int64_t)p.h * p.w, planes = (int64_t)p.n * p.c;
    if (hw <= kSmallPlane)
        hipLaunchKernelGGL((epilogue_bwd_kernel<T, kWave>), dim3((unsigned)((planes + 256 / kWave - 1) / (256 / kWave))), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((epilogue_bwd_kernel<T, 256>), dim3((unsigned)planes, (unsigned)((hw + kSegment - 1) / kSegment)), dim3(256), 0, stream, p);
}

}  // namespace
}  // namespace sg

extern "C" int sg_epilogue_backward(const sg_epilogue_bwd_params* p, sg_stream_t stream) {
    using namespace sg;
    SG_REQUIRE(p != nullptr, SG_ERR_NULL_POINTER, "params is NULL");
    SG_REQUIRE(p->n >= 0 && p->c >= 1 && p->h >= 1 && p->w >= 1, SG_ERR_INVALID_ARGUMENT, "epilogue_backward: bad sizes");
    SG_REQUIRE((int64_t)p->n * p->c * p->h * p->w <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "epilogue_backward: tensor is too large");
    SG_REQUIRE(p->ep.act == 1 || p->ep.act == 3, SG_ERR_UNSUPPORTED, "epilogue activation must be linear (1) or lrelu (3)");
    SG_REQUIRE(!p->dscale || p->ep.scale, SG_ERR_INVALID_ARGUMENT, "epilogue_backward: dscale needs the forward's scale");
    SG_REQUIRE(p->dnoise_slots >= 0, SG_ERR_INVALID_ARGUMENT, "epilogue_backward: dnoise_slots must not be negative");
    SG_REQUIRE(!(p->dscale && p->ep.act == 3 && p->ep.alpha == 0.f), SG_ERR_UNSUPPORTED,
               "epilogue_backward: dscale with alpha = 0 (the activation is not invertible)");
    if (p->n == 0) return SG_OK;
    SG_REQUIRE(p->dy && p->y, SG_ERR_NULL_POINTER, "epilogue_backward: dy and y must not be NULL");
    switch (p->dtype) {
        case SG_F32: launch_epilogue_bwd<float>(*p, (hipStream_t)stream); break;
        case SG_F16: launch_epilogue_bwd<__half>(*p, (hipStream_t)stream); break;
        case SG_BF16: launch_epilogue_bwd<__hip_bfloat16>(*p, (hipStream_t)stream); break;
        default: return fail(SG_ERR_UNSUPPORTED, "epilogue_backward: dtype %d is not supported", p->dtype);
    }
    return check_launch("sg_epilogue_backward");
}
