// Two streaming primitives over [planes][pixels] tensors (a plane = one channel of one sample):
//   sg_scale_planes:  y[p][i] = x[p][i] * s[p]                      (modulation by styles, demodulation by coefficients)
//   sg_dot_planes:    part[p][k] = sum over segment k of a[p][i] * b[p][i]   (the gradient of such a scale)
// Each is the derivative of the other, so the pair is closed under differentiation: the layers that the regularisation
// passes evaluate from differentiable primitives use them instead of a broadcast multiply (one generic element-wise
// kernel) and a multiply followed by a reduction (two passes, one of them writing a full-size product).
// 16-byte packets per lane, float arithmetic; a plane is cut into segments so that small batches still fill the chip.
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kThreads = 256;
constexpr int kSegPackets = 2048;     // packets (16 bytes) per workgroup: 8 per lane

template <class T> struct Unpack {
    static constexpr int N = 16 / (int)sizeof(T);
    static __device__ __forceinline__ void load(const T* p, float (&f)[N]) {
        const Pack<T> v = load_pack(p);
#pragma unroll
        for (int j = 0; j < N; j++) f[j] = Elem<T>::load(v.v[j]);
    }
    static __device__ __forceinline__ void store(T* p, const float (&f)[N]) {
        Pack<T> v;
#pragma unroll
        for (int j = 0; j < N; j++) v.v[j] = Elem<T>::store(f[j]);
        store_pack(p, v);
    }
};

template <class T>
__global__ void __launch_bounds__(kThreads) scale_planes_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                                 T* __restrict__ y, int64_t packets, int segs) {
    constexpr int N = Unpack<T>::N;
    const int64_t plane = blockIdx.x / segs;
    const int seg = blockIdx.x - (int)(plane * segs);
    const float sc = s[plane];
    const int64_t p0 = (int64_t)seg * kSegPackets, p1 = (p0 + kSegPackets < packets) ? p0 + kSegPackets : packets;
    const T* xp = x + plane * packets * N;
    T* yp = y + plane * packets * N;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kThreads) {
        float f[N];
        Unpack<T>::load(xp + p * N, f);
#pragma unroll
        for (int j = 0; j < N; j++) f[j] *= sc;
        Unpack<T>::store(yp + p * N, f);
    }
}

template <class T>
__global__ void __launch_bounds__(kThreads) dot_planes_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                               float* __restrict__ part, int64_t packets, int segs) {
    constexpr int N = Unpack<T>::N;
    __shared__ float s_sum[kThreads / 64];
    const int64_t plane = blockIdx.x / segs;
    const int seg = blockIdx.x - (int)(plane * segs);
    const int64_t p0 = (int64_t)seg * kSegPackets, p1 = (p0 + kSegPackets < packets) ? p0 + kSegPackets : packets;
    const T* ap = a + plane * packets * N;
    const T* bp = b + plane * packets * N;
    float acc = 0.f;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kThreads) {
        float fa[N], fb[N];
        Unpack<T>::load(ap + p * N, fa);
        Unpack<T>::load(bp + p * N, fb);
#pragma unroll
        for (int j = 0; j < N; j++) acc = __builtin_fmaf(fa[j], fb[j], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < kThreads / 64; w++) r += s_sum[w];
        part[plane * segs + seg] = r;
    }
}

int check_planes(const char* what, const void* x, const void* y, int dtype, int64_t planes, int64_t pixels) {
    SG_REQUIRE(dtype == SG_F32 || dtype == SG_F16 || dtype == SG_BF16, SG_ERR_UNSUPPORTED, "%s: dtype %d (f32, f16, bf16)", what, dtype);
    SG_REQUIRE(planes >= 0 && pixels >= 0, SG_ERR_INVALID_ARGUMENT, "%s: negative size", what);
    const int per = dtype == SG_F32 ? 4 : 8;
    SG_REQUIRE(pixels % per == 0, SG_ERR_UNSUPPORTED, "%s: planes of %lld pixels are no whole 16-byte packets", what, (long long)pixels);
    SG_REQUIRE(planes * (pixels / per + 1) <= INT32_MAX, SG_ERR_INVALID_ARGUMENT, "%s: tensor is too large", what);
    SG_REQUIRE(aligned16(x) && aligned16(y), SG_ERR_INVALID_ARGUMENT, "%s: tensors must be 16-byte aligned", what);
    return SG_OK;
}

}  // namespace
}  // namespace sg

extern "C" int sg_planes_segments(int dtype, int64_t pixels) {
    const int per = dtype == SG_F32 ? 4 : 8;
    if (pixels <= 0) return 1;
    return (int)((pixels / per + sg::kSegPackets - 1) / sg::kSegPackets);
}

extern "C" int sg_scale_planes(const void* x, const float* s, void* y, int dtype, int64_t planes, int64_t pixels, sg_stream_t stream) {
    using namespace sg;
    if (int rc = check_planes("scale_planes", x, y, dtype, planes, pixels)) return rc;
    if (planes == 0 || pixels == 0) return SG_OK;
    SG_REQUIRE(x && s && y, SG_ERR_NULL_POINTER, "scale_planes: x, s and y must not be NULL");
    const int segs = sg_planes_segments(dtype, pixels);
    const dim3 grid((unsigned)(planes * segs));
    switch (dtype) {
        case SG_F32: hipLaunchKernelGGL(scale_planes_kernel<float>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, s, (float*)y, pixels / 4, segs); break;
        case SG_F16: hipLaunchKernelGGL(scale_planes_kernel<__half>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const __half*)x, s, (__half*)y, pixels / 8, segs); break;
        default: hipLaunchKernelGGL(scale_planes_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const __hip_bfloat16*)x, s, (__hip_bfloat16*)y, pixels / 8, segs); break;
    }
    return check_launch("sg_scale_planes");
}

extern "C" int sg_dot_planes(const void* a, const void* b, float* part, int dtype, int64_t planes, int64_t pixels, sg_stream_t stream) {
    using namespace sg;
    if (int rc = check_planes("dot_planes", a, b, dtype, planes, pixels)) return rc;
    if (planes == 0) return SG_OK;
    SG_REQUIRE(a && b && part, SG_ERR_NULL_POINTER, "dot_planes: a, b and part must not be NULL");
    const int segs = sg_planes_segments(dtype, pixels);
    const dim3 grid((unsigned)(planes * segs));
    switch (dtype) {
        case SG_F32: hipLaunchKernelGGL(dot_planes_kernel<float>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)a, (const float*)b, part, pixels / 4, segs); break;
        case SG_F16: hipLaunchKernelGGL(dot_planes_kernel<__half>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const __half*)a, (const __half*)b, part, pixels / 8, segs); break;
        default: hipLaunchKernelGGL(dot_planes_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const __hip_bfloat16*)a, (const __hip_bfloat16*)b, part, pixels / 8, segs); break;
    }
    return check_launch("sg_dot_planes");
}
