// Weight packing: w [co][ci][kh][kw] (any element strides: a transposed view is passed by exchanging the first two) ->
// the flat [phase][ky][kx][co][ci] tensor the convolution kernels read, in one launch.  The framework's operators
// take a slice, a flip, a permuting copy per phase and a concatenation for it: 3-12 launches per weight and version.
#include "sg_common.h"

#include <algorithm>

namespace sg {
namespace {

template <class E>
__global__ void __launch_bounds__(256) pack_weights_kernel(sg_pack_params p, int64_t total) {
    const E* __restrict__ w = (const E*)p.w;
    E* __restrict__ out = (E*)p.out;
    const int64_t per_tap = (int64_t)p.co * p.ci;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        int ph = 0;
        while (ph + 1 < p.num_phases && r >= (int64_t)p.phase[ph].nky * p.phase[ph].nkx * per_tap) {
            r -= (int64_t)p.phase[ph].nky * p.phase[ph].nkx * per_tap;
            ph++;
        }
        const int tap = (int)(r / per_tap);
        const int64_t oi = r - (int64_t)tap * per_tap;
        const int o = (int)(oi / p.ci), i = (int)(oi - (int64_t)o * p.ci);
        const int ty = tap / p.phase[ph].nkx, tx = tap - ty * p.phase[ph].nkx;
        out[e] = w[o * p.stride_o + i * p.stride_i + p.phase[ph].ky[ty] * p.stride_y + p.phase[ph].kx[tx] * p.stride_x];
    }
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" int sg_pack_weights(const sg_pack_params* p, sg_stream_t stream) {
    SG_REQUIRE(p->w && p->out && p->co > 0 && p->ci > 0, SG_ERR_INVALID_ARGUMENT, "pack_weights: w, out and the channel counts must be given");
    SG_REQUIRE(p->num_phases >= 1 && p->num_phases <= 4, SG_ERR_INVALID_ARGUMENT, "pack_weights: %d phases", p->num_phases);
    int64_t total = 0;
    for (int i = 0; i < p->num_phases; i++) {
        const auto& ph = p->phase[i];
        SG_REQUIRE(ph.nky >= 1 && ph.nky <= SG_PACK_MAX_TAPS && ph.nkx >= 1 && ph.nkx <= SG_PACK_MAX_TAPS, SG_ERR_UNSUPPORTED,
                   "pack_weights: %d x %d taps in a phase (at most %d per axis)", ph.nky, ph.nkx, SG_PACK_MAX_TAPS);
        total += (int64_t)ph.nky * ph.nkx * p->co * p->ci;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    switch (p->dtype) {
        case SG_F32: hipLaunchKernelGGL(pack_weights_kernel<unsigned>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, total); break;
        case SG_F16:
        case SG_BF16: hipLaunchKernelGGL(pack_weights_kernel<unsigned short>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, total); break;
        default: return fail(SG_ERR_UNSUPPORTED, "pack_weights: dtype %d is not supported", p->dtype);
    }
    return check_launch("sg_pack_weights");
}
