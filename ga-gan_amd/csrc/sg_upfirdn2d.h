// Internal interface between the upfirdn2d entry points and the streaming blur kernel (sg_blur_stream.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "sg_ops.h"

namespace sg {

// up = down = 1, filter of at most 4 x 4 taps, 16-bit tensors with rows of whole aligned packets, planes at least 64
// pixels wide; `ep` (may be NULL) is the layer epilogue applied to the filtered value.
bool blur_stream_eligible(const sg_upfirdn2d_params& p, const sg_epilogue* ep);
int blur_stream_launch(const sg_upfirdn2d_params& p, const sg_epilogue* ep, hipStream_t stream);

// up = 1, down = 2 under the same conditions (the filter in front of the discriminator's 1x1 skip convolutions)
bool down2_stream_eligible(const sg_upfirdn2d_params& p);
int down2_stream_launch(const sg_upfirdn2d_params& p, hipStream_t stream);

// up = 2, down = 1 under the same conditions, output rows of whole pairs of packets (the adjoint of the above)
bool up2_stream_eligible(const sg_upfirdn2d_params& p);
// `addend` (may be NULL): a tensor with y's dtype and strides that is added to the filtered result
int up2_stream_launch(const sg_upfirdn2d_params& p, const void* addend, hipStream_t stream);

}  // namespace sg
