// Internal interface between the upfirdn2d entry points and the streaming blur kernel (sg_blur_stream.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "sg_ops.h"

namespace sg {

// up = down = 1, filter of at most 4 x 4 taps, 16-bit tensors with rows of whole aligned packets, planes at least 64
// pixels wide; `ep` (may be NULL) is the layer epilogue applied to the filtered value.
bool blur_stream_eligible(const sg_upfirdn2d_params& p, const sg_epilogue* ep);
int blur_stream_launch(const sg_upfirdn2d_params& p, const sg_epilogue* ep, hipStream_t stream);

}  // namespace sg
