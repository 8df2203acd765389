// Internal interface between the general convolution entry points and the 3x3 unit-stride kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "sg_ops.h"

namespace sg {

struct Conv3x3Args {
    const void* x; const void* w; void* y;
    const float* in_scale; const float* out_scale; const void* bias;
    const float* noise; const float* noise_gain; const void* reduce_x; float* reduce_out;
    int n, ci, co, h, wid, oh, ow;   // wid: input width
    int dy0;             // input row of output row a, tap ky: a + dy0 + ky
    int px_shift;        // pixel column (relative to the tile's first input column b0 * step) where the staged packets start
    int px_col;          // column of the footprint's first pixel inside the staged image, 0..7
    int tiles_a, tiles_b;
    int units;           // tiles x output-channel blocks
    int noise_stride;
    float slope, gain, hi;   // leaky-ReLU slope (1: linear), output gain, clamp bound (+inf: none)
};

bool conv3x3_eligible(const sg_conv2d_params& p);
int conv3x3_launch(const sg_conv2d_params& p, hipStream_t stream);
bool conv3x3_resident(const sg_conv2d_params& p);       // eligible and narrow (<= 64 channels both sides): weights stay in LDS
int conv3x3_reduce_rows(const sg_conv2d_params& p);   // partial rows per sample of the reduction, for eligible geometries

int compute_units();   // of the current device

// Same-size 3x3 convolution with 32 / 64 channels on both sides as persistent row bands (sg_band3x3.hip).
bool band3x3_eligible(const sg_conv2d_params& p);
int band3x3_launch(const sg_conv2d_params& p, hipStream_t stream);
int band3x3_reduce_rows(const sg_conv2d_params& p);

// 1x1 convolutions with at most 4 channels on one side as streaming kernels, and their weight gradient (sg_conv1x1_few.hip).
bool conv1x1_few_eligible(const sg_conv2d_params& p);
int conv1x1_few_launch(const sg_conv2d_params& p, hipStream_t stream);
int conv1x1_few_reduce_rows(const sg_conv2d_params& p);
bool wgrad1x1_few_eligible(const sg_conv2d_wgrad_params& p);
int wgrad1x1_few_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream);

// 1x1 convolutions with 32 or more channels on both sides and no modulation, as a streaming matrix-core kernel (sg_pointwise.hip).
bool pointwise_eligible(const sg_conv2d_params& p);
int pointwise_launch(const sg_conv2d_params& p, hipStream_t stream);

// Transposed 3x3 stride-2 convolution as one unit-stride contraction over the 2 x 2 input window (sg_convup.hip).
bool conv_up2_eligible(const sg_conv2d_params& p);
int conv_up2_launch(const sg_conv2d_params& p, hipStream_t stream);

// Weight gradient of the same geometry (sg_wgrad3x3.hip); sg_conv2d_wgrad dispatches.
bool wgrad3x3_eligible(const sg_conv2d_wgrad_params& p);
int wgrad3x3_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream);

// Weight gradient of same-size 3x3 layers with 32 channels on both sides, with the activation mask (sg_wgrad_rows.hip).
bool wgrad1x1_eligible(const sg_conv2d_wgrad_params& p);    // 16-bit 1x1 layers with 32 or more channels on both sides
int wgrad1x1_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream);
bool wgrad_rows_eligible(const sg_conv2d_wgrad_params& p);
int wgrad_rows_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream);

// Weight gradient of float32 layers with at least 32 channels on both sides (sg_wgrad_f32.hip).
bool wgrad_f32_eligible(const sg_conv2d_wgrad_params& p);
int wgrad_f32_launch(const sg_conv2d_wgrad_params& p, hipStream_t stream);

}  // namespace sg
