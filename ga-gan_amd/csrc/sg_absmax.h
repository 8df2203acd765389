// Largest magnitudes of float tensors, measured on the device in front of the kernels that multiply float
// operands as pairs of halves (sg_conv2d.hip, sg_wgrad_f32.hip): a half holds magnitudes up to 65504, so every
// operand is first multiplied by the power of two that brings its tensor's largest magnitude to 2^14 .. 2^15,
// and the accumulators are multiplied by the inverse afterwards (exact both ways).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sg {

// Measures max |a|, max |b|, max |c|, max |d| (absent tensors: n = 0) into four words of a slot of a fixed
// ring of slots in device memory and returns the slot in *slot; the words hold the float bit patterns (0 for an absent or
// all-zero tensor).  Two small launches on `stream`; nothing is allocated and the host does not wait.  The ring
// has 512 slots per device: a slot is written again 512 calls later, long after the kernel that read it has
// finished on any one stream (work recorded into a HIP graph keeps the slots it was recorded with).
int absmax_measure(hipStream_t stream, const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc,
                   const float* d, int64_t nd, const unsigned** slot);

// max |a| over n floats into word 1 of the four words at dst (the other three are zeroed; dst[4..7] are written too).
int absmax_into(hipStream_t stream, const float* a, int64_t na, unsigned* dst);

// The power of two `mul` with amax * mul in [2^14, 2^15) and its inverse; both 1 when amax is zero, denormal or
// not finite (non-finite operands poison the result either way).
__device__ __forceinline__ void half_range_scale(float amax, float& mul, float& inv) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
    int shift = (e == 0 || e == 255) ? 0 : 141 - e;
    shift = shift > 126 ? 126 : (shift < -126 ? -126 : shift);
    mul = __builtin_bit_cast(float, (unsigned)(127 + shift) << 23);
    inv = __builtin_bit_cast(float, (unsigned)(127 - shift) << 23);
}

// Scales of an operand tensor (word `tensor_word`) that is multiplied by a per-channel scale (word `scale_word`,
// ignored when `has_scale` is false) while it is staged.
__device__ __forceinline__ void half_range_scale(const unsigned* slot, int tensor_word, int scale_word, bool has_scale, float& mul,
                                                 float& inv) {
    float amax = __builtin_bit_cast(float, slot[tensor_word]);
    if (has_scale) amax *= __builtin_bit_cast(float, slot[scale_word]);
    half_range_scale(amax, mul, inv);
}

// ---- split tensors ---------------------------------------------------------------------------------------------
// The kernels that multiply float operands as pairs of halves read them from a copy made once per call in front
// of them ("split rows"): for every chunk cb of 32 channels and every row r (a pixel of a sample of an activation
// tensor, or a (tap, output channel) row of a weight tensor) 128 bytes [plane 0: 32 halves h0][plane 1: 32 halves
// h1], chunk-major: activations at byte ((n * chunks + cb) * pixels + r) * 128, weights at (cb * rows + r) * 128,
// chunks = ceil(c / 32), channels beyond c zero; v * mul = h0 + h1 (+ 2^-22 |v|), mul the tensor's half-range
// scale.  A tile of such rows goes to LDS with plain 16-byte copies of contiguous runs.
constexpr int kSplitChunk = 32;
inline int64_t split_chunks(int c) { return (c + kSplitChunk - 1) / kSplitChunk; }
inline size_t split_bytes(int64_t rows, int c) { return (size_t)rows * split_chunks(c) * 128; }

// Scratch memory of the current device for work on `stream`: grow-only, one buffer per (device, stream), kept until
// the process ends; contents are only valid between the calls of one entry point.  Growing allocates, so the
// first call of a given size must not be inside a stream capture.  nullptr (and the error message set) on failure.
void* split_workspace(hipStream_t stream, size_t bytes);

// x[n][c][pixels] float (dense NCHW) -> split rows [n][chunks][pixels], each value times scale[n][c] (when given)
// and the half-range scale of words (tensor_word, scale_word) of `slot`.
int split_pixels(hipStream_t stream, const float* x, const float* scale, int n, int c, int64_t pixels, const unsigned* slot,
                 int tensor_word, int scale_word, void* out);

// w[rows][c] float -> split rows [chunks][rows], times the half-range scale of word `tensor_word` of `slot`.
int split_rows(hipStream_t stream, const float* w, int64_t rows, int c, const unsigned* slot, int tensor_word, void* out);

}  // namespace sg
