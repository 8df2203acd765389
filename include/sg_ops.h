/*
 * sg_ops.h - C ABI of the MI355X (gfx950) kernels for the StyleGAN2 few-shot
 * training hot path.  Plain pointers and sizes only; every pointer is a device
 * pointer unless stated otherwise; every entry point enqueues on the HIP stream
 * it is given (NULL = the default stream), allocates nothing, keeps no state
 * between calls and may be called concurrently from several host threads.
 *
 * Return value: 0 on success, a negative sg_status on a rejected argument
 * (nothing was launched), or a positive hipError_t from the launch.  The text
 * for the calling thread's last failure is returned by sg_last_error().
 *
 * Citations name the interface of the reference (ndhlt/GA-GAN,
 * DissimilarDomains/) that each entry point stands in for.
 */
#ifndef SG_OPS_H
#define SG_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sg_stream_t; /* a hipStream_t */

enum sg_dtype { SG_F32 = 0, SG_F16 = 1, SG_BF16 = 2, SG_F64 = 3 };

enum sg_status {
    SG_OK = 0,
    SG_ERR_INVALID_ARGUMENT = -1, /* shape / size / enum out of range */
    SG_ERR_UNSUPPORTED = -2,      /* valid but not implemented for this dtype / layout */
    SG_ERR_NULL_POINTER = -3
};

/* Library version (major * 1000 + minor). */
int sg_version(void);

/* Message for the last non-zero return on the calling thread ("" if none). */
const char* sg_last_error(void);

/* Per-device set-up for the current device: resolves what the entry points otherwise look up on their first call
 * on a device (the device address of sg_conv2d_reduced's array).  Optional, idempotent, cheap; call it once per
 * device BEFORE recording calls into a HIP graph, because those look-ups are not stream operations and must not
 * run inside a stream capture.  No counterpart in the reference (its plugins keep no per-device state). */
int sg_init(void);

/* ---------------------------------------------------------------------------
 * upfirdn2d: per (sample, channel) plane  zero-stuff by `up`, pad (negative =
 * crop), correlate with the flipped FIR filter (not flipped when `flip` != 0),
 * keep every `down`-th sample, scale by `gain`.  Accumulation is float
 * (double for SG_F64).
 *
 * Replaces the native entry point `upfirdn2d(x, f, upx, upy, downx, downy,
 * padx0, padx1, pady0, pady1, flip, gain)` of
 * torch_utils/ops/upfirdn2d.cpp:16-94 and its parameter block
 * torch_utils/ops/upfirdn2d.h:14-40.  Strides are in elements, so
 * channels-last and sliced tensors are accepted; the filter is float32
 * [f_h, f_w] with its own strides.  The caller allocates y with
 *   out_w = (in_w * up_x + pad_x0 + pad_x1 - f_w + down_x) / down_x
 * (same for h), which sg_upfirdn2d_out_size() computes.
 */
typedef struct sg_upfirdn2d_params {
    const void* x;
    const float* f;
    void* y;
    int32_t dtype; /* sg_dtype of x and y */
    int32_t up_x, up_y, down_x, down_y;
    int32_t pad_x0, pad_y0;
    int32_t flip;
    float gain;
    int32_t in_w, in_h, in_c, in_n;
    int64_t in_stride_w, in_stride_h, in_stride_c, in_stride_n;
    int32_t f_w, f_h;
    int64_t f_stride_w, f_stride_h;
    int32_t out_w, out_h;
    int64_t out_stride_w, out_stride_h, out_stride_c, out_stride_n;
} sg_upfirdn2d_params;

int sg_upfirdn2d_out_size(int in_size, int up, int pad0, int pad1, int f_size, int down);
int sg_upfirdn2d(const sg_upfirdn2d_params* p, sg_stream_t stream);

/* y = round(upfirdn2d(x)) + addend, rounded once more: the filter followed by a separate element-wise sum, in one pass.
 * `addend` has y's dtype, shape and strides (it may be y itself).  This is the backward of a tensor that feeds both a
 * down-sampling filter and a second consumer (the input of a residual discriminator block, training/networks.py:
 * 792-808 - `self.skip(x)` beside `self.conv0(x)`): the adjoint of the filter (upfirdn2d.py:290-304) and the sum of
 * the two gradients that autograd would add in a pass of its own.  Only the up-sampling stream geometry is covered
 * (16-bit, up 2, down 1, filter up to 4x4, contiguous rows of whole aligned packets, output at least 64 pixels wide,
 * even number of output rows): ask sg_upfirdn2d_add_supported() first; anything else returns SG_ERR_UNSUPPORTED.
 * addend == NULL is sg_upfirdn2d. */
int sg_upfirdn2d_add(const sg_upfirdn2d_params* p, const void* addend, sg_stream_t stream);
int sg_upfirdn2d_add_supported(const sg_upfirdn2d_params* p);

/* Per-plane scale and its gradient, for dense tensors viewed as [planes][pixels] (plane = one channel of one sample):
 *   sg_scale_planes:  y[p][i] = x[p][i] * s[p]                         s float32 [planes]
 *   sg_dot_planes:    part[p][k] = sum over segment k of a[p][i] * b[p][i]   part float32 [planes][sg_planes_segments()]
 * (the caller adds the segments of a plane).  They stand for the broadcast multiplies of the reference's
 * modulated_conv2d - `x * styles.reshape(batch, -1, 1, 1)` and `x * dcoefs.reshape(batch, -1, 1, 1)`
 * (training/networks.py:642, :648-650) - and for the multiply-and-reduce that autograd derives from them; each is the
 * derivative of the other, so gradients of any order stay within the pair.  The scale stays float32 (the reference
 * rounds it to the tensor's dtype first).  pixels must be a multiple of 16 bytes' worth of elements, tensors 16-byte
 * aligned; SG_ERR_UNSUPPORTED otherwise. */
int sg_planes_segments(int dtype, int64_t pixels);
int sg_scale_planes(const void* x, const float* s, void* y, int dtype, int64_t planes, int64_t pixels, sg_stream_t stream);
int sg_dot_planes(const void* a, const void* b, float* part, int dtype, int64_t planes, int64_t pixels, sg_stream_t stream);

/* Packs a convolution weight w [co][ci][kh][kw] (element strides stride_o / stride_i / stride_y / stride_x: a
 * conv_transpose2d weight [ci][co][kh][kw] is passed with the first two exchanged) into the flat
 * [phase][tap row][tap column][co][ci] tensor that sg_conv2d reads (its `w`, with `phase[i].w_offset` = the element
 * offset of phase i): tap (ty, tx) of phase i is w[.][.][ky[ty]][kx[tx]].  One launch; same dtype in and out.  It stands
 * for what the reference leaves to cuDNN's own filter transformation behind conv2d / conv_transpose2d
 * (torch_utils/ops/conv2d_gradfix.py:48-61) plus the `w.flip([2, 3])` of conv2d_resample.py:28-31. */
#define SG_PACK_MAX_TAPS 8
typedef struct sg_pack_phase {
    int32_t nky, nkx;
    int32_t ky[SG_PACK_MAX_TAPS], kx[SG_PACK_MAX_TAPS];
} sg_pack_phase;
typedef struct sg_pack_params {
    const void* w;
    void* out;
    int32_t dtype;
    int32_t co, ci;
    int64_t stride_o, stride_i, stride_y, stride_x;
    int32_t num_phases;
    sg_pack_phase phase[4];
} sg_pack_params;
int sg_pack_weights(const sg_pack_params* p, sg_stream_t stream);

/* Styles and demodulation coefficients of one modulated layer from its row of the latent codes:
 *   styles[n, c] = gain * (a_gain * sum_k w[n, k] a[c, k] + b_gain * b[c])
 *   dcoefs[n, o] = rsqrt(sum_c styles[n, c]^2 * wsq_t[c, o] + eps)          (only when wsq_t and dcoefs are given)
 * All float32.  w [n, k] with w_stride elements between rows (a column of ws [n, num_ws, k] is passed as it lies);
 * a [c, k] and b [c] are the affine layer's parameters as stored, its run-time gains in a_gain / b_gain; wsq_t [c, o]
 * holds the sum over the taps of the squared convolution weights, transposed.  This is `styles = self.affine(w) *
 * weight_gain` of the reference's w_to_s (DissimilarDomains/training/networks.py:502, with FullyConnectedLayer.forward,
 * :680-700, for the gains) and the demodulation coefficients of modulated_conv2d (:631-636) written as a product with
 * sum_k W^2 instead of over the [N, O, I, k, k] tensor.  k and o must be multiples of 4, a and wsq_t 16-byte aligned. */
typedef struct sg_styles_params {
    const float* w;
    const float* a;
    const float* b;        /* may be NULL */
    const float* wsq_t;    /* may be NULL */
    float a_gain, b_gain, gain, eps;
    int32_t n, k, c, o;
    int64_t w_stride;
    float* styles;         /* [n, c] */
    float* dcoefs;         /* [n, o] or NULL */
} sg_styles_params;
int sg_styles(const sg_styles_params* p, sg_stream_t stream);

/* Gradients of sg_styles, given dstyles [n, c] and ddcoefs [n, o] (either may be NULL = zero); `f` as in the forward
 * call with styles and dcoefs holding its results.  t [n, c] receives the whole gradient with respect to the styles
 * (workspace, always written); dw [n, k], da [c, k], db [c] and dwsq_t [c, o] are written where given.  What autograd
 * derives from the reference's expressions above, in two launches. */
typedef struct sg_styles_bwd_params {
    sg_styles_params f;
    const float* dstyles;
    const float* ddcoefs;
    float* t;
    float* dw;
    float* da;
    float* db;
    float* dwsq_t;
} sg_styles_bwd_params;
int sg_styles_backward(const sg_styles_bwd_params* p, sg_stream_t stream);

/* Layer epilogue applied to a value v produced for sample n, channel c, pixel (oy, ox):
 *   clamp(act(v * scale[n,c] + noise[n,oy,ox] * *noise_gain + bias[c]) * gain)
 * scale [n, channels], noise [.., oh, ow] (noise_stride_n elements between samples, 0 = shared) and
 * *noise_gain are float32; bias [channels] has the dtype of the tensor; any pointer may be NULL.
 * act: 1 linear, 3 lrelu(alpha); clamp < 0 = off.  It is what the reference applies after a
 * resampling filter or convolution as separate passes: demodulation and noise
 * (training/networks.py:647-652) and bias_act (training/networks.py:919-921). */
typedef struct sg_epilogue {
    const float* scale;
    const float* noise;
    const float* noise_gain;
    int64_t noise_stride_n;
    const void* bias;
    int32_t act;
    float alpha, gain, clamp;
} sg_epilogue;

/* upfirdn2d followed by the epilogue in the same pass (up = down = 1, filter up to 4x4, rows
 * contiguous; other geometries return SG_ERR_UNSUPPORTED).  `ep` may be NULL (= sg_upfirdn2d). */
int sg_upfirdn2d_epilogue(const sg_upfirdn2d_params* p, const sg_epilogue* ep, sg_stream_t stream);

/* Name of the device kernel sg_upfirdn2d (ep == NULL) or sg_upfirdn2d_epilogue (ep != NULL) runs for this geometry:
 * the symbol a profiler reports, without template arguments ("blur4_stream_kernel", "down2_stream_kernel",
 * "up2_stream_kernel": the streaming kernels for 16-bit planes of whole aligned packets; "blur4_small_kernel": the
 * filter on planes of 4 x 4 to 32 x 32 outputs, several planes per workgroup; "upfirdn2d_blur4_kernel",
 * "upfirdn2d_tiled_kernel", "upfirdn2d_generic_kernel" otherwise).  Static storage.  No counterpart in the
 * reference, whose choice among its kernel specialisations (upfirdn2d.cu:205-341) is not observable. */
const char* sg_upfirdn2d_kernel(const sg_upfirdn2d_params* p, const sg_epilogue* ep);

/* Backward of the epilogue, one pass over dy and y (dense [n, c, hw] of `dtype`):
 *   g  = dy * gain * act'(.)  (zero where y was clamped; the activation's slope is taken from sign(y))
 *   dz = g * scale[n,c]                                   -> gradient w.r.t. the pre-epilogue value v
 *   dbias[c]      += sum_{n,hw} g                          (float atomics; caller zeroes)
 *   dscale[n,c]   += sum_hw g * v, with v recovered from y: v = (act^-1(y / gain) - noise * *noise_gain - bias[c]) / scale[n,c]
 *   sum_k dnoise_gain[k] += sum g * noise               (k < max(dnoise_slots, 1): a workgroup adds its share to one of
 *                                                        the slots, the gradient is their sum - thousands of float
 *                                                        atomics on ONE address run one after the other, 14 ns each)
 * (all three accumulated with float atomics: the caller zeroes them).  dz may be NULL: only the reductions are
 * computed then, for callers that hand dy and y to sg_conv2d's input mask instead of materialising dz.
 * No counterpart in the reference's native code: it is the fusion of `bias_act`'s grad=1 kernel
 * (torch_utils/ops/bias_act.cu:132-142) with the reductions that `fma`'s and `bias_act`'s Python
 * backward run as separate tensor ops (torch_utils/ops/fma.py:36-45, bias_act.py:211-212). */
typedef struct sg_epilogue_bwd_params {
    const void* dy;
    const void* y;
    void* dz;
    sg_epilogue ep;       /* the forward's epilogue */
    float* dbias;         /* [c] or NULL */
    float* dscale;        /* [n, c] or NULL (needs ep.scale) */
    float* dnoise_gain;   /* [max(dnoise_slots, 1)] or NULL (needs ep.noise) */
    int32_t dtype;
    int32_t n, c, h, w;
    int32_t dnoise_slots; /* floats at dnoise_gain to spread the additions over; 0 or 1: a single one */
} sg_epilogue_bwd_params;

int sg_epilogue_backward(const sg_epilogue_bwd_params* p, sg_stream_t stream);

/* ---------------------------------------------------------------------------
 * bias_act:  y = clamp(act(x + b[(i / step_b) % size_b]) * gain)  for grad = 0,
 * the first-order backward through the saved output `yref` (or input `xref`
 * for swish) for grad = 1, the second-order term for grad = 2.  x, xref, yref,
 * dy and y are dense arrays of size_x elements with one common layout; b,
 * xref, yref and dy may be NULL ("absent").  act = 1..9 (linear, relu, lrelu,
 * tanh, sigmoid, elu, selu, softplus, swish); clamp < 0 disables clamping.
 *
 * Replaces the native entry point `bias_act(x, b, xref, yref, dy, grad, dim,
 * act, alpha, gain, clamp)` of torch_utils/ops/bias_act.cpp:32-90 and its
 * parameter block torch_utils/ops/bias_act.h:12-31 (`dim` is folded into
 * step_b = stride of that dimension, as bias_act.cpp:73 does).
 */
typedef struct sg_bias_act_params {
    const void* x;
    const void* b;
    const void* xref;
    const void* yref;
    const void* dy;
    void* y;
    int32_t dtype;
    int32_t grad;
    int32_t act;
    float alpha, gain, clamp;
    int64_t size_x;
    int32_t size_b;
    int64_t step_b;
} sg_bias_act_params;

int sg_bias_act(const sg_bias_act_params* p, sg_stream_t stream);

/* ---------------------------------------------------------------------------
 * Minibatch standard deviation (training/networks.py:1284-1301, a composite of
 * seven tensor ops in the reference; no native counterpart there).
 * x is dense [n, c, h, w]; y is dense [n, c + f, h, w].  The batch is split
 * into n / g groups of g samples (sample i belongs to group i % (n / g)) and
 * the channels into f sets of c / f; channel c + j of every sample of a group
 * holds mean over (set j, h, w) of sqrt(var over the group + 1e-8).
 * Forward copies x into y[:, :c] as well.  Backward takes dy [n, c + f, h, w]
 * and writes dx [n, c, h, w] = dy[:, :c] + d(stat)/dx contribution.
 */
typedef struct sg_mbstd_params {
    const void* x;  /* [n, c, h, w] */
    const void* dy; /* backward only: [n, c + f, h, w] */
    void* out;      /* forward: y [n, c + f, h, w]; backward: dx [n, c, h, w] */
    int32_t dtype;
    int32_t n, c, hw; /* hw = h * w */
    int32_t g, f;
} sg_mbstd_params;

int sg_mbstd_forward(const sg_mbstd_params* p, sg_stream_t stream);
int sg_mbstd_backward(const sg_mbstd_params* p, sg_stream_t stream);

/* ---------------------------------------------------------------------------
 * conv2d: implicit-GEMM convolution on the matrix cores with the StyleGAN2 layer
 * fused around it,
 *   y[n,co,oy,ox] = clamp(act((sum_{ci,tap} W[phase][tap][co][ci] * x[n,ci,iy,ix] * in_scale[n,ci])
 *                             * out_scale[n,co] + noise[n,oy,ox] * *noise_gain + bias[co]) * gain)
 * Output pixel (oy, ox) = (oy0 + a * out_step, ox0 + b * out_step) of phase `ph` reads
 * input pixels (iy, ix) = (a * in_step + dy0 + ky, b * in_step + dx0 + kx), ky < nky, kx < nkx;
 * reads outside the input are zero.  One phase with in_step = out_step = 1 is a plain
 * correlation (dy0 = dx0 = -padding); in_step = 2 is a strided convolution; out_step = 2 with
 * up to four phases is a transposed (up-sampling) convolution in polyphase form.
 *
 * Stands in for what the reference composes per layer from tensor ops:
 * `modulated_conv2d` (training/networks.py:591-668: x * styles, conv2d_resample,
 * fma with the demodulation coefficients and noise), `bias_act` after it
 * (training/networks.py:919-921, 757-759) and the library convolution called by
 * torch_utils/ops/conv2d_gradfix.py:43-58.  x, y: dense NCHW of `dtype` (f32, f16, bf16);
 * w: `dtype`, packed [phase][ky][kx][co][ci] (w_offset = element offset of the phase);
 * in_scale [n, ci], out_scale [n, co], noise [.., oh, ow] (stride noise_stride_n between
 * samples, 0 = shared) and *noise_gain are float32; bias [co] is `dtype`.  Any of
 * in_scale, out_scale, noise, noise_gain, bias may be NULL.  act: 1 linear, 3 lrelu(alpha).
 */
typedef struct sg_conv2d_phase {
    int32_t oy0, ox0; /* first output pixel of the phase (0 .. out_step-1) */
    int32_t dy0, dx0; /* input offset of tap (0, 0) */
    int32_t nky, nkx; /* taps */
    int64_t w_offset; /* element offset of this phase's [nky][nkx][co][ci] block in w */
} sg_conv2d_phase;

typedef struct sg_conv2d_params {
    const void* x;
    const void* w;
    void* y;
    const float* in_scale;
    const float* out_scale;
    const void* bias;
    const float* noise;
    const float* noise_gain;
    int64_t noise_stride_n;
    int32_t dtype;
    int32_t n, ci, h, w_in, co, oh, ow;
    int32_t in_step, out_step;
    int32_t num_phases;
    sg_conv2d_phase phase[4];
    int32_t act;
    float alpha, gain, clamp;
    /* optional reduction of acc[n,co,oy,ox] * reduce_x[n,co,oy,ox] over the pixels, with acc the contraction before
     * out_scale and the epilogue (reduce_x has y's shape and dtype).  reduce_out is float32 [n, rows, co], zeroed by
     * the caller, rows = sg_conv2d_reduce_rows(p); the kernel leaves one partial sum per row (no contended atomics)
     * and the caller adds the rows.  With the data-gradient call this yields the gradient w.r.t. the modulation
     * styles in the same pass. */
    const void* reduce_x;
    float* reduce_out;
    /* optional input mask, for the call that computes the data gradient of a layer whose epilogue was fused: x is the
     * gradient dy that arrived at the layer's output and mask_y the output the forward pass saved (x's shape and
     * dtype).  The kernel then contracts x * m, m = (mask_y > 0 ? 1 : mask_alpha), m = 0 where |mask_y| >= mask_clamp
     * (mask_clamp < 0: no bound): `bias_act`'s first-order backward (torch_utils/ops/bias_act.cu:132-142, without its
     * gain, which the caller folds into `gain`) applied while the operand is staged.  NULL = no mask.  Only some
     * kernels implement it: ask sg_conv2d_supports_mask() first; an unsupported call returns SG_ERR_UNSUPPORTED. */
    const void* mask_y;
    float mask_alpha, mask_clamp;
    /* scratch memory in device memory, 256-byte aligned, of at least sg_conv2d_workspace_bytes(p) bytes; its contents
     * mean nothing before or after the call.  Only float32 calls need any (they multiply copies of both operands cut
     * into pairs of halves).  NULL with a non-zero requirement: the library uses a buffer of its own per device and
     * stream, which it grows with hipMalloc when a call needs more - so such a call must not be recorded into a HIP
     * graph, and the buffer stays allocated until the process ends. */
    void* workspace;
    size_t workspace_bytes;
    /* float32 calls only, may be NULL: the weights as sg_conv2d_prepare_weights left them for this geometry (their
     * largest magnitude and their split rows).  The call then neither measures nor splits `w` again - for weights that
     * do not change between calls (frozen layers; any layer between two optimizer steps) that is two launches and a
     * pass over the weight tensor saved per call - and p->workspace needs room for the input's rows only. */
    const void* w_prepared;
    /* may be NULL: a tensor of y's shape and dtype that is added to the result after the epilogue, y = epilogue(...) +
     * addend - the sum of a residual block (training/networks.py:1268-1271 `x = y.add_(x)`) formed while the skip
     * branch is stored instead of in a pass of its own; or the other gradient of a tensor that feeds a to-RGB layer
     * and the next block (training/networks.py:1041-1046), added by the to-RGB layer's data-gradient call.  Only where
     * sg_conv2d_supports_addend(p) returns 1 (16-bit 1x1 layers with at least 32 input and 16 output channels and no
     * modulation; 16-bit 1x1 layers with 3 input channels and a multiple of 4 output channels); other calls return
     * SG_ERR_UNSUPPORTED.  The sum is rounded as a separate pass would: round(round(result) + addend). */
    const void* addend;
} sg_conv2d_params;

int sg_conv2d(const sg_conv2d_params* p, sg_stream_t stream);

/* Bytes of p->workspace the call needs (0: none).  Depends on dtype, n, ci, h, w_in, co, the phases and on whether
 * w_prepared is given. */
size_t sg_conv2d_workspace_bytes(const sg_conv2d_params* p);

/* Prepared weights of a float32 call (see sg_conv2d_params.w_prepared): size of the buffer (0 for other dtypes), and
 * the two launches that fill it from p->w on `stream`.  `out` is device memory, 256-byte aligned; it depends on p->w,
 * ci, co and the phases only and may be kept as long as the weight values are.  No counterpart in the reference
 * (cuDNN transforms filters inside its own calls). */
int sg_conv2d_supports_addend(const sg_conv2d_params* p);
size_t sg_conv2d_prepared_weight_bytes(const sg_conv2d_params* p);
int sg_conv2d_prepare_weights(const sg_conv2d_params* p, void* out, size_t out_bytes, sg_stream_t stream);

/* 1 when sg_conv2d honours p->mask_y for this geometry (decided as if mask_y were set), else 0. */
int sg_conv2d_supports_mask(const sg_conv2d_params* p);

/* Number of partial rows per sample the reduction above writes for this geometry (>= 1), -1 on a bad argument. */
int sg_conv2d_reduce_rows(const sg_conv2d_params* p);

/* sg_conv2d with a reduction (p->reduce_x set) for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The
 * partial rows go to a fixed array inside the library (SG_CONV2D_REDUCED_MAX_PARTIALS floats, one copy per
 * device, held until the process ends) instead of a caller-provided workspace, and are added up on the device:
 *   reduce_sum[n, co] (float32, written, not accumulated) = sum over pixels of reduce_x * contraction.
 * p->reduce_out must be NULL.  The call allocates and frees nothing; it is rejected with SG_ERR_UNSUPPORTED when
 * n * sg_conv2d_reduce_rows(p) * co exceeds the array, and with SG_ERR_INVALID_ARGUMENT when a float32 call comes
 * without p->workspace (sg_conv2d_workspace_bytes(p) bytes: the scratch of the float32 kernels is the caller's here).  Every call on a device shares that device's array: calls
 * whose execution may overlap on one device (different streams) race; calls on one stream are safe.  The array is
 * cleared by a kernel of the call itself, so a recorded call clears it on every replay of its graph; call sg_init()
 * (or make one eager call) on the device before recording, so that the array's address is already known. */
#define SG_CONV2D_REDUCED_MAX_PARTIALS (1 << 20)
int sg_conv2d_reduced(const sg_conv2d_params* p, float* reduce_sum, sg_stream_t stream);

/* Name of the device kernel sg_conv2d runs for this geometry (the symbol a profiler reports, without template
 * arguments).  For 16-bit tensors whose rows are whole aligned packets: "conv3x3_kernel" (3x3, output step 1,
 * input step 1 or 2), "conv3x3_resident_kernel" (the same with at most 64 channels on both sides: the weights
 * stay in LDS), "conv_up2_kernel" (3x3 transposed, stride 2, no epilogue), "band3x3_kernel" (3x3 same-size, 32 input
 * and 32 or 64 output channels: persistent row bands with the weights in registers), "conv1x1_few_kernel" (1x1 with at
 * most 4 channels on one side, planes of at least 4096 pixels: streaming kernels on the vector units),
 * "pointwise_kernel" (1x1 with 32 or more channels on both sides, no modulation, noise, reduction or mask: the
 * residual skips of the discriminator); "conv2d_mfma_kernel" otherwise.
 * Static storage. */
const char* sg_conv2d_kernel(const sg_conv2d_params* p);

/* ---------------------------------------------------------------------------
 * conv_up2_fir: an up-sampling 3x3 layer in one pass, with no intermediate tensor:
 *   t = conv_transpose2d(x * in_scale, W, stride 2)            [n, co, 2h+1, 2w+1]
 *   u = upfirdn2d(t, f, padding 1 on every side, gain 4)       [n, co, 2h, 2w],  f = fy (x) fx separable, 4 taps
 *   y = epilogue(u)                                            (sg_epilogue: demodulation, noise, bias, act, gain, clamp)
 * i.e. the `up=2` branch of torch_utils/ops/conv2d_resample.py:125-142 (transposed convolution, then the resampling
 * filter with gain up^2) as called by `modulated_conv2d` (training/networks.py:640-653), with the `bias_act` that
 * follows it (training/networks.py:919-921).  The horizontal filter is folded into the weights by the caller:
 *   w [3][6][co][ci] (dtype), w[r][c] = sum over (kx, j) with kx - j + 1 == D[c] of 2 * fx[3 - j] * W[co, ci, KY[r], kx],
 *   KY = {0, 2, 1}, D = {-2, 0, 2, -1, 1, 3}                  (W[co, ci, ky, kx]: t[2a+ky, 2b+kx] += x[a, b] * W)
 * and the vertical one runs on the accumulators: fy4 = device array of the 4 values 2 * fy[3 - i], i = 0..3, that
 * multiply rows oy - 1 + i of t for output row oy.  x [n, ci, h, w_in] and y [n, co, 2h, 2w_in] are dense NCHW of
 * `dtype` (f16 or bf16); in_scale [n, ci] float32 or NULL.  Geometries outside sg_conv_up2_fir_supported()
 * (16-byte aligned tensors, ci % 8 == 0, ci >= 32, w_in % 8 == 0, w_in >= 32, h >= 2, gain > 0, 0 <= alpha <= 1)
 * return SG_ERR_UNSUPPORTED: compose the layer from sg_conv2d and sg_upfirdn2d_epilogue instead.
 */
typedef struct sg_conv_up2_fir_params {
    const void* x;
    const void* w;
    void* y;
    const float* in_scale;
    const float* fy4;
    sg_epilogue ep;
    int32_t dtype;
    int32_t n, ci, h, w_in, co;
} sg_conv_up2_fir_params;

int sg_conv_up2_fir(const sg_conv_up2_fir_params* p, sg_stream_t stream);
int sg_conv_up2_fir_supported(const sg_conv_up2_fir_params* p);

/* ---------------------------------------------------------------------------
 * conv2d_wgrad: weight gradient of the convolution above (one phase),
 *   dw[ky][kx][cg][cx] += sum_{n,a,b} g[n,cg,a,b] * g_scale[n,cg]
 *                                     * x[n,cx,a*step+dy0+ky,b*step+dx0+kx] * x_scale[n,cx]
 * x [n, cx, h, w_in] and g [n, cg, gh, gw] are dense NCHW of `dtype`; dw is float32,
 * packed [nky][nkx][cg][cx], and is ACCUMULATED into with float atomics (the caller
 * zeroes it).  For a forward y = conv(x, W) pass g = dy (cg = co, cx = ci); for a
 * transposed convolution exchange the roles of x and dy.  x_scale [n, cx] and
 * g_scale [n, cg] (float32) may be NULL.
 *
 * Stands in for the library weight-gradient call of
 * torch_utils/ops/conv2d_gradfix.py:175-191 (`cudnn_convolution_backward_weight`).
 */
typedef struct sg_conv2d_wgrad_params {
    const void* x;
    const void* g;
    float* dw;
    const float* x_scale;
    const float* g_scale;
    int32_t dtype;
    int32_t n, cx, h, w_in, cg, gh, gw;
    int32_t step, dy0, dx0, nky, nkx;
    /* optional mask on the gradient operand g, as sg_conv2d_params.mask_y: g is the gradient that arrived at the output
     * of a layer with a fused epilogue, mask_y that output as saved by the forward pass (g's shape and dtype); the
     * kernel contracts g * m.  NULL = none.  Ask sg_conv2d_wgrad_supports_mask() first. */
    const void* mask_y;
    float mask_alpha, mask_clamp;
    /* scratch memory as in sg_conv2d_params, at least sg_conv2d_wgrad_workspace_bytes(p) bytes */
    void* workspace;
    size_t workspace_bytes;
} sg_conv2d_wgrad_params;

int sg_conv2d_wgrad(const sg_conv2d_wgrad_params* p, sg_stream_t stream);

/* Bytes of p->workspace the call needs (0: none). */
size_t sg_conv2d_wgrad_workspace_bytes(const sg_conv2d_wgrad_params* p);

/* 1 when sg_conv2d_wgrad honours p->mask_y for this geometry (decided as if mask_y were set), else 0. */
int sg_conv2d_wgrad_supports_mask(const sg_conv2d_wgrad_params* p);

/* Name of the device kernel sg_conv2d_wgrad runs for this geometry: "wgrad1x1_few_kernel" (16-bit, 1x1, at most 4
 * channels on one side and at least 8 on the other, planes of at least 4096 pixels), "wgrad3x3_kernel" (16-bit, 3x3, step 1 or 2,
 * at least 32 channels on both sides, rows of whole packets, gradient planes 32 pixels wide or more),
 * "wgrad_f32_kernel" (float32, 1x1 or 3x3, at least 32 channels on both sides) or "conv2d_wgrad_kernel".  Static storage. */
const char* sg_conv2d_wgrad_kernel(const sg_conv2d_wgrad_params* p);

/* ---------------------------------------------------------------------------
 * Optimizer step over a list of float32 tensors in one launch.  For every tensor:
 *   g    = grad with NaN -> 0, +Inf -> +grad_limit, -Inf -> -grad_limit   (written back to grad)
 *   m    = m + (g - m) * (1 - beta1);   v = v * beta2 + g * g * (1 - beta2)
 *   p    = p - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * i.e. `param.grad.nan_to_num(0, 1e5, -1e5)` followed by `torch.optim.Adam.step()` without weight decay or
 * amsgrad, what the reference's loop runs per phase (training/training_loop.py:505-510 with the optimizer of
 * :303-312; a composite of ~20 list operations plus one kernel per gradient there).  `step` is the number of
 * updates the tensor has received including this one.  The table is host memory and is consumed before the call
 * returns; at most SG_OPTIM_MAX_TENSORS tensors per call (the caller cuts longer lists).
 */
#define SG_OPTIM_MAX_TENSORS 64

typedef struct sg_adam_tensor {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    int32_t step;
} sg_adam_tensor;

typedef struct sg_adam_params {
    const sg_adam_tensor* tensors; /* host array [count] */
    int32_t count;
    float lr, beta1, beta2, eps;
    float grad_limit; /* 1e5 in the reference */
} sg_adam_params;

int sg_adam_step(const sg_adam_params* p, sg_stream_t stream);

/* dst = src + beta * (dst - src) over a list of float32 tensors in one launch: the moving average of the generator
 * weights, `p_ema.copy_(p.lerp(p_ema, ema_beta))` with dst = p_ema, src = p (training/training_loop.py:516-521). */
typedef struct sg_lerp_tensor {
    float* dst;
    const float* src;
    int64_t numel;
} sg_lerp_tensor;

typedef struct sg_lerp_params {
    const sg_lerp_tensor* tensors; /* host array [count] */
    int32_t count;
    float beta;
} sg_lerp_params;

int sg_lerp_multi(const sg_lerp_params* p, sg_stream_t stream);

/* ---------------------------------------------------------------------------
 * A batch of training images from a uint8 image set resident in device memory:
 *   out[b, c, y, x] = (float)images[index[b], c, y, flip[b] ? w - 1 - x : x] / 127.5f - 1.0f
 * (float32 arithmetic with a correctly rounded division, then rounded once to `dtype`).  It replaces the host-side
 * gather of the reference's loader (`Dataset.__getitem__` with its left-right mirror, training/dataset.py:85-93,
 * batched by the DataLoader of training/training_loop.py:221-223) plus the conversion on the device
 * (`phase_real_img.to(device).to(torch.float32) / 127.5 - 1`, training/training_loop.py:441) by one pass that reads
 * one byte and writes one element per pixel.  `index` and `flip` are device arrays of n entries; either may be NULL
 * (identity order / no mirror).  An index outside [0, count) yields NaN for that sample instead of a wild read.
 */
typedef struct sg_images_params {
    const uint8_t* images;  /* [count, c, h, w] dense */
    const int64_t* index;   /* device [n] or NULL */
    const uint8_t* flip;    /* device [n] or NULL */
    void* out;              /* [n, c, h, w] dense, element type dtype (SG_F32, SG_F16 or SG_BF16) */
    int32_t dtype;
    int32_t n, c, h, w;
    int64_t count;
} sg_images_params;

int sg_images_to_float(const sg_images_params* p, sg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SG_OPS_H */
