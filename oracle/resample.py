"""Oracle (test infrastructure): FIR resampling and resampling convolution on CPU.

Restates, in plain differentiable PyTorch:
  * upfirdn2d                 DissimilarDomains/torch_utils/ops/upfirdn2d.py:130-219
    (kernel arithmetic: torch_utils/ops/upfirdn2d.cu:29-92, output size
    torch_utils/ops/upfirdn2d.cpp:32-33)
  * setup_filter              torch_utils/ops/upfirdn2d.py:81-125
  * filter2d/upsample2d/downsample2d   torch_utils/ops/upfirdn2d.py:292-404
  * conv2d_resample           torch_utils/ops/conv2d_resample.py:59-154

The formulation differs from the reference's fallback (one strided depthwise
correlation on the zero-stuffed, padded plane instead of pad/slice/conv/slice),
the arithmetic is the same.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def pair(v):
    if isinstance(v, int):
        return v, v
    a, b = v
    return int(a), int(b)


def quad_padding(padding):
    """-> (x_before, x_after, y_before, y_after)."""
    if isinstance(padding, int):
        return padding, padding, padding, padding
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        return padding[0], padding[0], padding[1], padding[1]
    x0, x1, y0, y1 = padding
    return x0, x1, y0, y1


def filter_extent(f):
    """-> (width, height) of a filter; None is the 1x1 identity."""
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def make_filter(taps, normalize=True, flip=False, gain=1.0, separable=None):
    """FIR filter tensor from taps (torch_utils/ops/upfirdn2d.py:81-125)."""
    f = torch.as_tensor(1 if taps is None else taps, dtype=torch.float32)
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip:
        f = f.flip(tuple(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """y = decimate(correlate(pad(zero_stuff(x, up)), flipped f) * gain, down)."""
    n, c, h, w = x.shape
    ux, uy = pair(up)
    dx, dy = pair(down)
    px0, px1, py0, py1 = quad_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)

    # Zero-stuffed plane: sample (i, j) lands on (i*uy, j*ux), trailing zeros kept.
    if ux > 1 or uy > 1:
        z = x.new_zeros([n, c, h, uy, w, ux])
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(n, c, h * uy, w * ux)

    # Positive padding adds zeros, negative padding crops.
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]

    # Separable filters are applied as two 1-D passes, each carrying sqrt(gain).
    f = f.to(x.dtype) * (gain ** (f.ndim / 2))
    if not flip_filter:
        f = f.flip(tuple(range(f.ndim)))
    if f.ndim == 2:
        k = f[None, None].expand(c, 1, *f.shape)
        return F.conv2d(x, k, stride=(dy, dx), groups=c)
    kx = f[None, None, None, :].expand(c, 1, 1, -1)
    ky = f[None, None, :, None].expand(c, 1, -1, 1)
    x = F.conv2d(x, kx, stride=(1, dx), groups=c)
    return F.conv2d(x, ky, stride=(dy, 1), groups=c)


def filter2d(x, f, padding=0, flip_filter=False, gain=1.0):
    px0, px1, py0, py1 = quad_padding(padding)
    fw, fh = filter_extent(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1.0):
    ux, uy = pair(up)
    px0, px1, py0, py1 = quad_padding(padding)
    fw, fh = filter_extent(f)
    p = [px0 + (fw + ux - 1) // 2, px1 + (fw - ux) // 2, py0 + (fh + uy - 1) // 2, py1 + (fh - uy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * ux * uy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1.0):
    dx, dy = pair(down)
    px0, px1, py0, py1 = quad_padding(padding)
    fw, fh = filter_extent(f)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """F.conv2d is a correlation (flip_weight=True); flip the taps for a true convolution."""
    if not flip_weight:
        w = w.flip([2, 3])
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False, between=None):
    """Convolution with optional FIR up/down-sampling; the six execution paths
    of torch_utils/ops/conv2d_resample.py:106-154 in the same order.
    `between`: applied to the tensor that passes from the first of a path's two operations to the second (the
    place where an implementation that keeps its tensors in 16 bits rounds)."""
    between = between or (lambda t: t)
    cout, cin_g, kh, kw = w.shape
    fw, fh = filter_extent(f)
    px0, px1, py0, py1 = quad_padding(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and down > 1 and up == 1:          # decimate, then 1x1
        x = between(upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter))
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:          # 1x1, then interpolate
        x = between(_conv(x, w, groups=groups, flip_weight=flip_weight))
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                                  # blur, then strided conv
        x = between(upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter))
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                                # transposed conv, then blur
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = between(_conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:    # plain conv
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d(x, f if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


def upfirdn2d_numpy(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """Loop-level float64 statement of the kernel arithmetic
    (torch_utils/ops/upfirdn2d.cu:42-89) for small cases; a second, independent
    check of `upfirdn2d` above and of the HIP kernels' tap indexing."""
    x = np.asarray(x, dtype=np.float64)
    n, c, h, w = x.shape
    ux, uy = pair(up)
    dx, dy = pair(down)
    px0, px1, py0, py1 = quad_padding(padding)
    f = np.ones([1, 1]) if f is None else np.asarray(f, dtype=np.float64)
    if f.ndim == 1:
        f = np.outer(f, f)
    fh, fw = f.shape
    ow = (w * ux + px0 + px1 - fw + dx) // dx
    oh = (h * uy + py0 + py1 - fh + dy) // dy
    y = np.zeros([n, c, oh, ow])
    for oy in range(oh):
        mid_y = oy * dy + uy - 1 - py0
        in_y0 = min(max(math.floor(mid_y / uy), 0), h)
        in_y1 = min(max(math.floor((mid_y + fh) / uy), 0), h)
        for ox in range(ow):
            mid_x = ox * dx + ux - 1 - px0
            in_x0 = min(max(math.floor(mid_x / ux), 0), w)
            in_x1 = min(max(math.floor((mid_x + fw) / ux), 0), w)
            acc = np.zeros([n, c])
            for iy in range(in_y0, in_y1):
                ty = mid_y + fh - (iy + 1) * uy
                ty = fh - 1 - ty if flip_filter else ty
                for ix in range(in_x0, in_x1):
                    tx = mid_x + fw - (ix + 1) * ux
                    tx = fw - 1 - tx if flip_filter else tx
                    acc += x[:, :, iy, ix] * f[ty, tx]
            y[:, :, oy, ox] = acc * gain
    return y


def grid_sample(input, grid):
    """Bilinear sampling of input [N, C, H, W] at grid [N, Ho, Wo, 2] (x, y in [-1, 1]; zero padding,
    align_corners=False), written out with gathers so that it is differentiable to any order in the image and the
    grid.  Restates what the reference's grid_sample_gradfix computes
    (DissimilarDomains/torch_utils/ops/grid_sample_gradfix.py:31-40: torch.nn.functional.grid_sample with
    mode='bilinear', padding_mode='zeros', align_corners=False)."""
    n, c, h, w = input.shape
    ix = ((grid[..., 0] + 1) * w - 1) / 2
    iy = ((grid[..., 1] + 1) * h - 1) / 2
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = 0
    flat = input.reshape(n, c, h * w)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            weight = (1 - (ix - xi).abs()) * (1 - (iy - yi).abs())              # [N, Ho, Wo]
            inside = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            index = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long().reshape(n, 1, -1).expand(n, c, -1)
            values = torch.gather(flat, 2, index).reshape(n, c, *ix.shape[1:])
            out = out + values * (weight * inside.to(weight.dtype)).unsqueeze(1)
    return out
