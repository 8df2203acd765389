"""Adaptive discriminator augmentation pipeline on the gfx950 op API: the step in front of
every discriminator call when training with `--aug=ada` (reference: training/augment.py:184-520,
same constructor arguments, same buffers `p`, `Hz_geom`, `Hz_fbank`, same `forward(images,
debug_percentile)`).

Every augmentation is applied to each image with probability `<its multiplier> * p`:
  * pixel blitting        x-flip, 90-degree rotations, integer translation;
  * geometric transforms  isotropic / anisotropic scaling, rotation before and after the anisotropic
                          scaling, fractional translation;
  * colour transforms     brightness, contrast, luma flip, hue rotation, saturation;
  * image-space filtering per-band amplification with a 4-band filter bank;
  * corruptions           additive noise, cutout.
The geometric ones compose into ONE inverse homogeneous transform per image, executed once: reflect-pad,
2x up-sample with the 12-tap `sym6` low-pass (upfirdn2d), bilinear resampling on the fine grid
(grid_sample_gradfix, differentiable to any order for R1), low-pass + 2x decimation + crop (upfirdn2d
with negative padding and flipped filter).  The colour ones compose into one 4 x 4 matrix per image.
`debug_percentile` replaces every random draw by that percentile of its distribution, which is how the
parity fixtures are made deterministic.

The per-image parameters (coins, angles, factors, the composed matrices, the reflect-padding margins that follow
from them) are a few hundred numbers: they are drawn and composed on the host and uploaded once per call from
pinned memory, so that the call never waits for the GPU (the reference composes them on the device and reads the
margins back, augment.py:341-347, which drains the launch queue in front of every discriminator pass).  Only `p`
lives on the device, as the reference's buffer does; its value is read back when the buffer changes.
"""
import numpy as np
import torch

from ..torch_utils import persistence
from ..torch_utils.ops import conv2d_gradfix
from ..torch_utils.ops import grid_sample_gradfix
from ..torch_utils.ops import upfirdn2d

# Low-pass decomposition filters of the wavelets the pipeline uses (reference: training/augment.py:21-90 tabulates
# sixteen; `sym6` drives the geometric resampling, `sym2` the image-space filter bank).
WAVELETS = {
    'sym2': [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    'sym6': [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
             0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
             0.0017677118642428036, -0.007800708325034148],
}


def _mat(rows, like=None, device=None):
    """[..., r, c] matrix from a nested list whose entries are numbers or tensors of one common shape."""
    tensors = [e for row in rows for e in row if isinstance(e, torch.Tensor)]
    if not tensors:
        return torch.as_tensor(np.asarray(rows, dtype=np.float32), device=device)
    ref = tensors[0]
    flat = [e if isinstance(e, torch.Tensor) else torch.full_like(ref, float(e)) for row in rows for e in row]
    return torch.stack(flat, dim=-1).reshape(*ref.shape, len(rows), len(rows[0]))


def _shift2(tx, ty, **kw):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], **kw)


def _scale2(sx, sy, **kw):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], **kw)


def _turn2(theta, **kw):
    c, s = torch.cos(theta), torch.sin(theta)
    return _mat([[c, -s, 0], [s, c, 0], [0, 0, 1]], **kw)


def _shift3(t, **kw):
    return _mat([[1, 0, 0, t], [0, 1, 0, t], [0, 0, 1, t], [0, 0, 0, 1]], **kw)


def _scale3(s, **kw):
    return _mat([[s, 0, 0, 0], [0, s, 0, 0], [0, 0, s, 0], [0, 0, 0, 1]], **kw)


def _turn3(axis, theta):
    """Rotation by theta about the unit axis (Rodrigues), homogeneous 4 x 4."""
    x, y, z = (float(axis[i]) for i in range(3))
    s, c = torch.sin(theta), torch.cos(theta)
    k = 1 - c
    return _mat([[x * x * k + c, x * y * k - z * s, x * z * k + y * s, 0],
                 [y * x * k + z * s, y * y * k + c, y * z * k - x * s, 0],
                 [z * x * k - y * s, z * y * k + x * s, z * z * k + c, 0],
                 [0, 0, 0, 1]])


def _filter_bank():
    """Four band-pass filters from the `sym2` half-band pair: band 0 the coarsest (reference: augment.py:240-249)."""
    lo = np.asarray(WAVELETS['sym2'])
    hi = lo * ((-1) ** np.arange(lo.size))
    lo2 = np.convolve(lo, lo[::-1]) / 2
    hi2 = np.convolve(hi, hi[::-1]) / 2
    bank = np.eye(4, 1)
    for i in range(1, 4):
        spread = np.zeros((4, 2 * bank.shape[1] - 1))
        spread[:, ::2] = bank                                   # insert zeros between taps
        bank = np.stack([np.convolve(row, lo2) for row in spread])
        mid = bank.shape[1] // 2
        bank[i, mid - hi2.size // 2: mid - hi2.size // 2 + hi2.size] += hi2
    return bank


@persistence.records_init
class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125,
                 scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125,
                 brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5,
                 hue_max=1, saturation_std=1,
                 imgfilter=0, imgfilter_bands=[1, 1, 1, 1], imgfilter_std=1,
                 noise=0, cutout=0, noise_std=0.1, cutout_size=0.5):
        super().__init__()
        self.register_buffer('p', torch.ones([]))       # overall strength: multiplies every probability
        for name, value in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate,
                                aniso=aniso, xfrac=xfrac, scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std,
                                xfrac_std=xfrac_std, brightness=brightness, contrast=contrast, lumaflip=lumaflip, hue=hue,
                                saturation=saturation, brightness_std=brightness_std, contrast_std=contrast_std,
                                hue_max=hue_max, saturation_std=saturation_std, imgfilter=imgfilter,
                                imgfilter_std=imgfilter_std, noise=noise, cutout=cutout, noise_std=noise_std,
                                cutout_size=cutout_size).items():
            setattr(self, name, float(value))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(WAVELETS['sym6']))
        self.register_buffer('Hz_fbank', torch.as_tensor(_filter_bank(), dtype=torch.float32))

        self._p_seen = None                             # (stamp of the buffer, its value on the host)

    # -- random draws -----------------------------------------------------------------
    def strength(self):
        """Value of the buffer `p` on the host; read back (one synchronisation) only after the buffer was written."""
        stamp = (self.p.data_ptr(), self.p._version, self.p.device)
        if self._p_seen is None or self._p_seen[0] != stamp:
            self._p_seen = (stamp, float(self.p))
        return self._p_seen[1]

    def _gated(self, value, multiplier, identity):
        """`value` where a per-image coin with probability multiplier * p comes up, else the identity parameter."""
        coin = torch.rand([value.shape[0]] + [1] * (value.ndim - 1), device=value.device)
        return torch.where(coin < multiplier * self.strength(), value, torch.full_like(value, identity))

    @staticmethod
    def _upload(t, device):
        """Host tensor of parameters -> device, without waiting for the device."""
        if device.type == 'cpu':
            return t
        return t.contiguous().pin_memory().to(device, non_blocking=True)

    @staticmethod
    def _normal_percentile(q):
        return torch.erfinv(q * 2 - 1)      # (of a normal with standard deviation 1/sqrt(2), as the reference uses it)

    def forward(self, images, debug_percentile=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        n, channels, height, width = images.shape
        dev = images.device
        host = torch.device('cpu')                  # where the per-image parameters are drawn and composed
        q = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32, device=host)
        rand = lambda *shape: torch.rand(shape, device=host)
        randn = lambda *shape: torch.randn(shape, device=host)
        fixed = lambda like, value: torch.full_like(like, value) if not isinstance(value, torch.Tensor) else value.expand_as(like).clone()

        # ---- geometric part: G maps output pixel coordinates to input pixel coordinates -----------------------
        G = None

        def compose(m):
            nonlocal G
            G = m if G is None else G @ m

        if self.xflip > 0:
            i = self._gated(torch.floor(rand(n) * 2), self.xflip, 0)
            if q is not None:
                i = fixed(i, torch.floor(q * 2))
            compose(_scale2(1 / (1 - 2 * i), 1))
        if self.rotate90 > 0:
            i = self._gated(torch.floor(rand(n) * 4), self.rotate90, 0)
            if q is not None:
                i = fixed(i, torch.floor(q * 4))
            compose(_turn2(np.pi / 2 * i))                       # inverse of a rotation by -pi/2 i
        if self.xint > 0:
            t = self._gated((rand(n, 2) * 2 - 1) * self.xint_max, self.xint, 0)
            if q is not None:
                t = fixed(t, (q * 2 - 1) * self.xint_max)
            compose(_shift2(-torch.round(t[:, 0] * width), -torch.round(t[:, 1] * height)))
        if self.scale > 0:
            s = self._gated(torch.exp2(randn(n) * self.scale_std), self.scale, 1)
            if q is not None:
                s = fixed(s, torch.exp2(self._normal_percentile(q) * self.scale_std))
            compose(_scale2(1 / s, 1 / s))
        # a rotation before and one after the anisotropic scaling; each with probability p_rot so that
        # P(at least one) = rotate * p
        p_rot = 1 - np.sqrt(np.clip(1 - self.rotate * self.strength(), 0, 1))
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(rand(n) < p_rot, theta, torch.zeros_like(theta))
            if q is not None:
                theta = fixed(theta, (q * 2 - 1) * np.pi * self.rotate_max)
            compose(_turn2(theta))                               # inverse of a rotation by -theta
        if self.aniso > 0:
            s = self._gated(torch.exp2(randn(n) * self.aniso_std), self.aniso, 1)
            if q is not None:
                s = fixed(s, torch.exp2(self._normal_percentile(q) * self.aniso_std))
            compose(_scale2(1 / s, s))
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(rand(n) < p_rot, theta, torch.zeros_like(theta))
            if q is not None:
                theta = torch.zeros_like(theta)
            compose(_turn2(theta))
        if self.xfrac > 0:
            t = self._gated(randn(n, 2) * self.xfrac_std, self.xfrac, 0)
            if q is not None:
                t = fixed(t, self._normal_percentile(q) * self.xfrac_std)
            compose(_shift2(-t[:, 0] * width, -t[:, 1] * height))

        if G is not None:
            images = self._resample(images, G)

        # ---- colour part: C maps input colours (r, g, b, 1) to output colours ----------------------------------
        C = None

        def colour(m):
            nonlocal C
            C = m if C is None else m @ C

        luma = np.asarray([1, 1, 1, 0]) / np.sqrt(3)
        outer = torch.as_tensor(np.outer(luma, luma), dtype=torch.float32, device=host)
        eye4 = torch.eye(4, device=host)
        if self.brightness > 0:
            b = self._gated(randn(n) * self.brightness_std, self.brightness, 0)
            if q is not None:
                b = fixed(b, self._normal_percentile(q) * self.brightness_std)
            colour(_shift3(b))
        if self.contrast > 0:
            c = self._gated(torch.exp2(randn(n) * self.contrast_std), self.contrast, 1)
            if q is not None:
                c = fixed(c, torch.exp2(self._normal_percentile(q) * self.contrast_std))
            colour(_scale3(c))
        if self.lumaflip > 0:
            i = self._gated(torch.floor(rand(n, 1, 1) * 2), self.lumaflip, 0)
            if q is not None:
                i = fixed(i, torch.floor(q * 2))
            colour(eye4 - 2 * outer * i)                         # Householder reflection about the luma axis
        if self.hue > 0 and channels > 1:
            theta = self._gated((rand(n) * 2 - 1) * np.pi * self.hue_max, self.hue, 0)
            if q is not None:
                theta = fixed(theta, (q * 2 - 1) * np.pi * self.hue_max)
            colour(_turn3(luma, theta))
        if self.saturation > 0 and channels > 1:
            s = self._gated(torch.exp2(randn(n, 1, 1) * self.saturation_std), self.saturation, 1)
            if q is not None:
                s = fixed(s, torch.exp2(self._normal_percentile(q) * self.saturation_std))
            colour(outer + (eye4 - outer) * s)
        if C is not None:
            C = self._upload(C.expand(n, 4, 4) if C.ndim == 2 else C, dev)
            flat = images.reshape(n, channels, height * width)
            if channels == 3:
                flat = C[:, :3, :3] @ flat + C[:, :3, 3:]
            elif channels == 1:
                row = C[:, :3, :].mean(dim=1, keepdim=True)
                flat = flat * row[:, :, :3].sum(dim=2, keepdim=True) + row[:, :, 3:]
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')
            images = flat.reshape(n, channels, height, width)

        # ---- image-space filtering ---------------------------------------------------------------------------------
        if self.imgfilter > 0:
            bands = self.Hz_fbank.shape[0]
            assert len(self.imgfilter_bands) == bands
            expected_power = torch.as_tensor(np.array([10, 1, 1, 1]) / 13, dtype=torch.float32, device=host)   # 1/f spectrum
            gain = torch.ones([n, bands], device=host)
            for i, strength in enumerate(self.imgfilter_bands):
                t_i = self._gated(torch.exp2(randn(n) * self.imgfilter_std), self.imgfilter * strength, 1)
                if q is not None:
                    t_i = fixed(t_i, torch.exp2(self._normal_percentile(q) * self.imgfilter_std)) if strength > 0 else torch.ones_like(t_i)
                t = torch.ones([n, bands], device=host)
                t[:, i] = t_i
                t = t / (expected_power * t.square()).sum(dim=-1, keepdim=True).sqrt()      # keep the expected power
                gain = gain * t
            taps = (self._upload(gain, dev) @ self.Hz_fbank).unsqueeze(1).repeat([1, channels, 1]).reshape(n * channels, 1, -1)
            pad = self.Hz_fbank.shape[1] // 2
            images = images.reshape(1, n * channels, height, width)
            images = torch.nn.functional.pad(images, [pad, pad, pad, pad], mode='reflect')
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(2), groups=n * channels)
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(3), groups=n * channels)
            images = images.reshape(n, channels, height, width)

        # ---- corruptions -------------------------------------------------------------------------------------------
        if self.noise > 0:
            sigma = self._gated(randn(n, 1, 1, 1).abs() * self.noise_std, self.noise, 0)
            if q is not None:
                sigma = fixed(sigma, torch.erfinv(q) * self.noise_std)
            images = images + torch.randn([n, channels, height, width], device=dev) * self._upload(sigma, dev)
        if self.cutout > 0:
            size = self._gated(torch.full([n, 2, 1, 1, 1], self.cutout_size, device=host), self.cutout, 0)
            center = rand(n, 2, 1, 1, 1)
            if q is not None:
                size = torch.full_like(size, self.cutout_size)
                center = fixed(center, q)
            xs = (torch.arange(width, device=dev).reshape(1, 1, 1, -1) + 0.5) / width
            ys = (torch.arange(height, device=dev).reshape(1, 1, -1, 1) + 0.5) / height
            size, center = self._upload(size, dev), self._upload(center, dev)
            keep = ((xs - center[:, 0]).abs() >= size[:, 0] / 2) | ((ys - center[:, 1]).abs() >= size[:, 1] / 2)
            images = images * keep.to(torch.float32)
        return images

    # -- geometric execution ---------------------------------------------------------------
    def _resample(self, images, G):
        """images sampled through the inverse transform G [N, 3, 3] on the host (output pixel -> input pixel, origin
        at the image centre), band-limited by working on a 2x finer grid (reference: augment.py:338-370)."""
        n, channels, height, width = images.shape
        dev = G.device
        G = G.expand(n, 3, 3) if G.ndim == 2 else G
        taps_pad = self.Hz_geom.shape[0] // 4
        # how far the transformed image corners reach beyond the frame decides the reflect padding
        cx, cy = (width - 1) / 2, (height - 1) / 2
        corners = torch.as_tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32, device=dev)
        reach = (G @ corners.t())[:, :2, :].permute(1, 0, 2).flatten(1)             # [xy, N * 4]
        reach = torch.cat([-reach, reach]).max(dim=1).values                         # x0, y0, x1, y1
        reach = reach + torch.as_tensor([taps_pad * 2 - cx, taps_pad * 2 - cy] * 2, dtype=torch.float32, device=dev)
        reach = reach.max(torch.zeros(4, device=dev)).min(torch.as_tensor([width - 1, height - 1] * 2, dtype=torch.float32, device=dev))
        mx0, my0, mx1, my1 = (int(v) for v in reach.ceil().to(torch.int32).tolist())
        images = torch.nn.functional.pad(images, [mx0, mx1, my0, my1], mode='reflect')
        G = _shift2((mx0 - mx1) / 2, (my0 - my1) / 2, device=dev) @ G

        images = upfirdn2d.upsample2d(x=images, f=self.Hz_geom, up=2)
        G = _scale2(2, 2, device=dev) @ G @ _scale2(0.5, 0.5, device=dev)
        G = _shift2(-0.5, -0.5, device=dev) @ G @ _shift2(0.5, 0.5, device=dev)

        shape = [n, channels, (height + taps_pad * 2) * 2, (width + taps_pad * 2) * 2]
        G = _scale2(2 / images.shape[3], 2 / images.shape[2], device=dev) @ G @ _scale2(shape[3] / 2, shape[2] / 2, device=dev)
        grid = torch.nn.functional.affine_grid(theta=self._upload(G[:, :2, :], images.device), size=shape, align_corners=False)
        images = grid_sample_gradfix.grid_sample(images, grid)
        return upfirdn2d.downsample2d(x=images, f=self.Hz_geom, down=2, padding=-taps_pad * 2, flip_filter=True)
