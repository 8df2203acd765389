"""Binding of `sg_images_to_float` (include/sg_ops.h): a float batch in [-1, 1] out of a uint8 image set that
resides in device memory, gathered by index and mirrored per sample in the same pass.

It stands in for the loader + conversion in front of the reference's training step (`Dataset.__getitem__`,
training/dataset.py:85-93, and `phase_real_img.to(device).to(torch.float32) / 127.5 - 1`,
training/training_loop.py:441).
"""
import ctypes

import torch

from . import _lib


class ImagesParams(ctypes.Structure):
    _fields_ = [('images', ctypes.c_void_p), ('index', ctypes.c_void_p), ('flip', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('dtype', ctypes.c_int32), ('n', ctypes.c_int32), ('c', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('count', ctypes.c_int64)]


def images_to_float(images, index=None, flip=None, dtype=torch.float32):
    """images: uint8 [count, C, H, W] on the GPU; index: int64 [n] on the GPU (None: all `count`
    images in order); flip: uint8 / bool [n] on the GPU (None: no mirror).  Returns [n, C, H, W] of `dtype`."""
    if not (isinstance(images, torch.Tensor) and images.is_cuda):
        raise RuntimeError('images must reside on CUDA device')
    if images.dtype != torch.uint8 or images.ndim != 4:
        raise RuntimeError(f'images must be a uint8 [count, C, H, W] tensor, got {images.dtype} {tuple(images.shape)}')
    if dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError(f'unsupported output dtype {dtype}')
    images = images.contiguous()
    count, c, h, w = images.shape
    n = count if index is None else int(index.shape[0])
    if index is not None:
        if index.dtype != torch.int64 or index.ndim != 1 or index.device != images.device:
            raise RuntimeError('index must be an int64 vector on the device of the images')
        index = index.contiguous()
    if flip is not None:
        if flip.dtype not in (torch.uint8, torch.bool) or flip.ndim != 1 or flip.device != images.device or flip.shape[0] != n:
            raise RuntimeError(f'flip must be a uint8 or bool vector of {n} entries on the device of the images')
        flip = flip.contiguous().view(torch.uint8)
    out = torch.empty([n, c, h, w], dtype=dtype, device=images.device)
    if out.numel() >= 2 ** 31 * 16:
        raise RuntimeError('batch is too large')
    args = ImagesParams(_lib.ptr(images), _lib.ptr(index), _lib.ptr(flip), _lib.ptr(out), _lib.DTYPES[dtype], n, c, h, w, count)
    _lib.launch('sg_images_to_float', args, images, 'images_kernel', out.numel() * (1 + out.element_size()))
    return out
