"""Small helpers shared by the op API and the networks (shape assertions, profiler
scopes, DDP-free gradient synchronisation context)."""
import contextlib

import torch


def assert_shape(tensor, ref_shape):
    """None in ref_shape leaves that dimension free (torch_utils/misc.py:84-98 of the reference)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref_size) in enumerate(zip(tensor.shape, ref_shape)):
        if ref_size is not None and size != ref_size:
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref_size}')


def profiled_function(fn):
    """Wrap fn in a torch profiler scope named after it."""
    def wrapper(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


def require_gpu(x, what):
    """The product path has no CPU fallback: a CPU tensor is a usage error."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError(f'{what}: this build runs on the GPU only (got a {getattr(x, "device", type(x))} input); '
                           'the CPU restatement lives in the test oracle, not in the product path')


def named_params_and_buffers(module):
    return list(module.named_parameters()) + list(module.named_buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Copy tensors by name (torch_utils/misc.py:164-171 of the reference)."""
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert name in src or not require_all, name
        if name in src:
            tensor.copy_(src[name].detach()).requires_grad_(tensor.requires_grad)


def nan_to_num_(t, nan=0.0, posinf=1e5, neginf=-1e5):
    return torch.nan_to_num(t, nan=nan, posinf=posinf, neginf=neginf, out=t)
