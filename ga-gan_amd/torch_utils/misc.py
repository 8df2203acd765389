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


class InfiniteSampler:
    """Endless stream of dataset indices for one rank (same arguments and the same sequence as the reference's
    sampler, torch_utils/misc.py:117-148): a seeded permutation is walked cyclically; after position i has been
    visited, its entry trades places with one at most `window_size * len(dataset)` positions earlier, so the
    order keeps drifting; rank r of num_replicas takes the visits whose running number is r modulo num_replicas.
    Usable as the `sampler` of a torch DataLoader and as a plain iterator (`take(n)`)."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        if len(dataset) <= 0:
            raise ValueError('the dataset is empty')
        if not (num_replicas > 0 and 0 <= rank < num_replicas):
            raise ValueError(f'rank {rank} is not one of {num_replicas} replicas')
        if not 0 <= window_size <= 1:
            raise ValueError('window_size must lie in [0, 1]')
        self.dataset = dataset
        self.rank = rank
        self.num_replicas = num_replicas
        self.shuffle = shuffle
        self.seed = seed
        self.window_size = window_size
        self._running = None

    def __iter__(self):
        import numpy as np
        size = len(self.dataset)
        order = np.arange(size)
        draw, window = None, 0
        if self.shuffle:
            draw = np.random.RandomState(self.seed)
            draw.shuffle(order)
            window = int(np.rint(size * self.window_size))
        visit = 0
        while True:
            here = visit % size
            if visit % self.num_replicas == self.rank:
                yield order[here]
            if window >= 2:
                there = (here - draw.randint(window)) % size
                order[here], order[there] = order[there], order[here]
            visit += 1

    def take(self, n):
        """The next n indices of one continuing pass (host int64 array)."""
        import numpy as np
        if self._running is None:
            self._running = iter(self)
        return np.fromiter((next(self._running) for _ in range(n)), dtype=np.int64, count=n)
