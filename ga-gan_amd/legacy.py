"""Reader for the reference's network snapshots (`network-snapshot-*.pkl`) that executes nothing from the file.

The reference writes `pickle.dump(dict(G=..., D=..., G_ema=..., augment_pipe=..., training_set_kwargs=...))`
(training/training_loop.py:586-601) where every network is a `persistence.persistent_class` object: its pickle record is
`_reconstruct_persistent_obj(dict(type='class', version, module_src, class_name, state))`
(torch_utils/persistence.py:117-125) - the SOURCE TEXT of the defining module plus the module's `__dict__`.  The
reference's loader (legacy.py:21-61, persistence.py:179-227) `exec`s that source.  This one never does: a restricted
unpickler admits only data constructors (containers, numpy arrays and scalars, tensors) and turns every persistent
record into a plain `Record(class_name, state)`; the tensors of a record tree are then collected under their
`state_dict()` names and loaded into THIS package's networks, which carry the reference's class names, constructor
arguments and parameter names.  The embedded source is dropped unread.

Parity unpinned: the reference ships no snapshot and none can be fetched here; `tests/test_legacy.py` writes records in
the layout above and reads them back.  TensorFlow-era pickles (legacy.py:117-299) are not converted.
"""
import collections
import io
import pickle

import numpy as np
import torch

_TORCH_DATA = {'_rebuild_tensor_v2', '_rebuild_parameter', '_rebuild_tensor', '_rebuild_parameter_with_state'}
_NUMPY_DATA = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
               ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
               ('numpy', 'ndarray'), ('numpy', 'dtype')}
_BUILTINS = {'set': set, 'frozenset': frozenset, 'slice': slice, 'range': range, 'complex': complex, 'bytearray': bytearray}


class Record:
    """A persistent object of the reference, unconstructed: the class name it was written with and its `__dict__`."""

    def __init__(self, class_name, state):
        self.class_name = class_name
        self.state = state

    @property
    def init_kwargs(self):
        return dict(self.state.get('_init_kwargs', {}))

    @property
    def init_args(self):
        return tuple(self.state.get('_init_args', ()))

    def __repr__(self):
        return f'Record({self.class_name!r}, {len(self.state)} fields)'


def _record(meta):
    if not isinstance(meta, dict) or meta.get('type') != 'class' or not isinstance(meta.get('class_name'), str):
        raise pickle.UnpicklingError('malformed persistent-object record')
    state = meta.get('state')
    return Record(meta['class_name'], dict(state) if isinstance(state, dict) else {})      # module_src: dropped unread


def _storage_from_bytes(data):
    """What `torch.storage._load_from_bytes` does for a tensor pickled outside `torch.save`, through the loader that
    builds tensors and nothing else."""
    return torch.load(io.BytesIO(data), map_location='cpu', weights_only=True)


class _DataOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in (('torch_utils.persistence', '_reconstruct_persistent_obj'), ('persistence', '_reconstruct_persistent_obj')):
            return _record
        if (module, name) in (('dnnlib.util', 'EasyDict'), ('dnnlib', 'EasyDict')):
            return dict
        if module == 'collections' and name == 'OrderedDict':
            return collections.OrderedDict
        if module == 'torch._utils' and name in _TORCH_DATA:
            return getattr(torch._utils, name)
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            return _storage_from_bytes
        if module == 'torch' and isinstance(getattr(torch, name, None), torch.dtype):
            return getattr(torch, name)
        if (module, name) in (('torch', 'Size'), ('torch', 'device')):
            return getattr(torch, name)
        if (module, name) in _NUMPY_DATA:
            if name in ('_reconstruct', 'scalar'):
                multiarray = np._core.multiarray if hasattr(np, '_core') else np.core.multiarray
                return getattr(multiarray, name)
            return getattr(np, name)
        if module == 'builtins' and name in _BUILTINS:
            return _BUILTINS[name]
        raise pickle.UnpicklingError(f'snapshot refers to {module}.{name}: only data is read from snapshots, code is not run')


def load_records(f):
    """The snapshot's top-level object with every persistent object as a `Record`.  `f`: a binary file object."""
    return _DataOnlyUnpickler(f).load()


def state_dict_of(record, prefix=''):
    """Parameters and persistent buffers of a record tree under the names `torch.nn.Module.state_dict()` gives them."""
    out = collections.OrderedDict()
    state = record.state
    for name, value in (state.get('_parameters') or {}).items():
        if value is not None:
            out[prefix + name] = value.detach()
    skip = set(state.get('_non_persistent_buffers_set') or ())
    for name, value in (state.get('_buffers') or {}).items():
        if value is not None and name not in skip:
            out[prefix + name] = value
    for name, child in (state.get('_modules') or {}).items():
        if isinstance(child, Record):
            out.update(state_dict_of(child, prefix + name + '.'))
        elif child is not None:
            raise pickle.UnpicklingError(f'sub-module {prefix + name} is not a persistent object')
    return out


def _build(record, classes):
    if record.class_name not in classes:
        raise pickle.UnpicklingError(f'snapshot holds a {record.class_name}, which this package does not define')
    net = classes[record.class_name](*record.init_args, **_plain(record.init_kwargs))
    net.load_state_dict(state_dict_of(record), strict=True)
    return net.eval().requires_grad_(False)


def _plain(value):
    """Constructor arguments as plain Python containers (the reference keeps them in attribute dictionaries)."""
    if isinstance(value, Record):
        raise pickle.UnpicklingError('a constructor argument is a persistent object')
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return type(value)(_plain(v) for v in value)
    if isinstance(value, np.generic):
        return value.item()
    return value


def load_network_pkl(f):
    """dict(G, D, G_ema, training_set_kwargs, augment_pipe) as the reference's `legacy.load_network_pkl` returns it
    (legacy.py:21-61), with the networks rebuilt from this package's classes (`training.networks.Generator` /
    `Discriminator`, `training.augment.AugmentPipe`) on the CPU, in eval mode, without gradients."""
    from .training import augment, networks
    data = load_records(f)
    if not isinstance(data, dict) or not all(isinstance(data.get(k), Record) for k in ('G', 'D', 'G_ema')):
        raise pickle.UnpicklingError('not a network snapshot: expected a dict with persistent G, D and G_ema')
    classes = {'Generator': networks.Generator, 'Discriminator': networks.Discriminator, 'AugmentPipe': augment.AugmentPipe}
    out = {k: _build(data[k], classes) for k in ('G', 'D', 'G_ema')}
    pipe = data.get('augment_pipe')
    out['augment_pipe'] = _build(pipe, classes) if isinstance(pipe, Record) else None
    kwargs = data.get('training_set_kwargs')
    out['training_set_kwargs'] = _plain(kwargs) if isinstance(kwargs, dict) else None
    return out
