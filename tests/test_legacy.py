"""The snapshot reader (`ga_gan_amd.legacy`) on records written here in the reference's layout: a pickled dict whose
networks are `torch_utils.persistence._reconstruct_persistent_obj(meta)` records carrying module source text and the
module `__dict__` (reference torch_utils/persistence.py:117-125, training/training_loop.py:586-601).  Parity unpinned:
the reference ships no snapshot file to read."""
import collections
import io
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from ga_gan_amd import legacy
from ga_gan_amd.training import networks

BOOBY_TRAP = 'raise SystemExit("the snapshot\'s embedded source was executed")\n'


class _Persistent:
    """Stands in for a `persistent_class` instance while the test pickles it."""

    def __init__(self, class_name, state, reconstruct):
        self.meta = dict(type='class', version=6, module_src=BOOBY_TRAP, class_name=class_name, state=state)
        self.reconstruct = reconstruct

    def __reduce__(self):
        return self.reconstruct, (self.meta,), None


def _write_snapshot(modules, extra=None):
    """Bytes of `pickle.dump(dict(name=module, ...))` as the reference would write it for these torch modules."""
    persistence = types.ModuleType('torch_utils.persistence')
    util = types.ModuleType('dnnlib.util')

    def _reconstruct_persistent_obj(meta):      # only its global name goes into the file
        raise AssertionError
    _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
    _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
    persistence._reconstruct_persistent_obj = _reconstruct_persistent_obj
    easy = type('EasyDict', (dict,), {'__module__': 'dnnlib.util', '__qualname__': 'EasyDict'})
    util.EasyDict = easy

    def wrap_kwargs(value):
        if isinstance(value, dict):
            return easy({k: wrap_kwargs(v) for k, v in value.items()})
        return value

    def convert(module, init_kwargs):
        state = collections.OrderedDict()
        for key, value in module.__dict__.items():
            if key == '_modules':
                state[key] = collections.OrderedDict((n, convert(m, {})) for n, m in value.items())
            elif key in ('_parameters', '_buffers', '_non_persistent_buffers_set', 'training') or isinstance(value, (int, float, str, bool, list, type(None))):
                state[key] = value
        state['_init_args'] = ()
        state['_init_kwargs'] = wrap_kwargs(init_kwargs)
        state['some_numpy_scalar'] = np.float32(1.5)
        state['some_numpy_array'] = np.arange(4)
        return _Persistent(type(module).__name__, state, _reconstruct_persistent_obj)

    saved = {k: sys.modules.get(k) for k in ('torch_utils', 'torch_utils.persistence', 'dnnlib', 'dnnlib.util')}
    sys.modules.update({'torch_utils': types.ModuleType('torch_utils'), 'torch_utils.persistence': persistence,
                        'dnnlib': types.ModuleType('dnnlib'), 'dnnlib.util': util})
    try:
        data = {name: convert(module, kwargs) for name, (module, kwargs) in modules.items()}
        data.update(extra or {})
        data['training_set_kwargs'] = easy(class_name='training.dataset.ImageFolderDataset', path='x.zip', resolution=32)
        return pickle.dumps(data)
    finally:
        for key, value in saved.items():
            if value is None:
                sys.modules.pop(key, None)
            else:
                sys.modules[key] = value


G_KWARGS = dict(z_dim=32, c_dim=0, w_dim=32, img_resolution=32, img_channels=3, mapping_kwargs=dict(num_layers=2),
                synthesis_kwargs=dict(channel_base=512, channel_max=32, num_fp16_res=1, conv_clamp=256))
D_KWARGS = dict(c_dim=0, img_resolution=32, img_channels=3, channel_base=512, channel_max=32, epilogue_kwargs=dict(mbstd_group_size=2))


def _nets():
    torch.manual_seed(0)
    G = networks.Generator(**G_KWARGS)
    D = networks.Discriminator(**D_KWARGS)
    G_ema = networks.Generator(**G_KWARGS)
    with torch.no_grad():
        for p in G_ema.parameters():
            p.add_(0.25)
    return G, D, G_ema


def test_snapshot_round_trip_without_running_its_source():
    G, D, G_ema = _nets()
    blob = _write_snapshot({'G': (G, G_KWARGS), 'D': (D, D_KWARGS), 'G_ema': (G_ema, G_KWARGS)})
    assert BOOBY_TRAP.encode() in blob                       # the source text is in the file ...
    out = legacy.load_network_pkl(io.BytesIO(blob))          # ... and reading the file does not run it
    assert out['augment_pipe'] is None
    assert out['training_set_kwargs'] == dict(class_name='training.dataset.ImageFolderDataset', path='x.zip', resolution=32)
    for name, ref in (('G', G), ('D', D), ('G_ema', G_ema)):
        got = out[name]
        assert type(got) is type(ref) and not got.training
        want = ref.state_dict()
        have = got.state_dict()
        assert list(have) == list(want)
        for key in want:
            assert torch.equal(have[key], want[key]), key
        assert not any(p.requires_grad for p in got.parameters())


def test_records_expose_constructor_arguments_and_plain_data():
    G, D, G_ema = _nets()
    data = legacy.load_records(io.BytesIO(_write_snapshot({'G': (G, G_KWARGS), 'D': (D, D_KWARGS), 'G_ema': (G_ema, G_KWARGS)})))
    rec = data['G']
    assert isinstance(rec, legacy.Record) and rec.class_name == 'Generator'
    assert type(rec.init_kwargs) is dict and rec.init_kwargs['synthesis_kwargs']['channel_max'] == 32
    assert rec.state['some_numpy_scalar'] == np.float32(1.5) and rec.state['some_numpy_array'].tolist() == [0, 1, 2, 3]
    names = list(legacy.state_dict_of(rec))
    assert names == list(G.state_dict())


class _Evil:
    def __reduce__(self):
        import os
        return os.system, ('echo pwned > /tmp/ga_gan_amd_pwned',)


def test_code_in_a_snapshot_is_refused():
    G, D, G_ema = _nets()
    blob = _write_snapshot({'G': (G, G_KWARGS), 'D': (D, D_KWARGS), 'G_ema': (G_ema, G_KWARGS)}, extra={'hook': _Evil()})
    with pytest.raises(pickle.UnpicklingError, match='code is not run'):
        legacy.load_network_pkl(io.BytesIO(blob))
    with pytest.raises(pickle.UnpicklingError):
        legacy.load_records(io.BytesIO(pickle.dumps({'f': torch.nn.Linear})))


def test_not_a_snapshot():
    with pytest.raises(pickle.UnpicklingError, match='not a network snapshot'):
        legacy.load_network_pkl(io.BytesIO(pickle.dumps({'G': 1})))
