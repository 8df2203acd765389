"""The snapshot reader (`ga_gan_amd.legacy`) on records written here in the reference's layout: a pickled dict whose
networks are `torch_utils.persistence._reconstruct_persistent_obj(meta)` records carrying module source text and the
module `__dict__` (reference torch_utils/persistence.py:117-125, training/training_loop.py:586-601); on a snapshot the
reference itself pickled (`tests/golden/snapshot_ref.pkl`, embedded source blanked before writing) with the state it
held; on a synthetic TensorFlow-era pickle with what the reference's converter made of it
(`tests/golden/snapshot_tf.*`); and the writer, read back here and - where /root/reference exists - by the reference's
own loader.  `tools/gen_golden_snapshots.py` writes the fixtures."""
import collections
import io
import json
import os
import pickle
import subprocess
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


# ----------------------------------------------------------------------------
# Fixtures written by the reference.

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
REFERENCE = '/root/reference/DissimilarDomains'


def _fixture(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        arrays = {k: z[k] for k in z.files}
    return json.loads(bytes(arrays.pop('meta')).decode()), arrays


def _same_state(net, arrays, prefix, names):
    have = net.state_dict()
    assert list(have) == names
    for key in names:
        want = torch.from_numpy(arrays[prefix + key])
        assert have[key].dtype == want.dtype and torch.equal(have[key], want), prefix + key


def test_reads_a_snapshot_pickled_by_the_reference():
    meta, arrays = _fixture('snapshot_ref')
    with open(os.path.join(GOLDEN, 'snapshot_ref.pkl'), 'rb') as f:
        out = legacy.load_network_pkl(f)
    assert out['training_set_kwargs'] == meta['training_set_kwargs']
    for name, cls in (('G', networks.Generator), ('D', networks.Discriminator), ('G_ema', networks.Generator)):
        assert type(out[name]) is cls and not out[name].training
        assert json.loads(json.dumps(out[name].init_kwargs)) == meta[name + '_init_kwargs']
        _same_state(out[name], arrays, name + '/', meta[name + '_state_names'])
    pipe = out['augment_pipe']
    assert type(pipe).__name__ == 'AugmentPipe' and pipe.init_kwargs == meta['augment_pipe_init_kwargs']
    _same_state(pipe, arrays, 'augment_pipe/', meta['augment_pipe_state_names'])
    assert float(pipe.p) == pytest.approx(0.3) and pipe.xflip == 1 and pipe.brightness == 1 and pipe.contrast == 0


def test_converts_a_tensorflow_pickle_as_the_reference_does():
    meta, arrays = _fixture('snapshot_tf')
    with open(os.path.join(GOLDEN, 'snapshot_tf.pkl'), 'rb') as f:
        blob = f.read()
    assert b'TensorFlow graph source was executed' in blob       # the graph source is in the file and is not run
    out = legacy.load_network_pkl(io.BytesIO(blob))
    assert out['augment_pipe'] is None and out['training_set_kwargs'] is None
    for name, cls in (('G', networks.Generator), ('D', networks.Discriminator), ('G_ema', networks.Generator)):
        assert type(out[name]) is cls and not out[name].training and not any(p.requires_grad for p in out[name].parameters())
        assert json.loads(json.dumps(out[name].init_kwargs)) == meta[name + '_init_kwargs']
        _same_state(out[name], arrays, name + '/', meta[name + '_state_names'])


def _tf_records():
    with open(os.path.join(GOLDEN, 'snapshot_tf.pkl'), 'rb') as f:
        return legacy.load_records(f)


def test_tensorflow_records_are_plain_data():
    tf_G, tf_D, tf_Gs = _tf_records()
    assert all(isinstance(net, legacy.TFNetwork) for net in (tf_G, tf_D, tf_Gs))
    assert not hasattr(tf_G, 'build_module_src')
    params = tf_G.parameters()
    assert params['synthesis/4x4/Const/const'].shape == (1, 16, 4, 4) and params['mapping/Dense0/weight'].shape == (20, 24)
    assert 'dlatent_avg' in params and set(tf_G.components) == {'mapping', 'synthesis'}


def test_tensorflow_conversion_refuses_what_it_does_not_understand():
    tf_G, tf_D, _ = _tf_records()
    tf_G.static_kwargs['fused_modconv_plus'] = True
    with pytest.raises(ValueError, match='Unknown TensorFlow kwarg'):
        legacy.convert_tf_generator(tf_G)
    tf_D.version = 3
    with pytest.raises(ValueError, match='version too low'):
        legacy.convert_tf_discriminator(tf_D)
    tf_G, tf_D, _ = _tf_records()
    tf_D.variables = [(n, v) for n, v in tf_D.variables if n != 'Output/bias']
    with pytest.raises(KeyError, match='Output/bias'):
        legacy.convert_tf_discriminator(tf_D)
    tf_G.components['synthesis'].variables = [(n, v[..., :-1] if n == '8x8/Conv1/weight' else v) for n, v in tf_G.components['synthesis'].variables]
    with pytest.raises(ValueError, match='8x8/Conv1/weight'):
        legacy.convert_tf_generator(tf_G)


def test_tensorflow_progressive_image_layers_select_the_orig_architecture():
    """`ToRGB_lod<n>` / `FromRGB_lod<n>` scopes (one image layer per level of detail, n = 0 at full resolution) stand
    for the 'orig' architecture (legacy.py:168-173, 268-273).  The reference's generator branch cannot run (it assigns
    to `kwargs.synthesis.kwargs`); here both convert."""
    tf_G, tf_D, _ = _tf_records()
    syn = tf_G.components['synthesis']
    keep = [(n, v) for n, v in syn.variables if '/ToRGB/' not in n]
    top = [(n.replace('16x16/ToRGB/', 'ToRGB_lod0/'), v) for n, v in syn.variables if n.startswith('16x16/ToRGB/')]
    syn.variables = keep + top
    G = legacy.convert_tf_generator(tf_G)
    assert G.init_kwargs['synthesis_kwargs']['architecture'] == 'orig'
    assert [n for n, _ in G.named_modules() if n.endswith('torgb')] == ['synthesis.b16.torgb']
    want = dict(top)['ToRGB_lod0/weight'].transpose(3, 2, 0, 1)
    assert np.array_equal(G.synthesis.b16.torgb.weight.numpy(), want)
    tf_D.variables = [(n.replace('16x16/FromRGB/', 'FromRGB_lod0/'), v) for n, v in tf_D.variables if '/Skip/' not in n]
    D = legacy.convert_tf_discriminator(tf_D)
    assert D.init_kwargs['architecture'] == 'orig' and not hasattr(D.b16, 'skip')


def test_force_fp16_rebuilds_with_sixteen_bit_blocks():
    with open(os.path.join(GOLDEN, 'snapshot_ref.pkl'), 'rb') as f:
        plain = legacy.load_network_pkl(f)
    with open(os.path.join(GOLDEN, 'snapshot_ref.pkl'), 'rb') as f:
        low = legacy.load_network_pkl(f, force_fp16=True)
    assert low['G'].init_kwargs['synthesis_kwargs']['num_fp16_res'] == 4 and low['G'].init_kwargs['synthesis_kwargs']['conv_clamp'] == 256
    assert low['D'].init_kwargs['num_fp16_res'] == 4 and low['D'].init_kwargs['conv_clamp'] == 256
    assert low['G'].synthesis.b32.use_fp16 and not plain['G'].synthesis.b32.use_fp16
    for name in ('G', 'D', 'G_ema'):
        for (ka, a), (kb, b) in zip(plain[name].state_dict().items(), low[name].state_dict().items()):
            assert ka == kb and torch.equal(a, b)


# ----------------------------------------------------------------------------
# Writer.

def _written(training=False):
    G, D, G_ema = _nets()
    from ga_gan_amd.training import augment
    pipe = augment.AugmentPipe(xflip=1, scale=1)
    pipe.p.copy_(torch.as_tensor(0.125))
    G.synthesis.b8.conv0.weight._sg_memo = {'cached': lambda: None}         # caches of this package stay out of the file
    G.synthesis.b8.conv0.__dict__['_sg_derived'] = {'k': object()}
    data = dict(G=G, D=D, G_ema=G_ema, augment_pipe=pipe, training_set_kwargs=dict(path='x.zip', resolution=32, use_labels=False))
    for m in (G, D, G_ema, pipe):
        m.train(training).requires_grad_(training)
    f = io.BytesIO()
    legacy.save_network_pkl(f, data)
    return data, f.getvalue()


def test_written_snapshot_reads_back():
    data, blob = _written()
    assert b'_sg_' not in blob and b'ga_gan_amd' not in blob and b'ga-gan_amd' not in blob
    assert b'torch_utils.persistence' in blob and b'from training.networks import Generator' in blob
    assert 'torch_utils.persistence' not in sys.modules or hasattr(sys.modules['torch_utils.persistence'], 'persistent_class')
    out = legacy.load_network_pkl(io.BytesIO(blob))
    assert out['training_set_kwargs'] == data['training_set_kwargs']
    for name in ('G', 'D', 'G_ema', 'augment_pipe'):
        assert type(out[name]) is type(data[name]) and out[name].init_kwargs == data[name].init_kwargs
        want, have = data[name].state_dict(), out[name].state_dict()
        assert list(want) == list(have) and all(torch.equal(want[k], have[k]) for k in want)
    rec = legacy.load_records(io.BytesIO(blob))['G']
    layer = rec.state['_modules']['synthesis'].state['_modules']['b8'].state['_modules']['conv0']
    assert layer.class_name == 'SynthesisLayer' and layer.init_args == (32, 32) and layer.state['up'] == 2
    assert rec.state['training'] is False and rec.state['z_dim'] == 32


def test_writer_refuses_code():
    G, D, G_ema = _nets()
    G.synthesis.callback = print
    with pytest.raises(TypeError, match='G.synthesis.callback'):
        legacy.save_network_pkl(io.BytesIO(), dict(G=G, D=D, G_ema=G_ema))
    with pytest.raises(TypeError, match='constructor arguments'):
        legacy.save_network_pkl(io.BytesIO(), dict(G=torch.nn.Linear(2, 2)))


_READ_WITH_THE_REFERENCE = r"""
import io, json, sys
import numpy as np, torch
sys.dont_write_bytecode = True
sys.path.insert(0, sys.argv[1])
import legacy
from torch_utils import persistence
from training import networks, augment
with open(sys.argv[2], 'rb') as f:
    data = legacy.load_network_pkl(f)
out, meta = {}, {}
for name, cls in (('G', networks.Generator), ('D', networks.Discriminator), ('G_ema', networks.Generator), ('augment_pipe', augment.AugmentPipe)):
    net = data[name]
    assert isinstance(net, cls) and persistence.is_persistent(net), name
    meta[name] = dict(init_kwargs=json.loads(json.dumps(net.init_kwargs)), training=net.training, names=list(net.state_dict()),
                      grads=sorted({p.requires_grad for p in net.parameters()}))
    out.update({name + '/' + k: v.numpy() for k, v in net.state_dict().items()})
meta['training_set_kwargs'] = data['training_set_kwargs']
torch.manual_seed(5)
img = torch.randn(4, 3, 32, 32)
z = torch.randn(2, 32)
with torch.no_grad():
    out['D.logits'] = data['D'](img, torch.zeros(4, 0)).numpy()
    out['G.ws'] = data['G_ema'].mapping(z, torch.zeros(2, 0), truncation_psi=0.7).numpy()
    rebuilt = networks.Generator(*data['G'].init_args, **data['G'].init_kwargs)        # persistence.py:86-89
    from torch_utils import misc
    misc.copy_params_and_buffers(data['G'], rebuilt, require_all=True)
np.savez(sys.argv[3], meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), img=img.numpy(), z=z.numpy(), **out)
"""


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='the reference is only present in the build container')
def test_the_reference_reads_a_snapshot_written_here(tmp_path):
    data, blob = _written()
    (tmp_path / 'snap.pkl').write_bytes(blob)
    (tmp_path / 'read.py').write_text(_READ_WITH_THE_REFERENCE)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH='')
    run = subprocess.run([sys.executable, str(tmp_path / 'read.py'), REFERENCE, str(tmp_path / 'snap.pkl'), str(tmp_path / 'out.npz')],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    with np.load(tmp_path / 'out.npz') as z:
        got = {k: z[k] for k in z.files}
    meta = json.loads(bytes(got.pop('meta')).decode())
    assert meta['training_set_kwargs'] == data['training_set_kwargs']
    for name in ('G', 'D', 'G_ema', 'augment_pipe'):
        assert meta[name]['init_kwargs'] == json.loads(json.dumps(data[name].init_kwargs))
        assert meta[name]['training'] is False and meta[name]['grads'] in ([False], [])
        want = data[name].state_dict()
        assert meta[name]['names'] == list(want)
        for key, value in want.items():
            assert np.array_equal(got[name + '/' + key], value.numpy()), name + '/' + key
    # what the reference computes with the networks it rebuilt from the file, against the CPU oracle on the same tensors
    from oracle import stylegan2 as oracle
    cfg = oracle.DiscriminatorConfig(img_resolution=32, channel_base=512, channel_max=32, mbstd_group_size=2, mapping_layers=0)
    D_state = {k: v.double() for k, v in data['D'].state_dict().items()}
    logits = oracle.discriminator(D_state, torch.from_numpy(got['img']).double(), torch.zeros(4, 0).double(), cfg)
    assert np.allclose(got['D.logits'], logits.numpy(), rtol=1e-4, atol=1e-5)
