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

`save_network_pkl` writes the same layout from this package's networks, for the reference to read: every module
becomes a `_reconstruct_persistent_obj(meta)` record whose `state` is the module's `__dict__` (the attribute names of
this package's classes are the reference's) with clean CPU tensors, and whose `module_src` is one import line naming
the reference module that defines the class - run by the reference's loader, it resolves the class to the reader's
own definition.

TensorFlow-era pickles - a tuple `(G, D, Gs)` of `dnnlib.tflib.network.Network` objects (legacy.py:25-30, 66-74) -
are read as data too (`TFNetwork`: `version`, `static_kwargs`, `variables`, `components`) and converted by
`convert_tf_generator` / `convert_tf_discriminator`: constructor arguments from `static_kwargs`, tensors through a
table of (parameter name -> TensorFlow variable name, layout change) (legacy.py:117-299).

What pins this file: `tests/golden/snapshot_ref.pkl` is a snapshot the reference itself pickled (its classes, its
`__reduce__`; the embedded source blanked before writing so that no source travels) with the state and outputs it
had next to it, and `tests/golden/snapshot_tf.npz` holds what the reference's converter made of a synthetic
TensorFlow-era pickle (`tools/gen_golden_snapshots.py` writes both); `tests/test_legacy.py` also hands a snapshot
written here to the reference's own loader where the reference is present.  No snapshot trained by the reference ships
with it, and no real TensorFlow pickle can be fetched: records of real training runs remain unread.
"""
import collections
import contextlib
import io
import pickle
import re
import sys
import types

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


class TFNetwork:
    """A `dnnlib.tflib.network.Network` of a TensorFlow-era pickle, as data: `version`, `static_kwargs` (constructor
    arguments of the TensorFlow graph function), `variables` ([(name, array)]) and `components` ({name: TFNetwork}).
    Its `build_module_src` - source text again - is dropped unread."""
    _FIELDS = ('version', 'name', 'static_kwargs', 'variables', 'components', 'build_func_name')

    def __setstate__(self, state):
        if not isinstance(state, dict):
            raise pickle.UnpicklingError('malformed TensorFlow network record')
        for key in self._FIELDS:
            if key in state:
                setattr(self, key, state[key])

    def parameters(self, prefix=''):
        """{'component/.../variable': array} over this network and its components (legacy.py:79-90)."""
        out = {prefix + name: np.asarray(value) for name, value in getattr(self, 'variables', [])}
        for name, part in dict(getattr(self, 'components', {})).items():
            if not isinstance(part, TFNetwork):
                raise pickle.UnpicklingError(f'component {prefix + name} is not a network record')
            out.update(part.parameters(prefix + name + '/'))
        return out


def _new_plain_object(cls, base=object, state=None):
    """`copyreg._reconstructor` (pickle protocols 0 and 1) for the one class such files may instantiate."""
    if cls is not TFNetwork or base is not object:
        raise pickle.UnpicklingError('only network records are built from protocol-0/1 object entries')
    return TFNetwork()


def _bytes_from_text(text, encoding='latin1'):
    """`_codecs.encode(text, 'latin1')`: how pickle protocols below 3 carry an array's bytes."""
    if not isinstance(text, str) or encoding not in ('latin1', 'latin-1'):
        raise pickle.UnpicklingError('only latin-1 byte strings are decoded')
    return text.encode('latin1')


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
        if (module, name) == ('dnnlib.tflib.network', 'Network'):
            return TFNetwork
        if (module, name) in (('copyreg', '_reconstructor'), ('copy_reg', '_reconstructor')):
            return _new_plain_object
        if (module, name) in (('builtins', 'object'), ('__builtin__', 'object')):
            return object
        if (module, name) == ('_codecs', 'encode'):
            return _bytes_from_text
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


def load_network_pkl(f, force_fp16=False):
    """dict(G, D, G_ema, training_set_kwargs, augment_pipe) as the reference's `legacy.load_network_pkl` returns it
    (legacy.py:21-61), with the networks rebuilt from this package's classes (`training.networks.Generator` /
    `Discriminator`, `training.augment.AugmentPipe`) on the CPU, in eval mode, without gradients.  A TensorFlow-era
    file - a tuple of three network records - is converted.  `force_fp16` rebuilds the networks with 16-bit arithmetic
    in their four highest resolutions and a clamp of 256, as the reference's flag does (legacy.py:46-60)."""
    from .training import augment, networks
    data = load_records(f)
    if isinstance(data, tuple) and len(data) == 3 and all(isinstance(net, TFNetwork) for net in data):
        out = dict(G=convert_tf_generator(data[0]), D=convert_tf_discriminator(data[1]), G_ema=convert_tf_generator(data[2]),
                   augment_pipe=None, training_set_kwargs=None)
    else:
        if not isinstance(data, dict) or not all(isinstance(data.get(k), Record) for k in ('G', 'D', 'G_ema')):
            raise pickle.UnpicklingError('not a network snapshot: expected a dict with persistent G, D and G_ema')
        classes = {'Generator': networks.Generator, 'Discriminator': networks.Discriminator, 'AugmentPipe': augment.AugmentPipe}
        out = {k: _build(data[k], classes) for k in ('G', 'D', 'G_ema')}
        pipe = data.get('augment_pipe')
        out['augment_pipe'] = _build(pipe, classes) if isinstance(pipe, Record) else None
        kwargs = data.get('training_set_kwargs')
        out['training_set_kwargs'] = _plain(kwargs) if isinstance(kwargs, dict) else None
    if force_fp16:
        for key in ('G', 'D', 'G_ema'):
            out[key] = _with_low_precision(out[key], generator=key.startswith('G'))
    return out


def _with_low_precision(net, generator):
    kwargs = net.init_kwargs
    target = kwargs
    if generator:
        target = kwargs['synthesis_kwargs'] = dict(kwargs.get('synthesis_kwargs', {}))
    target['num_fp16_res'] = 4
    target['conv_clamp'] = 256
    if kwargs == net.init_kwargs:
        return net
    rebuilt = type(net)(*net.init_args, **kwargs).eval().requires_grad_(False)
    rebuilt.load_state_dict(net.state_dict(), strict=True)
    return rebuilt


# ----------------------------------------------------------------------------
# TensorFlow-era networks.
#
# A rule is (pattern over this package's parameter names, TensorFlow variable name with {0}, {1} for the pattern's
# groups, layout change).  Layout changes: 'T' matrix transpose (TensorFlow dense weights are [in, out]); 'conv'
# [kh, kw, in, out] -> [out, in, kh, kw]; 'conv_flip' the same after reversing both spatial axes (TensorFlow's
# up-sampling layers correlate where these convolve); '+1' the modulation bias (TensorFlow adds one to the style after
# the affine map, here the bias starts at one); 'row' / 'plane' drop one / two leading unit axes.

_LAYOUT = {
    None: lambda a: a,
    'T': lambda a: a.transpose(),
    'conv': lambda a: a.transpose(3, 2, 0, 1),
    'conv_flip': lambda a: a[::-1, ::-1].transpose(3, 2, 0, 1),
    '+1': lambda a: a + 1,
    'row': lambda a: a[0],
    'plane': lambda a: a[0, 0],
}


def _tf_layer(block, conv):
    """TensorFlow scope of `synthesis.b<block>.conv<conv>`: the 4x4 block has one layer, 'Conv'; above it the first
    layer of a block up-samples."""
    if int(block) == 4:
        return f'synthesis/4x4/Conv'
    return f'synthesis/{block}x{block}/' + ('Conv0_up' if int(conv) == 0 else 'Conv1')


def _tf_noise(block, conv):
    """Noise inputs are numbered through the layers from the 4x4 one (legacy.py:190,196,202)."""
    return f'synthesis/noise{2 * (int(block).bit_length() - 1) - 5 + int(conv)}'


_G_RULES = [
    (r'mapping\.w_avg', lambda: 'dlatent_avg', None),
    (r'mapping\.embed\.weight', lambda: 'mapping/LabelEmbed/weight', 'T'),
    (r'mapping\.embed\.bias', lambda: 'mapping/LabelEmbed/bias', None),
    (r'mapping\.fc(\d+)\.weight', lambda i: f'mapping/Dense{i}/weight', 'T'),
    (r'mapping\.fc(\d+)\.bias', lambda i: f'mapping/Dense{i}/bias', None),
    (r'synthesis\.b4\.const', lambda: 'synthesis/4x4/Const/const', 'row'),
    (r'synthesis\.b(\d+)\.conv(\d)\.weight', lambda r, c: _tf_layer(r, c) + '/weight', lambda r, c: 'conv_flip' if int(r) > 4 and int(c) == 0 else 'conv'),
    (r'synthesis\.b(\d+)\.conv(\d)\.bias', lambda r, c: _tf_layer(r, c) + '/bias', None),
    (r'synthesis\.b(\d+)\.conv(\d)\.noise_const', _tf_noise, 'plane'),
    (r'synthesis\.b(\d+)\.conv(\d)\.noise_strength', lambda r, c: _tf_layer(r, c) + '/noise_strength', None),
    (r'synthesis\.b(\d+)\.conv(\d)\.affine\.weight', lambda r, c: _tf_layer(r, c) + '/mod_weight', 'T'),
    (r'synthesis\.b(\d+)\.conv(\d)\.affine\.bias', lambda r, c: _tf_layer(r, c) + '/mod_bias', '+1'),
    (r'synthesis\.b(\d+)\.torgb\.weight', lambda r: f'synthesis/{r}x{r}/ToRGB/weight', 'conv'),
    (r'synthesis\.b(\d+)\.torgb\.bias', lambda r: f'synthesis/{r}x{r}/ToRGB/bias', None),
    (r'synthesis\.b(\d+)\.torgb\.affine\.weight', lambda r: f'synthesis/{r}x{r}/ToRGB/mod_weight', 'T'),
    (r'synthesis\.b(\d+)\.torgb\.affine\.bias', lambda r: f'synthesis/{r}x{r}/ToRGB/mod_bias', '+1'),
    (r'synthesis\.b(\d+)\.skip\.weight', lambda r: f'synthesis/{r}x{r}/Skip/weight', 'conv_flip'),
    (r'.*\.resample_filter', None, None),
]

_D_RULES = [
    (r'b(\d+)\.fromrgb\.weight', lambda r: f'{r}x{r}/FromRGB/weight', 'conv'),
    (r'b(\d+)\.fromrgb\.bias', lambda r: f'{r}x{r}/FromRGB/bias', None),
    (r'b(\d+)\.conv(\d)\.weight', lambda r, c: f'{r}x{r}/Conv{c}' + ('_down' if int(c) else '') + '/weight', 'conv'),
    (r'b(\d+)\.conv(\d)\.bias', lambda r, c: f'{r}x{r}/Conv{c}' + ('_down' if int(c) else '') + '/bias', None),
    (r'b(\d+)\.skip\.weight', lambda r: f'{r}x{r}/Skip/weight', 'conv'),
    (r'mapping\.embed\.weight', lambda: 'LabelEmbed/weight', 'T'),
    (r'mapping\.embed\.bias', lambda: 'LabelEmbed/bias', None),
    (r'mapping\.fc(\d+)\.weight', lambda i: f'Mapping{i}/weight', 'T'),
    (r'mapping\.fc(\d+)\.bias', lambda i: f'Mapping{i}/bias', None),
    (r'b4\.conv\.weight', lambda: '4x4/Conv/weight', 'conv'),
    (r'b4\.conv\.bias', lambda: '4x4/Conv/bias', None),
    (r'b4\.fc\.weight', lambda: '4x4/Dense0/weight', 'T'),
    (r'b4\.fc\.bias', lambda: '4x4/Dense0/bias', None),
    (r'b4\.out\.weight', lambda: 'Output/weight', 'T'),
    (r'b4\.out\.bias', lambda: 'Output/bias', None),
    (r'.*\.resample_filter', None, None),
]


def _fill_from_tf(net, tf_params, rules):
    """Every parameter and buffer of `net` must be named by a rule (legacy.py:95-112); rules without a source leave the
    constructor's value."""
    tensors = collections.OrderedDict(list(net.named_parameters()) + list(net.named_buffers()))
    with torch.no_grad():
        for name, tensor in tensors.items():
            for pattern, source, layout in rules:
                match = re.fullmatch(pattern, name)
                if match is None:
                    continue
                if source is not None:
                    tf_name = source(*match.groups())
                    if tf_name not in tf_params:
                        raise KeyError(f'{name}: the TensorFlow network has no variable {tf_name!r}')
                    change = layout(*match.groups()) if callable(layout) else layout
                    value = np.array(_LAYOUT[change](np.asarray(tf_params[tf_name])), order='C')
                    if tuple(value.shape) != tuple(tensor.shape):
                        raise ValueError(f'{name}: {tf_name} has shape {list(value.shape)}, expected {list(tensor.shape)}')
                    tensor.copy_(torch.from_numpy(value))
                break
            else:
                raise KeyError(f'no TensorFlow variable is known for {name} {list(tensor.shape)}')


class _StaticKwargs:
    """Reads a TensorFlow network's `static_kwargs` and remembers which were asked for, so that an argument this
    conversion does not understand is an error instead of a silently different network (legacy.py:157-164)."""

    def __init__(self, net):
        if getattr(net, 'version', 0) < 4:
            raise ValueError('TensorFlow pickle version too low')
        self.values = dict(getattr(net, 'static_kwargs', {}) or {})
        self.asked = set()

    def __call__(self, name, default=None, none=None):
        self.asked.add(name)
        value = self.values.get(name, default)
        return none if value is None else value

    def check(self, *ignored):
        unknown = sorted(set(self.values) - self.asked - set(ignored))
        if unknown:
            raise ValueError('Unknown TensorFlow kwarg', unknown[0])


def _per_lod_layers(tf_params, scope, resolution):
    """The progressive-growing ('orig') graphs name their image layers by level of detail, `<scope>_lod<n>` with
    n = 0 at full resolution; rename them to the per-resolution scopes the tables use.  Returns whether any exist."""
    found = False
    for name in list(tf_params):
        match = re.fullmatch(rf'(.*/)?{scope}_lod(\d+)/(.*)', name)
        if match:
            r = resolution >> int(match.group(2))
            tf_params[f'{match.group(1) or ""}{r}x{r}/{scope}/{match.group(3)}'] = tf_params[name]
            found = True
    return found


def convert_tf_generator(tf_G):
    """This package's `Generator` with the weights of a TensorFlow StyleGAN2 / StyleGAN2-ADA generator record
    (legacy.py:117-213)."""
    from .training import networks
    kw = _StaticKwargs(tf_G)
    kwargs = dict(
        z_dim=kw('latent_size', 512), c_dim=kw('label_size', 0), w_dim=kw('dlatent_size', 512),
        img_resolution=kw('resolution', 1024), img_channels=kw('num_channels', 3),
        mapping_kwargs=dict(num_layers=kw('mapping_layers', 8), embed_features=kw('label_fmaps', None),
                            layer_features=kw('mapping_fmaps', None), activation=kw('mapping_nonlinearity', 'lrelu'),
                            lr_multiplier=kw('mapping_lrmul', 0.01), w_avg_beta=kw('w_avg_beta', 0.995, none=1)),
        synthesis_kwargs=dict(channel_base=kw('fmap_base', 16384) * 2, channel_max=kw('fmap_max', 512),
                              num_fp16_res=kw('num_fp16_res', 0), conv_clamp=kw('conv_clamp', None),
                              architecture=kw('architecture', 'skip'), resample_filter=kw('resample_kernel', [1, 3, 3, 1]),
                              activation=kw('nonlinearity', 'lrelu'), use_noise=kw('use_noise', True)))
    kw.check('truncation_psi', 'truncation_cutoff', 'style_mixing_prob', 'structure')
    tf_params = tf_G.parameters()
    if _per_lod_layers(tf_params, 'ToRGB', kwargs['img_resolution']):
        kwargs['synthesis_kwargs']['architecture'] = 'orig'
    G = networks.Generator(**kwargs).eval().requires_grad_(False)
    _fill_from_tf(G, tf_params, _G_RULES)
    return G


def convert_tf_discriminator(tf_D):
    """This package's `Discriminator` from a TensorFlow discriminator record (legacy.py:218-299)."""
    from .training import networks
    kw = _StaticKwargs(tf_D)
    kwargs = dict(
        c_dim=kw('label_size', 0), img_resolution=kw('resolution', 1024), img_channels=kw('num_channels', 3),
        architecture=kw('architecture', 'resnet'), channel_base=kw('fmap_base', 16384) * 2, channel_max=kw('fmap_max', 512),
        num_fp16_res=kw('num_fp16_res', 0), conv_clamp=kw('conv_clamp', None), cmap_dim=kw('mapping_fmaps', None),
        block_kwargs=dict(activation=kw('nonlinearity', 'lrelu'), resample_filter=kw('resample_kernel', [1, 3, 3, 1]),
                          freeze_layers=kw('freeze_layers', 0)),
        mapping_kwargs=dict(num_layers=kw('mapping_layers', 0), embed_features=kw('mapping_fmaps', None),
                            layer_features=kw('mapping_fmaps', None), activation=kw('nonlinearity', 'lrelu'),
                            lr_multiplier=kw('mapping_lrmul', 0.1)),
        epilogue_kwargs=dict(mbstd_group_size=kw('mbstd_group_size', None), mbstd_num_channels=kw('mbstd_num_features', 1),
                             activation=kw('nonlinearity', 'lrelu')))
    kw.check('structure')
    tf_params = tf_D.parameters()
    if _per_lod_layers(tf_params, 'FromRGB', kwargs['img_resolution']):
        kwargs['architecture'] = 'orig'
    D = networks.Discriminator(**kwargs).eval().requires_grad_(False)
    _fill_from_tf(D, tf_params, _D_RULES)
    return D


# ----------------------------------------------------------------------------
# Writer.

_PERSISTENCE_VERSION = 6        # torch_utils/persistence.py:27; the reference's reader asserts it (persistence.py:189)


class _PersistentRecord:
    """Pickles as `torch_utils.persistence._reconstruct_persistent_obj(meta)` (persistence.py:118-126)."""

    def __init__(self, meta, reconstruct):
        self.meta, self.reconstruct = meta, reconstruct

    def __reduce__(self):
        return self.reconstruct, (self.meta,), None


@contextlib.contextmanager
def _reference_globals():
    """The two globals a snapshot names - `torch_utils.persistence._reconstruct_persistent_obj` and
    `dnnlib.util.EasyDict` - importable under those names while the pickle is written (pickle stores a global by name
    and checks that the name resolves).  Modules already imported under those names are used as they are."""
    added = []

    def module(name):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            added.append(name)
        return sys.modules[name]

    module('torch_utils')
    persistence = module('torch_utils.persistence')
    if not hasattr(persistence, '_reconstruct_persistent_obj'):
        def _reconstruct_persistent_obj(meta):
            return _record(meta)
        _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
        _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
        persistence._reconstruct_persistent_obj = _reconstruct_persistent_obj
    module('dnnlib')
    util = module('dnnlib.util')
    if not hasattr(util, 'EasyDict'):
        util.EasyDict = type('EasyDict', (dict,), {'__module__': 'dnnlib.util', '__qualname__': 'EasyDict'})
    try:
        yield persistence._reconstruct_persistent_obj, util.EasyDict
    finally:
        for name in added:
            sys.modules.pop(name, None)


def _clean_tensor(t):
    return t.detach().to('cpu', copy=True).contiguous()


def _plain_value(value, where, easy):
    """`value` as data a pickle of the reference's would hold: tensors detached on the CPU without the caches this
    package attaches to them, dictionaries as the reference's attribute dictionaries."""
    if value is None or isinstance(value, (bool, int, float, str, bytes, np.generic, np.ndarray, torch.dtype, torch.Size)):
        return value
    if isinstance(value, torch.Tensor):
        return _clean_tensor(value)
    if isinstance(value, dict):
        return easy((k, _plain_value(v, where, easy)) for k, v in value.items())
    if isinstance(value, (list, tuple, set, frozenset)):
        return type(value)(_plain_value(v, where, easy) for v in value)
    raise TypeError(f'{where} holds a {type(value).__name__}: only data goes into a snapshot')


def _module_record(module, where, reconstruct, easy):
    from .torch_utils import persistence
    if not persistence.records_its_init(module):
        raise TypeError(f'{where}: {type(module).__name__} does not record its constructor arguments')
    state = collections.OrderedDict()
    for key, value in module.__dict__.items():
        if key.startswith('_sg'):                              # derived-tensor caches of this package
            continue
        if key == '_parameters':
            state[key] = collections.OrderedDict(
                (n, None if p is None else torch.nn.Parameter(_clean_tensor(p), requires_grad=p.requires_grad)) for n, p in value.items())
        elif key == '_buffers':
            state[key] = collections.OrderedDict((n, None if b is None else _clean_tensor(b)) for n, b in value.items())
        elif key == '_modules':
            state[key] = collections.OrderedDict(
                (n, None if m is None else _module_record(m, f'{where}.{n}', reconstruct, easy)) for n, m in value.items())
        elif key.endswith('_hooks') or key.endswith('_hooks_with_kwargs') or key.endswith('_hooks_always_called'):
            state[key] = collections.OrderedDict()             # hooks are code
        elif key == '_init_kwargs':
            state[key] = dict(_plain_value(value, f'{where}.{key}', easy))      # (the reference keeps a plain dict of attribute dicts)
        else:
            state[key] = _plain_value(value, f'{where}.{key}', easy)
    source = '.'.join(type(module).__module__.split('.')[-2:])   # 'training.networks', 'training.augment'
    meta = dict(type='class', version=_PERSISTENCE_VERSION, module_src=f'from {source} import {type(module).__name__}\n',
                class_name=type(module).__name__, state=state)
    return _PersistentRecord(meta, reconstruct)


def save_network_pkl(f, data):
    """Write `data` - dict(G, D, G_ema[, augment_pipe][, training_set_kwargs]) of this package's networks - in the
    layout of the reference's snapshots (training/training_loop.py:586-601): the reference's
    `legacy.load_network_pkl` and this module's read it.  The networks are written as they are, on the CPU, with
    their training flags and `requires_grad` settings; the reference snapshots copies in eval mode without gradients
    (`copy.deepcopy(module).eval().requires_grad_(False)`), which is left to the caller."""
    with _reference_globals() as (reconstruct, easy):
        out = {}
        for key, value in data.items():
            if isinstance(value, torch.nn.Module):
                out[key] = _module_record(value, key, reconstruct, easy)
            elif value is None:
                out[key] = None
            else:
                out[key] = _plain_value(value, key, easy)
                if isinstance(out[key], easy):
                    out[key] = dict(out[key])       # training_set_kwargs is written as dict(...) (training_loop.py:589)
        pickle.dump(out, f)
