#!/usr/bin/env python3
"""Fixtures for the snapshot reader / converter (`ga_gan_amd.legacy`), made by running the REFERENCE in the build
container (needs /root/reference; CPU only):

tests/golden/snapshot_ref.pkl    a network snapshot pickled by the reference itself - its persistent classes, its
                                 `__reduce__`, the dict of training/training_loop.py:586-601 - for a small generator,
                                 discriminator, averaged generator and augmentation pipe.  Before pickling, the source
                                 text each class would embed is replaced by an empty string: the file holds class
                                 names, attribute dictionaries and tensors, no source.
tests/golden/snapshot_ref.npz    what the reference had next to it: constructor arguments, every state tensor, and the
                                 outputs of the mapping network, the generator and the discriminator on seeded inputs.
tests/golden/snapshot_tf.pkl     a synthetic TensorFlow-era pickle written HERE (random arrays under the TensorFlow
                                 variable names and layouts of a small StyleGAN2 generator / discriminator, in objects
                                 that pickle as `dnnlib.tflib.network.Network`).
tests/golden/snapshot_tf.npz     the constructor arguments and state tensors of the networks the reference's
                                 `legacy.load_network_pkl` made of that pickle.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_snapshots.py
"""
import collections
import copy
import io
import json
import os
import pickle
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402   (puts the reference on sys.path and repairs its SynthesisBlock wiring)

import legacy as ref_legacy  # noqa: E402
from training import augment as ref_augment  # noqa: E402

OUT = gg.OUT


def _plain(value):
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    if isinstance(value, np.generic):
        return value.item()
    return value


def _blank_embedded_source(*modules):
    for root in modules:
        for m in root.modules():
            if hasattr(type(m), '_orig_module_src'):
                type(m)._orig_module_src = ''


def gen_reference_snapshot():
    gen = torch.Generator().manual_seed(901)
    G = gg.make_G(32, channel_base=512, channel_max=32, seed=31)
    gg.randomize(G, gen)
    G_ema = copy.deepcopy(G)
    with torch.no_grad():
        for p in G_ema.parameters():
            p.mul_(0.75)
        G.mapping.w_avg.copy_(torch.randn(G.mapping.w_avg.shape, generator=gen) * 0.1)
    D = gg.make_D(32, channel_base=512, channel_max=32, mbstd=2, seed=32)
    gg.randomize(D, gen)
    pipe = ref_augment.AugmentPipe(xflip=1, rotate90=1, xint=1, scale=1, brightness=1)
    pipe.p.copy_(torch.as_tensor(0.3))
    training_set_kwargs = dict(class_name='training.dataset.ImageFolderDataset', path='few_shot.zip', use_labels=False,
                               max_size=10, xflip=False, resolution=32)

    snapshot = dict(training_set_kwargs=dict(training_set_kwargs))          # as training_loop.py:589-595 builds it
    for name, module in [('G', G), ('D', D), ('G_ema', G_ema), ('augment_pipe', pipe)]:
        snapshot[name] = copy.deepcopy(module).eval().requires_grad_(False).cpu()
    _blank_embedded_source(G, D, G_ema, pipe)      # (after the deep copies: they run the same reduce / rebuild cycle)
    blob = pickle.dumps(snapshot)
    assert b'import ' not in blob and b'def forward' not in blob
    with open(os.path.join(OUT, 'snapshot_ref.pkl'), 'wb') as f:
        f.write(blob)

    z = torch.randn(3, G.z_dim, generator=gen)
    c = torch.zeros(3, 0)
    img = torch.randn(4, 3, 32, 32, generator=gen).clamp_(-1, 1)
    arrays = {}
    meta = dict(training_set_kwargs=training_set_kwargs)
    with torch.no_grad():
        for name, module in snapshot.items():
            if not isinstance(module, torch.nn.Module):
                continue
            meta[name + '_init_kwargs'] = _plain(module.init_kwargs)
            meta[name + '_state_names'] = list(module.state_dict())
            arrays.update(gg.state_arrays(module, prefix=name + '/'))
        for name in ('G', 'G_ema'):
            net = snapshot[name]
            arrays[name + '.ws'] = gg.npy(net.mapping(z, c))
            arrays[name + '.ws_truncated'] = gg.npy(net.mapping(z, c, truncation_psi=0.5, truncation_cutoff=4))
            arrays[name + '.img'] = gg.npy(net(z, c, noise_mode='const'))
        arrays['D.logits'] = gg.npy(snapshot['D'](img, torch.zeros(4, 0)))
    np.savez_compressed(os.path.join(OUT, 'snapshot_ref.npz'), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
                        z=gg.npy(z), img=gg.npy(img), **arrays)
    print('snapshot_ref.pkl', len(blob), 'bytes; D logits', arrays['D.logits'].flatten().tolist())


# ----------------------------------------------------------------------------
# TensorFlow-era pickle.  The variable names, shapes and layouts below describe the TensorFlow StyleGAN2 graphs
# (config F: skip generator, residual discriminator); they are written out here independently of the converter.

class _TFNet:
    """Pickles as a `dnnlib.tflib.network.Network` whose state is the dictionary that class's `__getstate__` returns."""

    def __init__(self, name, static_kwargs, variables, components=None):
        self.state = dict(version=4, name=name, static_kwargs=static_kwargs, components=components or {},
                          build_module_src='raise SystemExit("TensorFlow graph source was executed")\n',
                          build_func_name=name + '_main', variables=variables)

    def __getstate__(self):
        return self.state


def _tf_generator(rng, res, fmaps, w_dim, z_dim, layers):
    def arr(*shape):
        return rng.standard_normal(shape).astype(np.float32) * 0.3

    mapping = [(f'Dense{i}/weight', arr(z_dim if i == 0 else w_dim, w_dim)) for i in range(layers)]
    mapping += [(f'Dense{i}/bias', arr(w_dim)) for i in range(layers)]
    synthesis = []

    def layer(scope, cin, cout, k=3, noise=True):
        synthesis.extend([(scope + '/weight', arr(k, k, cin, cout)), (scope + '/bias', arr(cout)),
                          (scope + '/mod_weight', arr(w_dim, cin)), (scope + '/mod_bias', arr(cin))])
        if noise:
            synthesis.append((scope + '/noise_strength', arr()))

    synthesis.append(('4x4/Const/const', arr(1, fmaps, 4, 4)))
    layer('4x4/Conv', fmaps, fmaps)
    layer('4x4/ToRGB', fmaps, 3, k=1, noise=False)
    synthesis.append(('noise0', arr(1, 1, 4, 4)))
    n = 1
    r = 8
    while r <= res:
        layer(f'{r}x{r}/Conv0_up', fmaps, fmaps)
        layer(f'{r}x{r}/Conv1', fmaps, fmaps)
        layer(f'{r}x{r}/ToRGB', fmaps, 3, k=1, noise=False)
        synthesis.extend([(f'noise{n}', arr(1, 1, r, r)), (f'noise{n + 1}', arr(1, 1, r, r))])
        n += 2
        r *= 2
    kwargs = dict(latent_size=z_dim, dlatent_size=w_dim, resolution=res, num_channels=3, mapping_layers=layers,
                  fmap_base=fmaps * res // 2, fmap_max=fmaps, label_size=0, truncation_psi=0.5, style_mixing_prob=0.9)
    parts = collections.OrderedDict(mapping=_TFNet('G_mapping', {}, mapping), synthesis=_TFNet('G_synthesis', {}, synthesis))
    return _TFNet('G', kwargs, [('dlatent_avg', arr(w_dim))], parts)


def _tf_discriminator(rng, res, fmaps):
    def arr(*shape):
        return rng.standard_normal(shape).astype(np.float32) * 0.3

    variables = [(f'{res}x{res}/FromRGB/weight', arr(1, 1, 3, fmaps)), (f'{res}x{res}/FromRGB/bias', arr(fmaps))]
    r = res
    while r > 4:
        variables += [(f'{r}x{r}/Conv0/weight', arr(3, 3, fmaps, fmaps)), (f'{r}x{r}/Conv0/bias', arr(fmaps)),
                      (f'{r}x{r}/Conv1_down/weight', arr(3, 3, fmaps, fmaps)), (f'{r}x{r}/Conv1_down/bias', arr(fmaps)),
                      (f'{r}x{r}/Skip/weight', arr(1, 1, fmaps, fmaps))]
        r //= 2
    variables += [('4x4/Conv/weight', arr(3, 3, fmaps + 1, fmaps)), ('4x4/Conv/bias', arr(fmaps)),
                  ('4x4/Dense0/weight', arr(fmaps * 16, fmaps)), ('4x4/Dense0/bias', arr(fmaps)),
                  ('Output/weight', arr(fmaps, 1)), ('Output/bias', arr(1))]
    kwargs = dict(resolution=res, num_channels=3, fmap_base=fmaps * res // 2, fmap_max=fmaps, label_size=0, mbstd_group_size=2)
    return _TFNet('D', kwargs, variables)


def gen_tf_snapshot():
    rng = np.random.default_rng(77)
    nets = (_tf_generator(rng, 16, 16, 24, 20, 3), _tf_discriminator(rng, 16, 16), _tf_generator(rng, 16, 16, 24, 20, 3))
    module = types.ModuleType('dnnlib.tflib.network')
    _TFNet.__module__, _TFNet.__qualname__, _TFNet.__name__ = 'dnnlib.tflib.network', 'Network', 'Network'
    module.Network = _TFNet
    sys.modules.setdefault('dnnlib.tflib', types.ModuleType('dnnlib.tflib'))
    sys.modules['dnnlib.tflib.network'] = module
    try:
        blob = pickle.dumps(nets, protocol=2)
    finally:
        del sys.modules['dnnlib.tflib.network']
    with open(os.path.join(OUT, 'snapshot_tf.pkl'), 'wb') as f:
        f.write(blob)

    with gg.quiet():
        data = ref_legacy.load_network_pkl(io.BytesIO(blob))
    arrays = {}
    meta = {}
    for name in ('G', 'D', 'G_ema'):
        meta[name + '_init_kwargs'] = _plain(data[name].init_kwargs)
        meta[name + '_state_names'] = list(data[name].state_dict())
        arrays.update(gg.state_arrays(data[name], prefix=name + '/'))
    np.savez_compressed(os.path.join(OUT, 'snapshot_tf.npz'), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print('snapshot_tf.pkl', len(blob), 'bytes;', len(arrays), 'tensors converted by the reference')


if __name__ == '__main__':
    torch.set_num_threads(8)
    gen_reference_snapshot()
    gen_tf_snapshot()
