"""Data path in front of the training step, CPU part: the oracle and the host-side dataset / sampler classes against
fixtures written by the reference's own dataset.py and misc.py (tools/gen_golden.py data_path)."""
import os
import pickle

import numpy as np
import pytest

import golden_io
from oracle import data as odata
from ga_gan_amd.torch_utils import misc
from ga_gan_amd.training import dataset as ds_mod

CASES = golden_io.load('data_path')
DATASETS = [c for c in CASES if c.meta['op'] == 'dataset']
SAMPLERS = [c for c in CASES if c.meta['op'] == 'sampler']


def build_folder(tmp_path, m):
    images = odata.synthetic_images(m['count'], m['channels'], m['resolution'], m['seed'])
    labels = None
    if m['labels'] == 'int':
        labels = [int(v) for v in np.random.RandomState(m['seed'] + 100).randint(0, 4, size=m['count'])]
    elif m['labels'] == 'float':
        labels = np.random.RandomState(m['seed'] + 100).randn(m['count'], 3).astype(np.float32).round(3)
    path = os.path.join(tmp_path, m['folder'] + ('.zip' if m['zip'] else ''))
    if not m['zip']:
        os.makedirs(path, exist_ok=True)
    odata.write_image_folder(path, images, labels, as_zip=m['zip'])
    return path, images, labels


@pytest.mark.parametrize('case', DATASETS, ids=lambda c: c.meta['folder'] + '-' + '_'.join(f'{k}{v}' for k, v in c.meta['kwargs'].items()))
def test_dataset_matches_reference(case, tmp_path):
    m = case.meta
    path, images, labels = build_folder(str(tmp_path), m)
    ds = ds_mod.ImageFolderDataset(path=path, resolution=m['resolution'], **m['kwargs'])
    assert ds.name == m['name'] and len(ds) == m['length']
    assert ds.image_shape == m['image_shape'] and ds.label_shape == m['label_shape'] and ds.label_dim == m['label_dim']
    assert ds.has_labels == m['has_labels'] and ds.has_onehot_labels == m['has_onehot_labels']
    assert ds.num_channels == m['channels'] and ds.resolution == m['resolution']
    want_images, want_labels = case.np('images'), case.np('item_labels')
    for i in range(len(ds)):
        image, label = ds[i]
        assert image.dtype == np.uint8 and np.array_equal(image, want_images[i])
        assert label.dtype == np.float32 and np.array_equal(label, want_labels[i])
        assert np.array_equal(ds.get_label(i), case.np('get_label')[i])
        d = ds.get_details(i)
        assert [d['raw_idx'], d['xflip']] == m['details'][i]
    # the oracle's statement of which raw images are exposed, and of the mirror
    raw, flip = odata.subset_indices(m['count'], m['kwargs'].get('max_size'), m['kwargs'].get('xflip', False), m['kwargs'].get('random_seed', 0))
    assert [[int(r), bool(f)] for r, f in zip(raw, flip)] == m['details']
    picked = odata.images_to_float(images, raw, flip)
    assert np.array_equal(picked, want_images.astype(np.float32) / np.float32(127.5) - 1)
    # survives pickling (DataLoader workers) and reads again afterwards
    clone = pickle.loads(pickle.dumps(ds))
    assert np.array_equal(clone[0][0], want_images[0])
    ds.close()
    clone.close()


def test_dataset_errors(tmp_path):
    with pytest.raises(IOError, match='directory or zip'):
        ds_mod.ImageFolderDataset(path=str(tmp_path / 'missing.txt'))
    empty = tmp_path / 'empty'
    empty.mkdir()
    with pytest.raises(IOError, match='No image files'):
        ds_mod.ImageFolderDataset(path=str(empty))
    m = DATASETS[0].meta
    path, _, _ = build_folder(str(tmp_path), m)
    with pytest.raises(IOError, match='resolution'):
        ds_mod.ImageFolderDataset(path=path, resolution=m['resolution'] * 2)


class Sized:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


@pytest.mark.parametrize('case', SAMPLERS, ids=lambda c: 'n{size}r{num_replicas}s{seed}w{window_size}{shuffle}'.format(**c.meta))
def test_sampler_matches_reference(case):
    m = case.meta
    want = case.np('order')
    for rank in range(m['num_replicas']):
        kw = dict(rank=rank, num_replicas=m['num_replicas'], shuffle=m['shuffle'], seed=m['seed'], window_size=m['window_size'])
        assert np.array_equal(odata.sampler_indices(m['size'], want.shape[1], **kw), want[rank])
        sampler = misc.InfiniteSampler(Sized(m['size']), **kw)
        it = iter(sampler)
        assert [int(next(it)) for _ in range(want.shape[1])] == want[rank].tolist()
        first, second = sampler.take(40), sampler.take(24)        # one continuing pass, cut anywhere
        assert np.array_equal(np.concatenate([first, second]), want[rank])


def test_sampler_ranks_partition_the_stream():
    """All ranks together visit what a single rank would: rank r takes every num_replicas-th visit."""
    whole = odata.sampler_indices(23, 120, seed=4)
    for r in range(3):
        assert np.array_equal(odata.sampler_indices(23, 40, rank=r, num_replicas=3, seed=4), whole[r::3])
    with pytest.raises(ValueError):
        misc.InfiniteSampler(Sized(0))
    with pytest.raises(ValueError):
        misc.InfiniteSampler(Sized(4), rank=2, num_replicas=2)


def test_oracle_to_float_every_byte():
    case = [c for c in CASES if c.meta['op'] == 'to_float'][0]
    u8 = case.np('u8').reshape(1, 1, 16, 16)
    assert np.array_equal(odata.images_to_float(u8).reshape(-1), case.np('f32'))
    mirrored = odata.images_to_float(u8, index=[0, 0], flip=[0, 1])
    assert np.array_equal(mirrored[1], mirrored[0][:, :, ::-1])
