"""Data path on the GPU: `sg_images_to_float` through the C ABI against the CPU oracle (bit-exact in float32, one
rounding to the 16-bit types), and the device-resident image set against the reference's dataset fixtures."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import data as odata
from test_data_path import DATASETS, build_folder

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def plugin():
    from ga_gan_amd import images_plugin
    return images_plugin


def test_every_byte_value_matches_reference():
    case = [c for c in golden_io.load('data_path') if c.meta['op'] == 'to_float'][0]
    u8 = case.t('u8', DEV).reshape(1, 1, 16, 16)
    out = plugin().images_to_float(u8)
    assert out.dtype == torch.float32 and torch.equal(out.cpu().reshape(-1), case.t('f32'))
    narrow = plugin().images_to_float(u8.reshape(1, 1, 32, 8))       # rows shorter than a packet: the per-pixel kernel
    assert torch.equal(narrow.cpu().reshape(-1), case.t('f32'))


@pytest.mark.parametrize('shape', [(5, 3, 16, 16), (4, 3, 64, 48), (3, 1, 7, 13), (6, 3, 9, 32), (2, 3, 256, 256)], ids=str)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gather_and_mirror_vs_oracle(shape, dtype):
    rs = np.random.RandomState(sum(shape))
    images = rs.randint(0, 256, size=shape).astype(np.uint8)
    n = 11
    index = rs.randint(0, shape[0], size=n)
    flip = rs.randint(0, 2, size=n).astype(np.uint8)
    want = torch.from_numpy(odata.images_to_float(images, index, flip)).to(dtype)
    d_images = torch.from_numpy(images).to(DEV)
    got = plugin().images_to_float(d_images, torch.from_numpy(index).to(DEV), torch.from_numpy(flip).to(DEV), dtype=dtype)
    assert got.dtype == dtype and torch.equal(got.cpu(), want)
    # no index / no mirror; bool mirror flags; an empty batch
    assert torch.equal(plugin().images_to_float(d_images, dtype=dtype).cpu(), torch.from_numpy(odata.images_to_float(images)).to(dtype))
    got = plugin().images_to_float(d_images, torch.from_numpy(index).to(DEV), torch.from_numpy(flip).to(DEV).bool(), dtype=dtype)
    assert torch.equal(got.cpu(), want)
    assert plugin().images_to_float(d_images, torch.zeros([0], dtype=torch.int64, device=DEV)).shape == (0,) + shape[1:]


def test_unaligned_view_and_bad_index():
    rs = np.random.RandomState(5)
    images = rs.randint(0, 256, size=(4, 3, 16, 16)).astype(np.uint8)
    flat = torch.zeros(images.size + 3, dtype=torch.uint8, device=DEV)
    flat[3:] = torch.from_numpy(images).to(DEV).reshape(-1)
    view = flat[3:].reshape(images.shape)                       # base address not a multiple of 16
    got = plugin().images_to_float(view, torch.tensor([3, 1], device=DEV), torch.tensor([1, 0], dtype=torch.uint8, device=DEV))
    assert torch.equal(got.cpu(), torch.from_numpy(odata.images_to_float(images, [3, 1], [1, 0])))
    out = plugin().images_to_float(torch.from_numpy(images).to(DEV), torch.tensor([0, 4, -1, 2], device=DEV))
    assert torch.isnan(out[1]).all() and torch.isnan(out[2]).all()          # out-of-range entries are flagged, not read
    assert torch.equal(out[[0, 3]].cpu(), torch.from_numpy(odata.images_to_float(images, [0, 2])))


def test_errors():
    p = plugin()
    with pytest.raises(RuntimeError, match='CUDA'):
        p.images_to_float(torch.zeros([1, 3, 4, 4], dtype=torch.uint8))
    with pytest.raises(RuntimeError, match='uint8'):
        p.images_to_float(torch.zeros([1, 3, 4, 4], device=DEV))
    with pytest.raises(RuntimeError, match='int64'):
        p.images_to_float(torch.zeros([1, 3, 4, 4], dtype=torch.uint8, device=DEV), torch.zeros([2], dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match='flip'):
        p.images_to_float(torch.zeros([1, 3, 4, 4], dtype=torch.uint8, device=DEV), torch.zeros([2], dtype=torch.int64, device=DEV),
                          torch.zeros([3], dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError, match='dtype'):
        p.images_to_float(torch.zeros([1, 3, 4, 4], dtype=torch.uint8, device=DEV), dtype=torch.float64)


def test_full_size_batch_properties():
    """At the bench size (32 x 3 x 1024 x 1024): mirroring twice is the identity, the range is [-1, 1], and the mean
    follows the mean byte."""
    g = torch.Generator(device=DEV).manual_seed(2)
    images = torch.randint(0, 256, [48, 3, 1024, 1024], dtype=torch.uint8, device=DEV, generator=g)
    index = torch.randint(0, 48, [32], device=DEV, generator=g)
    ones = torch.ones([32], dtype=torch.uint8, device=DEV)
    plain = plugin().images_to_float(images, index)
    mirrored = plugin().images_to_float(images, index, ones)
    assert torch.equal(mirrored.flip(3), plain)
    assert float(plain.min()) == -1.0 and float(plain.max()) == 1.0
    assert abs(float(plain.double().mean()) - (float(images[index].double().mean()) / 127.5 - 1)) < 1e-6
    assert torch.equal(plain, images[index].to(torch.float32).div(127.5).sub(1)) or \
        float((plain - (images[index].to(torch.float32) / 127.5 - 1)).abs().max()) <= 2 ** -23   # torch multiplies by 1/127.5 on the GPU


@pytest.mark.parametrize('case', DATASETS[2:5], ids=lambda c: c.meta['folder'] + '-' + '_'.join(c.meta['kwargs']))
def test_resident_images_match_reference_items(case, tmp_path):
    from ga_gan_amd.training import dataset as ds_mod
    m = case.meta
    path, _, _ = build_folder(str(tmp_path), m)
    ds = ds_mod.ImageFolderDataset(path=path, resolution=m['resolution'], **m['kwargs'])
    resident = ds_mod.ResidentImages(ds, DEV)
    assert len(resident) == len(ds)
    items = np.random.RandomState(1).randint(0, len(ds), size=13)
    images, labels = resident.batch(items)
    want = case.t('images')[items].to(torch.float32) / 127.5 - 1
    assert torch.equal(images.cpu(), want)
    assert torch.equal(labels.cpu(), case.t('item_labels')[items].reshape(13, -1))
    with pytest.raises(IndexError):
        resident.batch([len(ds)])
    # the stream follows the sampler: rank 1 of 2
    stream = resident.stream(batch=5, rank=1, num_replicas=2, seed=3)
    order = odata.sampler_indices(len(ds), 10, rank=1, num_replicas=2, seed=3)
    for k in range(2):
        got, _ = next(stream)
        assert torch.equal(got.cpu(), case.t('images')[order[5 * k:5 * k + 5]].to(torch.float32) / 127.5 - 1)
