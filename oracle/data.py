"""CPU restatement of the data path in front of the training step: which images a dataset exposes, the order a
rank draws them in, and the conversion of a uint8 batch to the network's input range.

TEST INFRASTRUCTURE: only tests/, tools/ and the smoke check import this module; the product path never does.
Pinned by tests/golden/data_path.npz, which tools/gen_golden.py writes by running the reference's own
`training/dataset.py` and `torch_utils/misc.py` on the folders `write_image_folder` below produces.
"""
import io
import json
import os
import zipfile

import numpy as np


def synthetic_images(count, channels, resolution, seed):
    """uint8 [count, channels, resolution, resolution], reproducible on any machine."""
    return np.random.RandomState(seed).randint(0, 256, size=[count, channels, resolution, resolution]).astype(np.uint8)


def image_names(count):
    """Relative file names in the layout the reference's dataset tool writes: one folder per thousand images."""
    return [f'{i // 1000:05d}/img{i:08d}.png' for i in range(count)]


def write_image_folder(path, images, labels=None, as_zip=False):
    """PNG files (+ dataset.json when labels are given: class indices or float vectors) as a directory or a zip."""
    import PIL.Image
    names = image_names(len(images))
    files = {}
    for name, img in zip(names, images):
        hwc = img.transpose(1, 2, 0)
        pil = PIL.Image.fromarray(hwc[:, :, 0], 'L') if hwc.shape[2] == 1 else PIL.Image.fromarray(hwc, 'RGB')
        buf = io.BytesIO()
        pil.save(buf, format='png', compress_level=1)
        files[name] = buf.getvalue()
    meta = dict(labels=None if labels is None else [[n, (l.tolist() if hasattr(l, 'tolist') else l)] for n, l in zip(names, labels)])
    files['dataset.json'] = json.dumps(meta).encode()
    if as_zip:
        with zipfile.ZipFile(path, 'w', compression=zipfile.ZIP_STORED) as z:
            for name, data in files.items():
                z.writestr(name, data)
    else:
        for name, data in files.items():
            full = os.path.join(path, name)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, 'wb') as f:
                f.write(data)
    return names


def subset_indices(count, max_size=None, xflip=False, random_seed=0):
    """(raw index, mirrored?) of every item a dataset of `count` files exposes (reference: dataset.py:39-49):
    a seeded random subset of max_size files in ascending order, then optionally everything a second time mirrored."""
    raw = np.arange(count, dtype=np.int64)
    if max_size is not None and count > max_size:
        np.random.RandomState(random_seed).shuffle(raw)
        raw = np.sort(raw[:max_size])
    flip = np.zeros(raw.size, dtype=np.uint8)
    if xflip:
        raw = np.tile(raw, 2)
        flip = np.concatenate([flip, np.ones_like(flip)])
    return raw, flip


def sampler_indices(size, take, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
    """First `take` dataset indices the sampler hands to `rank` (reference: misc.py:131-148): a seeded permutation
    walked round and round; after visiting position i its entry is swapped with one up to `window` places back;
    global position idx belongs to rank idx % num_replicas."""
    order = np.arange(size)
    rnd, window = None, 0
    if shuffle:
        rnd = np.random.RandomState(seed)
        rnd.shuffle(order)
        window = int(np.rint(order.size * window_size))
    out = []
    idx = 0
    while len(out) < take:
        i = idx % order.size
        if idx % num_replicas == rank:
            out.append(int(order[i]))
        if window >= 2:
            j = (i - rnd.randint(window)) % order.size
            order[i], order[j] = order[j], order[i]
        idx += 1
    return np.asarray(out, dtype=np.int64)


def onehot(labels, index):
    """Label of item `index`: class indices become one-hot float32 rows (reference: dataset.py:95-101)."""
    labels = np.asarray(labels)
    if labels.dtype == np.int64:
        out = np.zeros([int(labels.max()) + 1], dtype=np.float32)
        out[labels[index]] = 1
        return out
    return labels[index].astype(np.float32).copy()


def images_to_float(images, index=None, flip=None, dtype=np.float32):
    """out[b] = images[index[b]] (mirrored left-right where flip[b]) / 127.5 - 1 in float32 arithmetic with a true
    division (reference: dataset.py:90-92 for the mirror, training_loop.py:441 for the range)."""
    images = np.asarray(images)
    assert images.dtype == np.uint8 and images.ndim == 4
    if index is None:
        index = np.arange(images.shape[0])
    picked = images[np.asarray(index)]
    if flip is not None:
        mirrored = picked[:, :, :, ::-1]
        picked = np.where(np.asarray(flip).astype(bool)[:, None, None, None], mirrored, picked)
    out = picked.astype(np.float32) / np.float32(127.5) - np.float32(1)
    return out.astype(dtype)
