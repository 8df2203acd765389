"""Image sets for the training loop: the folder / zip reader with the reference's interface
(training/dataset.py:24-234: `Dataset`, `ImageFolderDataset`, same constructor arguments, properties and item
format), and `ResidentImages`, which decodes a set once into device memory and hands out float batches with one
kernel per batch (`images_plugin.images_to_float`).

Few-shot sets are tens of images and the source-domain sets a few tens of gigabytes of uint8 at most, against 288 GB
of HBM: keeping the decoded set on the device removes the per-step decode, host batch assembly and host-to-device
copy of the reference's DataLoader (training/training_loop.py:221-223) from the loop altogether.
"""
import json
import os
import zipfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import images_plugin
from ..torch_utils import misc


class Dataset(torch.utils.data.Dataset):
    """Items are (uint8 image [C, H, W], float32 label).  `max_size` keeps a seeded random subset of the raw
    images (ascending raw order), `xflip` then exposes every kept image a second time, mirrored."""

    def __init__(self, name, raw_shape, max_size=None, use_labels=False, xflip=False, random_seed=0):
        self._name = name
        self._raw_shape = list(raw_shape)
        self._use_labels = use_labels
        self._raw_labels = None
        self._label_shape = None
        keep = np.arange(self._raw_shape[0], dtype=np.int64)
        if max_size is not None and keep.size > max_size:
            np.random.RandomState(random_seed).shuffle(keep)
            keep = np.sort(keep[:max_size])
        mirrored = np.zeros(keep.size, dtype=np.uint8)
        if xflip:
            keep, mirrored = np.tile(keep, 2), np.concatenate([mirrored, np.ones_like(mirrored)])
        self._raw_idx, self._xflip = keep, mirrored

    # -- to be provided by the storage back end --------------------------------
    def _load_raw_image(self, raw_idx):
        raise NotImplementedError

    def _load_raw_labels(self):
        raise NotImplementedError

    def close(self):
        pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        return dict(self.__dict__, _raw_labels=None)

    # -- items ------------------------------------------------------------------
    def _get_raw_labels(self):
        if self._raw_labels is None:
            labels = self._load_raw_labels() if self._use_labels else None
            if labels is None:
                labels = np.zeros([self._raw_shape[0], 0], dtype=np.float32)
            if not isinstance(labels, np.ndarray) or labels.shape[0] != self._raw_shape[0]:
                raise ValueError('labels do not match the images')
            if labels.dtype == np.int64:
                if labels.ndim != 1 or np.any(labels < 0):
                    raise ValueError('class labels must be a vector of non-negative integers')
            elif labels.dtype != np.float32:
                raise ValueError(f'labels must be int64 class indices or float32 vectors, not {labels.dtype}')
            self._raw_labels = labels
        return self._raw_labels

    def __len__(self):
        return self._raw_idx.size

    def __getitem__(self, idx):
        image = self._load_raw_image(self._raw_idx[idx])
        if not (isinstance(image, np.ndarray) and image.dtype == np.uint8 and list(image.shape) == self.image_shape):
            raise ValueError(f'image {idx} is not a uint8 array of shape {self.image_shape}')
        if self._xflip[idx]:
            image = image[:, :, ::-1]
        return image.copy(), self.get_label(idx)

    def get_label(self, idx):
        label = self._get_raw_labels()[self._raw_idx[idx]]
        if label.dtype == np.int64:
            hot = np.zeros(self.label_shape, dtype=np.float32)
            hot[label] = 1
            return hot
        return label.copy()

    def get_details(self, idx):
        raw = int(self._raw_idx[idx])
        return dict(raw_idx=raw, xflip=bool(self._xflip[idx]), raw_label=self._get_raw_labels()[raw].copy())

    # -- description --------------------------------------------------------------
    @property
    def name(self):
        return self._name

    @property
    def image_shape(self):
        return list(self._raw_shape[1:])

    @property
    def num_channels(self):
        assert len(self.image_shape) == 3
        return self.image_shape[0]

    @property
    def resolution(self):
        assert len(self.image_shape) == 3 and self.image_shape[1] == self.image_shape[2]
        return self.image_shape[1]

    @property
    def label_shape(self):
        if self._label_shape is None:
            labels = self._get_raw_labels()
            self._label_shape = [int(labels.max()) + 1] if labels.dtype == np.int64 else list(labels.shape[1:])
        return list(self._label_shape)

    @property
    def label_dim(self):
        assert len(self.label_shape) == 1
        return self.label_shape[0]

    @property
    def has_labels(self):
        return any(d != 0 for d in self.label_shape)

    @property
    def has_onehot_labels(self):
        return self._get_raw_labels().dtype == np.int64


class _DirStore:
    """Files below a directory, named by their path relative to it."""
    threads = True

    def __init__(self, path):
        self.path = path

    def names(self):
        return [os.path.relpath(os.path.join(root, f), start=self.path) for root, _, files in os.walk(self.path) for f in files]

    def open(self, name):
        return open(os.path.join(self.path, name), 'rb')

    def close(self):
        pass


class _ZipStore:
    """Members of a zip archive; the handle is opened on first use and never pickled (DataLoader workers reopen it)."""
    threads = False     # one handle, one reader at a time

    def __init__(self, path):
        self.path = path
        self._zip = None

    def _handle(self):
        if self._zip is None:
            self._zip = zipfile.ZipFile(self.path)
        return self._zip

    def names(self):
        return self._handle().namelist()

    def open(self, name):
        return self._handle().open(name, 'r')

    def close(self):
        handle, self._zip = self._zip, None
        if handle is not None:
            handle.close()

    def __getstate__(self):
        return dict(self.__dict__, _zip=None)


def _extension(fname):
    return os.path.splitext(fname)[1].lower()


class ImageFolderDataset(Dataset):
    """Every image file (any format PIL reads) below a directory or inside a zip, in sorted name order; labels from an
    optional `dataset.json` = {"labels": [[file name, class index or vector], ...]} (reference: dataset.py:154-234)."""

    def __init__(self, path, resolution=None, **super_kwargs):
        import PIL.Image
        self._path = path
        if os.path.isdir(path):
            self._store = _DirStore(path)
        elif _extension(path) == '.zip':
            self._store = _ZipStore(path)
        else:
            raise IOError('Path must point to a directory or zip')
        PIL.Image.init()
        self._all_fnames = set(self._store.names())
        self._image_fnames = sorted(f for f in self._all_fnames if _extension(f) in PIL.Image.EXTENSION)
        if not self._image_fnames:
            raise IOError('No image files found in the specified path')
        shape = list(self._load_raw_image(0).shape)
        if resolution is not None and (shape[1] != resolution or shape[2] != resolution):
            raise IOError('Image files do not match the specified resolution')
        super().__init__(name=os.path.splitext(os.path.basename(path))[0], raw_shape=[len(self._image_fnames)] + shape, **super_kwargs)

    def close(self):
        store = getattr(self, '_store', None)
        if store is not None:
            store.close()

    def _load_raw_image(self, raw_idx):
        import PIL.Image
        with self._store.open(self._image_fnames[raw_idx]) as f:
            pixels = np.array(PIL.Image.open(f))
        if pixels.ndim == 2:
            pixels = pixels[:, :, np.newaxis]
        return pixels.transpose(2, 0, 1)       # HWC -> CHW

    def _load_raw_labels(self):
        if 'dataset.json' not in self._all_fnames:
            return None
        with self._store.open('dataset.json') as f:
            listed = json.load(f)['labels']
        if listed is None:
            return None
        by_name = dict(listed)
        labels = np.array([by_name[f.replace('\\', '/')] for f in self._image_fnames])
        return labels.astype({1: np.int64, 2: np.float32}[labels.ndim])


class ResidentImages:
    """A dataset decoded once and kept in device memory as uint8 [raw images in use, C, H, W] (mirrored items share
    the pixels of their originals), with its labels; `batch(items)` returns the float images in [-1, 1] and the
    labels of the given dataset items."""

    def __init__(self, dataset, device, workers=8):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('ResidentImages: the image set lives in GPU memory; got device ' + str(device))
        self.image_shape = list(dataset.image_shape)
        raw_idx = np.asarray(dataset._raw_idx)
        used, slot_of_item = np.unique(raw_idx, return_inverse=True)
        need = int(used.size) * int(np.prod(self.image_shape))
        free, _ = torch.cuda.mem_get_info(self.device)
        if need > 0.8 * free:
            raise RuntimeError(f'ResidentImages: {need / 2**30:.1f} GiB of pixels do not fit the {free / 2**30:.1f} GiB free on {self.device}')
        self._slot = slot_of_item.astype(np.int64)
        self._flip = np.asarray(dataset._xflip, dtype=np.uint8)
        self.images = torch.empty([used.size] + self.image_shape, dtype=torch.uint8, device=self.device)
        # decode in a few host threads, upload in groups through one pinned staging buffer
        group = max(1, min(int(used.size), (64 << 20) // max(1, int(np.prod(self.image_shape)))))
        staging = torch.empty([group] + self.image_shape, dtype=torch.uint8).pin_memory()
        serial = not getattr(getattr(dataset, '_store', None), 'threads', True)
        with ThreadPoolExecutor(max_workers=1 if serial else max(1, min(workers, 16))) as pool:
            for start in range(0, used.size, group):
                ids = used[start:start + group]
                for k, pixels in enumerate(pool.map(dataset._load_raw_image, ids)):
                    if list(pixels.shape) != self.image_shape or pixels.dtype != np.uint8:
                        raise ValueError(f'raw image {ids[k]} is not a uint8 array of shape {self.image_shape}')
                    staging[k].copy_(torch.from_numpy(np.ascontiguousarray(pixels)))
                self.images[start:start + len(ids)].copy_(staging[:len(ids)], non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()   # the staging buffer is reused by the next group
        labels = np.stack([dataset.get_label(i) for i in range(len(dataset))]) if len(dataset) else np.zeros([0, 0], np.float32)
        self.labels = torch.from_numpy(labels.astype(np.float32)).to(self.device)

    def __len__(self):
        return self._slot.size

    def batch(self, items, dtype=torch.float32):
        """items: dataset indices on the host (sequence or array).  -> (images [n, C, H, W] dtype, labels [n, label_dim])."""
        items = np.asarray(items, dtype=np.int64).reshape(-1)
        if items.size and (items.min() < 0 or items.max() >= len(self)):
            raise IndexError(f'item index out of range for a set of {len(self)}')
        table = torch.from_numpy(np.stack([self._slot[items], items])).pin_memory().to(self.device, non_blocking=True)
        flip = torch.from_numpy(self._flip[items]).pin_memory().to(self.device, non_blocking=True)
        images = images_plugin.images_to_float(self.images, table[0], flip, dtype=dtype)
        return images, self.labels.index_select(0, table[1])

    def stream(self, batch, rank=0, num_replicas=1, seed=0, shuffle=True, window_size=0.5, dtype=torch.float32):
        """Endless iterator of (images, labels) batches for one rank, items in the sampler's order."""
        sampler = misc.InfiniteSampler(self, rank=rank, num_replicas=num_replicas, shuffle=shuffle, seed=seed, window_size=window_size)
        while True:
            yield self.batch(sampler.take(batch), dtype=dtype)
