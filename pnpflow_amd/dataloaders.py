"""Test-set readers with the API of pnpflow/dataloaders.py (reference :17-219):

    loaders = DataLoaders(dataset_name, batch_size_train, batch_size_test).load_data()
    for clean_img, labels in loaders["test"]: ...        # clean_img in [-1, 1], (B, 3, H, W) fp32

The reference builds torchvision transforms + torch DataLoaders; torchvision is not part of this engine's
environment, so the same pixel pipeline is restated on PIL + numpy (torchvision itself calls PIL for PIL inputs):
    celeba    CenterCrop(178) -> Resize((128,128)) bilinear -> ToTensor -> Normalize(0.5, 0.5)   (:22-27)
    afhq_cat  Resize((256,256)) bilinear -> ToTensor -> Normalize(0.5, 0.5)                       (:80-84)
    celebahq  Resize(256) (shorter edge) -> ToTensor -> 2x-1                                      (:61-64, :178-180)
File lists, the partition CSV parsing (including its pandas call, which drops the first listed image), sorted
directory order, `shuffle=False` for val/test and the None-filtering collate (:209-216) are kept.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch


def _open_rgb(path):
    from PIL import Image
    return Image.open(path).convert('RGB')


def center_crop(img, size):
    """torchvision.transforms.functional.center_crop on a PIL image (zero padding when the image is smaller)."""
    from PIL import Image
    w, h = img.size
    if w < size or h < size:
        pl, pt = max((size - w) // 2, 0), max((size - h) // 2, 0)
        canvas = Image.new(img.mode, (max(w, size), max(h, size)))
        canvas.paste(img, (pl, pt))
        img = canvas
        w, h = img.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def resize(img, size):
    """torchvision Resize on a PIL image: (h, w) tuple -> exact size; int -> shorter edge, aspect kept. Bilinear (PIL antialiases)."""
    from PIL import Image
    if isinstance(size, int):
        w, h = img.size
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long_ / short)
        ow, oh = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        oh, ow = size
    return img.resize((ow, oh), Image.BILINEAR)


def to_tensor(img) -> torch.Tensor:
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)


def normalize_half(x: torch.Tensor) -> torch.Tensor:
    return (x - 0.5) / 0.5


class _FolderDataset:
    def __init__(self, img_dir, names, transform, post=None):
        self.img_dir, self.names, self.transform, self.post = img_dir, list(names), transform, post

    def __len__(self):
        return len(self.names)

    def __getitem__(self, idx):
        path = os.path.join(self.img_dir, self.names[idx])
        if not os.path.exists(path):
            warnings.warn(f"File not found: {path}. Skipping.")
            return None, None
        x = self.transform(_open_rgb(path))
        if self.post is not None:
            x = self.post(x)
        return x.float(), 0


class CelebADataset(_FolderDataset):
    """reference :121-153"""

    def __init__(self, img_dir, partition_csv, partition, transform=None):
        import pandas as pd
        df = pd.read_csv(partition_csv, header=0, names=['image', 'partition'], skiprows=1)      # the reference's exact call
        super().__init__(img_dir, df[df['partition'] == partition]['image'].values, transform)
        self.partition = partition


class AFHQDataset(_FolderDataset):
    """reference :185-206"""

    def __init__(self, img_dir, batchsize, category='cat', transform=None):
        super().__init__(img_dir, sorted(os.listdir(img_dir)), transform)
        self.batchsize = batchsize


class CelebAHQDataset(_FolderDataset):
    """reference :156-182 (images mapped to [-1, 1] by 2x-1 after the transform)"""

    def __init__(self, data_dir, batchsize, transform=None):
        super().__init__(data_dir, sorted(os.listdir(data_dir)), transform, post=lambda x: 2 * x - 1)
        self.batchsize = batchsize


def custom_collate(batch):
    """reference :209-216: missing files are dropped from the batch; an empty batch is a pair of empty tensors."""
    batch = [b for b in batch if b[0] is not None]
    if len(batch) == 0:
        return torch.tensor([]), torch.tensor([])
    return torch.stack([b[0] for b in batch]), torch.tensor([b[1] for b in batch])


class DataLoader:
    """Sequential batch iterator with the torch.utils.data.DataLoader surface the solvers use (iter / len)."""

    def __init__(self, dataset, batch_size, shuffle=False, collate_fn=custom_collate, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.collate_fn, self.drop_last = dataset, int(batch_size), shuffle, collate_fn, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                return
            yield self.collate_fn([self.dataset[j] for j in idx])


class DataLoaders:
    def __init__(self, dataset_name, batch_size_train, batch_size_test, root='./'):
        self.dataset_name = dataset_name
        self.batch_size_train = batch_size_train
        self.batch_size_test = batch_size_test
        self.root = root

    def paths(self):
        """The folders / files `load_data` reads (reference :29-31, :66, :88-90)."""
        d = os.path.join(self.root, 'data')
        if self.dataset_name == 'celeba':
            return [os.path.join(d, 'celeba/img_align_celeba/'), os.path.join(d, 'celeba/list_eval_partition.csv')]
        if self.dataset_name == 'celebahq':
            return [os.path.join(d, 'celebahq/test/')]
        if self.dataset_name == 'afhq_cat':
            return [os.path.join(d, 'afhq_cat/test/cat/'), os.path.join(d, 'afhq_cat/val/cat/'), os.path.join(d, 'afhq_cat/train/cat/')]
        raise ValueError("The dataset your entered does not exist")

    def available(self, split='test'):
        p = self.paths()
        need = p if self.dataset_name != 'afhq_cat' else [p[{'test': 0, 'val': 1, 'train': 2}[split]]]
        return all(os.path.exists(q) for q in need)

    def load_data(self):
        if self.dataset_name == 'celeba':
            tf = lambda im: normalize_half(to_tensor(resize(center_crop(im, 178), (128, 128))))
            img_dir, csv = self.paths()
            mk = lambda part, bs, sh: DataLoader(CelebADataset(img_dir, csv, partition=part, transform=tf), bs, shuffle=sh)
            train_loader, val_loader, test_loader = mk(0, self.batch_size_train, True), mk(1, self.batch_size_test, False), mk(2, self.batch_size_test, False)
        elif self.dataset_name == 'celebahq':
            tf = lambda im: to_tensor(resize(im, 256))
            train_loader = val_loader = None
            test_loader = DataLoader(CelebAHQDataset(self.paths()[0], self.batch_size_test, transform=tf), self.batch_size_test)
        elif self.dataset_name == 'afhq_cat':
            tf = lambda im: normalize_half(to_tensor(resize(im, (256, 256))))
            t, v, tr = self.paths()
            mk = lambda dr, bs, sh, dl=False: DataLoader(AFHQDataset(dr, bs, transform=tf), bs, shuffle=sh, drop_last=dl) if os.path.isdir(dr) else None
            test_loader, val_loader, train_loader = mk(t, self.batch_size_test, False), mk(v, self.batch_size_test, False), mk(tr, self.batch_size_train, True, True)
        else:
            raise ValueError("The dataset your entered does not exist")
        return {'train': train_loader, 'test': test_loader, 'val': val_loader}
