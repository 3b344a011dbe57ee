"""`tools/dataset.py:127-167` — ``Robomis``: ``<dir>/images/<split>/*.png`` + ``<dir>/annotations/<split>/<same name>``.

Host side exactly as the reference (PIL decode, RGB, mask ``> 0``, resize to ``imsize`` with BILINEAR / NEAREST); what
differs is where the augmentation runs: with ``transform=None`` the item is the raw uint8 pair (HWC image, HW mask) and the
batch goes through ``tools.augment.TrainAugment`` on the GPU (``collate_u8`` + ``TrainAugment.__call__``); a callable
``transform(image=..., mask=...)`` (albumentations protocol) is still honoured on the host and then the item is the
reference's (float CHW / 255, long mask, index).  ``.npy`` arrays (``images.npy`` uint8 [N,H,W,3] or float [N,3,H,W],
``masks.npy``) are accepted in place of PNG folders: decode-free.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch


class Robomis(torch.utils.data.Dataset):
    def __init__(self, dir_main, split, transform=None, imsize=None):
        super().__init__()
        self.transform, self.imsize = transform, imsize
        self.arrays = None
        npy = os.path.join(dir_main, split, "images.npy")
        if os.path.isfile(npy):
            self.arrays = (np.load(npy, mmap_mode="r"), np.load(os.path.join(dir_main, split, "masks.npy"), mmap_mode="r"))
            self.img_files = list(range(self.arrays[0].shape[0]))
            self.mask_files = self.img_files
        else:
            self.img_files = sorted(glob.glob(os.path.join(dir_main, "images", split, "*.png")))
            self.mask_files = [os.path.join(dir_main, "annotations", split, os.path.basename(p)) for p in self.img_files]

    def _load(self, index):
        if self.arrays is not None:
            img, mask = np.array(self.arrays[0][index]), np.array(self.arrays[1][index])   # copies out of the memory map
            if img.dtype != np.uint8:                       # float [3,H,W] in [0,1] -> uint8 HWC
                img = np.clip(np.rint(np.moveaxis(img, 0, -1) * 255.0), 0, 255).astype(np.uint8)
            mask = (mask > 0).astype(np.uint8)
            if self.imsize is not None and img.shape[:2] != (self.imsize, self.imsize):
                # same resize as the PNG path (`tools/dataset.py:147-149`): PIL BILINEAR for the image, NEAREST for the mask
                from PIL import Image
                img = np.array(Image.fromarray(img).resize((self.imsize, self.imsize), resample=Image.BILINEAR)).astype(np.uint8)
                mask = np.array(Image.fromarray(mask * 255).resize((self.imsize, self.imsize), resample=Image.NEAREST))
                mask = (mask > 0).astype(np.uint8)
            return img, mask
        from PIL import Image
        with open(self.img_files[index], "rb") as f:
            img = Image.open(f).convert("RGB")
        with open(self.mask_files[index], "rb") as f:
            mask = Image.open(f)
            mask = mask.point(lambda x: 1 if x > 0 else 0, mode="1")
        if self.imsize is not None:
            img = img.resize((self.imsize, self.imsize), resample=Image.BILINEAR)
            mask = mask.resize((self.imsize, self.imsize), resample=Image.NEAREST)
        return np.array(img).astype(np.uint8), np.array(mask).astype(np.uint8)

    def __getitem__(self, index):
        img_np, mask_np = self._load(index)
        if self.transform is not None:
            t = self.transform(image=img_np, mask=mask_np)
            return torch.from_numpy(t["image"].transpose(2, 0, 1).copy()) / 255.0, torch.from_numpy(t["mask"].copy()).long(), index
        return torch.from_numpy(img_np), torch.from_numpy(mask_np), index

    def __len__(self):
        return len(self.img_files)


def collate_u8(items):
    """DataLoader ``collate_fn`` for the GPU-augmented path: -> (uint8 [B,H,W,3], uint8 [B,H,W], int64 [B])."""
    return (torch.stack([i[0] for i in items]), torch.stack([i[1] for i in items]),
            torch.tensor([i[2] for i in items], dtype=torch.int64))
