"""Training-time augmentation on the GPU — the albumentations pipeline of `train.py:139-163`, which the reference runs on
uint8 numpy images inside its DataLoader workers (`tools/dataset.py:150-161`), as ONE HIP kernel per batch
(`csrc/augment.hip`) behind a host-side parameter sampler.  At ~70 ms per 12-image step a CPU pipeline (PIL decode +
albumentations ≈ 15-25 ms per 588x588 image per worker) is the next bottleneck.

What the reference composes (albumentations, version unpinned in `README.md:12`; semantics restated from its 1.x sources):

    OneOf([RandomSizedCrop(min_max_height=(294, 588), height=588, width=588, p=0.5),
           PadIfNeeded(588, 588, border_mode=BORDER_CONSTANT)], p=1)        -> crop with prob 0.5 / (0.5 + 1.0) = 1/3, else no-op
    HorizontalFlip(p=0.5); RandomRotate90(p=0.5)
    OneOf([ElasticTransform, GridDistortion, OpticalDistortion], p=0)       -> never applied
    CLAHE(p=0.8); RandomBrightnessContrast(p=0.8); RandomGamma(p=0.8)

Built here: crop + cv2.resize back to S x S (INTER_LINEAR, 8-bit fixed-point form; mask INTER_NEAREST), flip, rot90,
CLAHE (OpenCV's 8-bit RGB <-> Lab integer paths + tiled contrast-limited equalisation of L on the 8 x 8 grid, clip limit
~ U(1, 4): ``tools/clahe.py`` + the ``asis_clahe`` kernels), brightness/contrast and gamma as uint8 look-up tables (exactly how
albumentations applies them to uint8 images), float / 255.  A batch with CLAHE samples runs three kernels (the tile histograms
need the whole geometrically transformed image), a batch without any runs the single fused one.
The random draws come from this module's own ``numpy`` generator — the same distributions, not albumentations' stream.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import ops
from . import clahe as _clahe

COEF_BITS = 11   # OpenCV INTER_RESIZE_COEF_BITS


def _round_half_even(x: np.ndarray) -> np.ndarray:
    return np.rint(x)          # cvRound: round to nearest, ties to even


def resize_tables(src0: int, src_len: int, dst_len: int, full: int):
    """OpenCV ``resize`` (INTER_LINEAR, 8U) coordinate tables for one axis of a crop [src0, src0+src_len) of an axis of
    ``full`` pixels resized to ``dst_len``: (ofs int32 [dst], coef int16 [dst, 2], nearest int32 [dst])."""
    scale = float(src_len) / float(dst_len)                        # double
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)               # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src_len - 1
    f[hi] = 0.0
    s[hi] = src_len - 1
    one = np.float32(1.0)
    scale_c = np.float32(1 << COEF_BITS)
    a0 = np.clip(_round_half_even((one - f) * scale_c), -32768, 32767).astype(np.int16)
    a1 = np.clip(_round_half_even(f * scale_c), -32768, 32767).astype(np.int16)
    near = np.minimum(np.floor(d * scale).astype(np.int32), src_len - 1)   # INTER_NEAREST: sx = min(floor(dx*scale), ssize-1)
    return (s + src0).astype(np.int32), np.stack([a0, a1], -1), (near + src0).astype(np.int32)


def brightness_contrast_lut(alpha: float, beta: float) -> np.ndarray:
    """albumentations ``_brightness_contrast_adjust_uint`` (brightness_by_max=True): lut = clip(arange*alpha + beta*255)."""
    lut = np.arange(0, 256, dtype=np.float32)
    if alpha != 1:
        lut *= np.float32(alpha)
    if beta != 0:
        lut += np.float32(beta * 255.0)
    return np.clip(lut, 0, 255).astype(np.uint8)


def gamma_lut(gamma: float) -> np.ndarray:
    """albumentations ``gamma_transform`` on uint8: table = (arange/255)**gamma * 255 (truncated), applied with cv2.LUT."""
    return (np.power(np.arange(0, 256, dtype=np.float64) / 255.0, gamma) * 255.0).astype(np.uint8)


class TrainAugment:
    def __init__(self, size: int = 588, seed: int = 0, crop_p: float = 0.5, pad_p: float = 1.0, flip_p: float = 0.5,
                 rot_p: float = 0.5, clahe_p: float = 0.8, bc_p: float = 0.8, gamma_p: float = 0.8,
                 min_crop: Optional[int] = None, clahe_clip=(1.0, 4.0)):
        if size < 16:
            raise ValueError("TrainAugment: size must be at least 16 (8 x 8 CLAHE tile grid)")
        self.size = size
        self.clahe_p, self.clahe_clip = clahe_p, clahe_clip        # A.CLAHE(clip_limit=4.0) -> clip ~ U(1, 4), tiles (8, 8)
        self.min_crop = int(size * 0.5) if min_crop is None else min_crop      # min_max_height=(int(588*0.5), 588)
        self.p_crop = crop_p / (crop_p + pad_p)                                # OneOf normalises its members' p
        self.flip_p, self.rot_p, self.bc_p, self.gamma_p = flip_p, rot_p, bc_p, gamma_p
        self.rng = np.random.RandomState(seed)

    # ---- parameter draws (host) -----------------------------------------------------------------------------------
    def draw(self, n: int):
        """-> list of per-sample dicts (crop box or None, flip, rotk, alpha, beta, gamma)."""
        S, r = self.size, self.rng
        out = []
        for _ in range(n):
            p: Dict = {"crop": None, "flip": False, "rotk": 0, "clahe": None, "alpha": 1.0, "beta": 0.0, "gamma": None}
            if r.random_sample() < self.p_crop:
                ch = int(r.randint(self.min_crop, S + 1))          # random.randint(min, max), both inclusive
                cw = ch                                            # w2h_ratio = 1.0
                hs, ws = r.random_sample(), r.random_sample()
                y1 = int((S - ch + 1) * hs)                        # get_random_crop_coords
                x1 = int((S - cw + 1) * ws)
                p["crop"] = (x1, y1, cw, ch)
            p["flip"] = bool(r.random_sample() < self.flip_p)
            if r.random_sample() < self.rot_p:
                p["rotk"] = int(r.randint(0, 4))
            if r.random_sample() < self.clahe_p:
                p["clahe"] = float(r.uniform(*self.clahe_clip))
            if r.random_sample() < self.bc_p:
                p["alpha"] = 1.0 + r.uniform(-0.2, 0.2)
                p["beta"] = 0.0 + r.uniform(-0.2, 0.2)
            if r.random_sample() < self.gamma_p:
                p["gamma"] = r.uniform(80, 120) / 100.0
            out.append(p)
        return out

    def tables(self, params, device) -> Dict[str, torch.Tensor]:
        S = self.size
        B = len(params)
        geo = np.zeros((B, 4), np.int32)
        xofs = np.zeros((B, S), np.int32); yofs = np.zeros((B, S), np.int32)
        xa = np.zeros((B, S, 2), np.int16); ya = np.zeros((B, S, 2), np.int16)
        mx = np.zeros((B, S), np.int32); my = np.zeros((B, S), np.int32)
        lut = np.zeros((B, 256), np.uint8)
        clahe = np.zeros((B, 2), np.int32)
        for b, p in enumerate(params):
            if p.get("clahe") is not None:
                clahe[b] = (1, _clahe.clip_limit_int(p["clahe"], S))
            geo[b] = (int(p["flip"]), p["rotk"], int(p["crop"] is None), 0)
            if p["crop"] is not None:
                x1, y1, cw, ch = p["crop"]
                xofs[b], xa[b], mx[b] = resize_tables(x1, cw, S, S)
                yofs[b], ya[b], my[b] = resize_tables(y1, ch, S, S)
            l = np.arange(256, dtype=np.uint8)
            if p["alpha"] != 1.0 or p["beta"] != 0.0:
                l = brightness_contrast_lut(p["alpha"], p["beta"])[l]
            if p["gamma"] is not None:
                l = gamma_lut(p["gamma"])[l]
            lut[b] = l
        t = dict(geo=geo, xofs=xofs, yofs=yofs, xa=xa, ya=ya, mx=mx, my=my, lut=lut, clahe=clahe)
        out = {k: torch.from_numpy(v).to(device, non_blocking=True) for k, v in t.items()}
        out["clahe_any"] = bool(clahe[:, 0].any())      # host flag: no CLAHE sample -> the single fused kernel
        return out

    def __call__(self, img_u8: torch.Tensor, mask_u8: torch.Tensor, params=None):
        """uint8 [B,S,S,3] + uint8 [B,S,S] on the device -> (fp32 [B,3,S,S] in [0,1], int64 [B,S,S])."""
        if params is None:
            params = self.draw(img_u8.shape[0])
        return ops.augment(img_u8, mask_u8, self.tables(params, img_u8.device))
