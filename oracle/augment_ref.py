"""CPU restatement (numpy) of the reference's training-time augmentation — TEST INFRASTRUCTURE ONLY (imported by tests/
and nothing else; the product path is adaptersis_amd/tools/augment.py + csrc/augment.hip).

Follows `train.py:139-163` (the albumentations ``Compose``) and `tools/dataset.py:150-161` (uint8 numpy in, float CHW / 255
out).  albumentations and OpenCV are third-party dependencies that are absent from /root/reference and from this image
(`README.md:12` lists ``albumentations`` without a version), so their published algorithms are restated:

  * ``A.RandomSizedCrop`` = crop [y1:y1+ch, x1:x1+cw] + ``cv2.resize(..., INTER_LINEAR)`` for the image, ``INTER_NEAREST`` for
    the mask (albumentations ``functional.resize`` / ``DualTransform.apply_to_mask``);
  * ``cv2.resize`` INTER_LINEAR on 8-bit data (OpenCV ``imgproc/resize.cpp``): fx = (float)((dx + 0.5) * scale - 0.5),
    sx = floor(fx), border handling by zeroing the fraction, coefficients cvRound(c * 2048) as int16, horizontal pass in int32,
    vertical pass ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``; INTER_NEAREST: sx = min(floor(dx * scale), n - 1);
  * ``A.HorizontalFlip`` = ``img[:, ::-1]``; ``A.RandomRotate90`` = ``np.rot90(img, k)``;
  * ``A.RandomBrightnessContrast`` on uint8 = LUT clip(arange(256) * alpha + beta * 255) (``brightness_by_max=True``);
  * ``A.RandomGamma`` on uint8 = LUT ((arange(256) / 255) ** gamma * 255) truncated.

PARITY UNPINNED against the libraries themselves (they cannot be run here); pinned only as "GPU == this restatement", bit for bit.
"""
import numpy as np


def _axis_tables(n_src, n_dst):
    scale = n_src / n_dst
    ofs = np.empty(n_dst, np.int64)
    c0 = np.empty(n_dst, np.int64)
    c1 = np.empty(n_dst, np.int64)
    near = np.empty(n_dst, np.int64)
    for d in range(n_dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= n_src - 1:
            s, f = n_src - 1, np.float32(0)
        ofs[d] = s
        c0[d] = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
        c1[d] = int(np.rint(np.float32(f * np.float32(2048))))
        near[d] = min(int(np.floor(d * scale)), n_src - 1)
    return ofs, c0, c1, near


def resize_linear_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC."""
    H, W = img.shape[:2]
    xo, xa0, xa1, _ = _axis_tables(W, out_w)
    yo, ya0, ya1, _ = _axis_tables(H, out_h)
    src = img.astype(np.int64)
    x1 = np.minimum(xo + 1, W - 1)
    hor = src[:, xo] * xa0[None, :, None] + src[:, x1] * xa1[None, :, None]           # [H, out_w, C]
    y1 = np.minimum(yo + 1, H - 1)
    t = ((ya0[:, None, None] * (hor[yo] >> 4)) >> 16) + ((ya1[:, None, None] * (hor[y1] >> 4)) >> 16)
    return np.clip((t + 2) >> 2, 0, 255).astype(np.uint8)


def resize_nearest_u8(m, out_h, out_w):
    H, W = m.shape[:2]
    _, _, _, nx = _axis_tables(W, out_w)
    _, _, _, ny = _axis_tables(H, out_h)
    return m[ny][:, nx]


def apply(img, mask, p, size):
    """One sample through the pipeline with the draws ``p`` (a dict of adaptersis_amd.tools.augment.TrainAugment.draw)."""
    if p["crop"] is not None:
        x1, y1, cw, ch = p["crop"]
        img = resize_linear_u8(img[y1:y1 + ch, x1:x1 + cw], size, size)
        mask = resize_nearest_u8(mask[y1:y1 + ch, x1:x1 + cw], size, size)
    if p["flip"]:
        img, mask = img[:, ::-1], mask[:, ::-1]
    if p["rotk"]:
        img, mask = np.rot90(img, p["rotk"]), np.rot90(mask, p["rotk"])
    if p["alpha"] != 1.0 or p["beta"] != 0.0:
        lut = np.arange(0, 256, dtype=np.float32)
        if p["alpha"] != 1:
            lut *= np.float32(p["alpha"])
        if p["beta"] != 0:
            lut += np.float32(p["beta"] * 255.0)
        img = np.clip(lut, 0, 255).astype(np.uint8)[img]
    if p["gamma"] is not None:
        img = (np.power(np.arange(0, 256, dtype=np.float64) / 255.0, p["gamma"]) * 255.0).astype(np.uint8)[img]
    out = np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)     # tools/dataset.py:159
    return out, np.ascontiguousarray(mask).astype(np.int64)
