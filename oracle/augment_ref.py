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
    if p.get("clahe") is not None:
        img = clahe_rgb_u8(np.ascontiguousarray(img), p["clahe"])
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


# ======================================================================================================================
# CLAHE (`train.py:161` ``A.CLAHE(p=0.8)``): albumentations ``functional.clahe`` on an RGB uint8 image =
#     lab = cv2.cvtColor(img, cv2.COLOR_RGB2LAB); lab[..., 0] = cv2.createCLAHE(clipLimit, (8, 8)).apply(lab[..., 0]);
#     img = cv2.cvtColor(lab, cv2.COLOR_LAB2RGB)                       with clipLimit ~ U(1, 4) per call (clip_limit=4.0 -> (1, 4))
# Restated from the OpenCV 4.x sources as published (no OpenCV in this image: PARITY UNPINNED against the library; where the
# restatement could not be checked against a build it says so):
#   * modules/imgproc/src/color_lab.cpp, 8-bit paths: ``RGB2Lab_b`` (integer: gamma table, 12-bit matrix, cube-root table,
#     lab_shift = 12, gamma_shift = 3) and ``Lab2RGBinteger`` (the bit-exact integer inverse that ``Lab2RGB_b`` uses for the
#     default sRGB / D65 case: LabToYF_b, abToXZ_b, sRGBInvGammaTab_b, base_shift = 14, inv_gamma_shift = 12);
#     the tables are built by ``initLabTabs`` in IEEE binary32 / binary64 soft-float — restated with numpy float32 / float64
#     (pow / cbrt from libm instead of OpenCV's own soft-float routines: a last-place difference could move a table entry that
#     sits within 1e-7 of a rounding boundary; not checkable here);
#   * modules/imgproc/src/clahe.cpp: ``CLAHE_Impl::apply`` (reflect-101 padding to a multiple of the tile grid, integer clip
#     limit), ``CLAHE_CalcLut_Body`` (histogram, clip, redistribute, cumulative LUT through a float scale),
#     ``CLAHE_Interpolation_Body`` (bilinear blend of the four neighbouring tile LUTs in float, cvRound).
# ======================================================================================================================
_F32 = np.float32
LAB_SHIFT, GAMMA_SHIFT = 12, 3
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
LAB_CBRT_TAB_SIZE_B = 256 * 3 // 2 * (1 << GAMMA_SHIFT)        # 3072
BASE_SHIFT, INV_GAMMA_SHIFT = 14, 12
LAB_BASE = 1 << BASE_SHIFT
INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT
MIN_AB = -8145
_SRGB2XYZ = (0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227)
_XYZ2SRGB = (3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311)
_D65 = (0.950456, 1.0, 1.088754)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _cvround(x):
    return np.rint(x).astype(np.int64)        # lrint: nearest, ties to even


def lab_tables():
    """``initLabTabs`` (8-bit tables) + the integer matrices of ``RGB2Lab_b`` / ``Lab2RGBinteger`` (sRGB, D65, RGB order)."""
    t = {}
    i = np.arange(256)
    x = (i.astype(_F32) / _F32(255)).astype(np.float64)                                   # softfloat(i) / f255, widened
    g = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))              # applyGamma in softdouble
    t["gamma"] = _cvround(_F32(255 * (1 << GAMMA_SHIFT)) * g.astype(_F32)).astype(np.int64)
    j = np.arange(LAB_CBRT_TAB_SIZE_B)
    xs = (_F32(1) / (_F32(255) * _F32(1 << GAMMA_SHIFT))) * j.astype(_F32)                 # cbTabScale * i
    lthresh, lscale, lbias = _F32(216) / _F32(24389), _F32(841) / _F32(108), _F32(16) / _F32(116)
    lin = (xs.astype(np.float64) * np.float64(lscale) + np.float64(lbias)).astype(_F32)    # mulAdd: one rounding
    t["cbrt"] = _cvround(_F32(1 << LAB_SHIFT2) * np.where(xs < lthresh, lin, np.cbrt(xs).astype(_F32))).astype(np.int64)
    c = np.array(_SRGB2XYZ, np.float64).reshape(3, 3) / np.array(_D65, np.float64)[:, None]
    t["rgb2xyz"] = _cvround(c * (1 << LAB_SHIFT))                                           # rows X, Y, Z over (R, G, B)
    # L -> (y, fy), both scaled by LAB_BASE
    yf = np.zeros((256, 2), np.int64)
    for L in range(256):
        if L <= 20:
            yf[L, 0] = int(_cvround(_F32(L * LAB_BASE * 20 * 9) / _F32(17 * 29 * 29 * 29)))
            yf[L, 1] = int(_cvround(_F32(LAB_BASE) * (_F32(16) / _F32(116) + _F32(L * 5) / _F32(3 * 17 * 29))))
        else:
            fy = _F32(L * 100 * LAB_BASE) / _F32(255 * 116) + _F32(16 * LAB_BASE) / _F32(116)
            yf[L, 1] = int(_cvround(fy))
            yf[L, 0] = int(_cvround(fy * fy * fy / _F32(LAB_BASE * LAB_BASE)))
    t["l2yf"] = yf
    k = np.arange(MIN_AB, LAB_BASE * 9 // 4 + MIN_AB, dtype=np.int64)
    # C++ '/' truncates toward zero: written out for k < 0
    lo = np.where(k * 108 < 0, -((-k * 108) // 841), (k * 108) // 841) - LAB_BASE * 16 // 116 * 108 // 841
    hi = np.where(k < 0, -(((k * k) // LAB_BASE * -k) // LAB_BASE), ((k * k) // LAB_BASE * k) // LAB_BASE)
    t["ab2xz"] = np.where(k <= 3390, lo, hi)
    n = np.arange(INV_GAMMA_TAB_SIZE)
    xi = ((_F32(1) / _F32(INV_GAMMA_TAB_SIZE)) * n.astype(_F32)).astype(np.float64)
    ig = np.where(xi <= 0.0031308, xi * 12.92, np.power(xi, 1.0 / 2.4) * 1.055 - 0.055)    # applyInvGamma in softdouble
    t["invgamma"] = _cvround(_F32(255) * ig.astype(_F32)).astype(np.int64)
    m = np.array(_XYZ2SRGB, np.float64).reshape(3, 3) * np.array(_D65, np.float64)[None, :]
    t["xyz2rgb"] = _cvround(m * (1 << LAB_SHIFT))                                           # rows R, G, B over (X, Y, Z)
    return t


_LT = None


def _lt():
    global _LT
    if _LT is None:
        _LT = lab_tables()
    return _LT


def rgb2lab_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_RGB2LAB) for uint8 HxWx3 (``RGB2Lab_b``)."""
    t = _lt()
    R, G, B = (t["gamma"][img[..., c]] for c in range(3))
    C = t["rgb2xyz"]
    f = [t["cbrt"][_descale(R * C[r, 0] + G * C[r, 1] + B * C[r, 2], LAB_SHIFT)] for r in range(3)]
    Lscale, Lshift = (116 * 255 + 50) // 100, -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * f[1] + Lshift, LAB_SHIFT2)
    a = _descale(500 * (f[0] - f[1]) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    b = _descale(200 * (f[1] - f[2]) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], -1), 0, 255).astype(np.uint8)


def lab2rgb_u8(lab):
    """cv2.cvtColor(lab, cv2.COLOR_LAB2RGB) for uint8 HxWx3 (``Lab2RGBinteger::process``)."""
    t = _lt()
    LL, aa, bb = (lab[..., c].astype(np.int64) for c in range(3))
    y, ify = t["l2yf"][LL, 0], t["l2yf"][LL, 1]
    adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE // 500
    bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE // 200 + 1
    x = t["ab2xz"][ify + adiv - MIN_AB]
    z = t["ab2xz"][ify - bdiv - MIN_AB]
    C = t["xyz2rgb"]
    shift = LAB_SHIFT + (BASE_SHIFT - INV_GAMMA_SHIFT)
    out = []
    for r in range(3):
        v = _descale(C[r, 0] * x + C[r, 1] * y + C[r, 2] * z, shift)
        out.append(t["invgamma"][np.clip(v, 0, INV_GAMMA_TAB_SIZE - 1)])
    return np.stack(out, -1).astype(np.uint8)


def clahe_clip_limit(clip: float, tile_area: int) -> int:
    """``CLAHE_Impl::apply``: clipLimit = max(static_cast<int>(clipLimit_ * tileSizeTotal / histSize), 1)."""
    return max(int(float(clip) * tile_area / 256), 1)


def clahe_u8(L, clip: float, tiles: int = 8):
    """cv2.createCLAHE(clipLimit=clip, tileGridSize=(tiles, tiles)).apply(L) for a uint8 HxW plane."""
    H, W = L.shape
    ph, pw = (tiles - H % tiles) if H % tiles else 0, (tiles - W % tiles) if W % tiles else 0
    if ph or pw:      # copyMakeBorder(..., 0, tiles - H % tiles, 0, tiles - W % tiles, BORDER_REFLECT_101): both sides padded
        ph, pw = tiles - H % tiles, tiles - W % tiles
        ext = np.pad(L, ((0, ph), (0, pw)), mode="reflect")
    else:
        ext = L
    th, tw = ext.shape[0] // tiles, ext.shape[1] // tiles
    area = th * tw
    lut_scale = _F32(255) / _F32(area)
    limit = clahe_clip_limit(clip, area)
    luts = np.zeros((tiles, tiles, 256), np.uint8)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            over = hist > limit
            clipped = int((hist[over] - limit).sum())
            hist[over] = limit
            batch = clipped // 256
            residual = clipped - batch * 256
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                hist[idx] += 1
            cum = np.cumsum(hist)
            luts[ty, tx] = np.clip(_cvround(cum.astype(_F32) * lut_scale), 0, 255).astype(np.uint8)
    inv_th, inv_tw = _F32(1) / _F32(th), _F32(1) / _F32(tw)
    yy, xx = np.arange(H), np.arange(W)
    tyf = yy.astype(_F32) * inv_th - _F32(0.5)
    txf = xx.astype(_F32) * inv_tw - _F32(0.5)
    ty1, tx1 = np.floor(tyf).astype(np.int64), np.floor(txf).astype(np.int64)
    ya, xa = (tyf - ty1.astype(_F32)).astype(_F32), (txf - tx1.astype(_F32)).astype(_F32)
    ya1, xa1 = _F32(1) - ya, _F32(1) - xa
    ty2, tx2 = np.minimum(ty1 + 1, tiles - 1), np.minimum(tx1 + 1, tiles - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = L.astype(np.int64)
    l11 = luts[ty1[:, None], tx1[None, :], v].astype(_F32)
    l12 = luts[ty1[:, None], tx2[None, :], v].astype(_F32)
    l21 = luts[ty2[:, None], tx1[None, :], v].astype(_F32)
    l22 = luts[ty2[:, None], tx2[None, :], v].astype(_F32)
    top = (l11 * xa1[None, :]).astype(_F32) + (l12 * xa[None, :]).astype(_F32)
    bot = (l21 * xa1[None, :]).astype(_F32) + (l22 * xa[None, :]).astype(_F32)
    res = (top * ya1[:, None]).astype(_F32) + (bot * ya[:, None]).astype(_F32)
    return np.clip(_cvround(res), 0, 255).astype(np.uint8)


def clahe_rgb_u8(img, clip: float, tiles: int = 8):
    """albumentations ``functional.clahe`` on an RGB uint8 image."""
    lab = rgb2lab_u8(img)
    lab[..., 0] = clahe_u8(lab[..., 0], clip, tiles)
    return lab2rgb_u8(lab)
