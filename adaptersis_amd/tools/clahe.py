"""Host side of the CLAHE stage of the GPU input pipeline (`train.py:161`: ``A.CLAHE(p=0.8)`` — albumentations
``functional.clahe`` = OpenCV RGB -> Lab, ``cv2.createCLAHE(clipLimit, (8, 8))`` on L, Lab -> RGB, with clipLimit ~ U(1, 4)).

The 8-bit colour conversions of OpenCV (``modules/imgproc/src/color_lab.cpp``: ``RGB2Lab_b`` and the bit-exact integer
inverse ``Lab2RGBinteger``) are integer arithmetic on look-up tables that ``initLabTabs`` builds once; this module builds
the same tables with numpy and hands them to ``csrc/augment.hip`` (``asis_clahe``), which does everything else on the device.
albumentations / OpenCV are not pinned by the reference (`README.md:12`) and absent from this image: parity with the
libraries is unpinned; the kernels are bit-identical to the independent numpy restatement in ``oracle/augment_ref.py``.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

LAB_SHIFT, GAMMA_SHIFT, BASE_SHIFT, INV_GAMMA_SHIFT = 12, 3, 14, 12
MIN_AB = -8145                                   # minABvalue of color_lab.cpp
TILES = 8                                        # albumentations default tile_grid_size=(8, 8)


def _rnd(x):
    return np.rint(x).astype(np.int64)           # cvRound


@functools.lru_cache(maxsize=None)
def lab_tables_np():
    f32 = np.float32
    base = 1 << BASE_SHIFT
    # sRGBGammaTab_b[i] = cvRound(255 * 8 * applyGamma(i / 255))
    x = (np.arange(256).astype(f32) / f32(255)).astype(np.float64)
    gam = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4)).astype(f32)
    gamma = _rnd(f32(255 << GAMMA_SHIFT) * gam)
    # LabCbrtTab_b[i] = cvRound(2^15 * f(i / (255 * 8))), f = cube root with the linear toe below (6/29)^3
    n_cb = 256 * 3 // 2 * (1 << GAMMA_SHIFT)
    xs = (f32(1) / (f32(255) * f32(1 << GAMMA_SHIFT))) * np.arange(n_cb).astype(f32)
    toe = (xs.astype(np.float64) * np.float64(f32(841) / f32(108)) + np.float64(f32(16) / f32(116))).astype(f32)
    cbrt = _rnd(f32(1 << (LAB_SHIFT + GAMMA_SHIFT)) * np.where(xs < f32(216) / f32(24389), toe, np.cbrt(xs).astype(f32)))
    # LabToYF_b[L] = (y, fy) scaled by 2^14
    l2yf = np.zeros((256, 2), np.int64)
    for L in range(256):
        if L <= 20:
            l2yf[L, 0] = int(_rnd(f32(L * base * 20 * 9) / f32(17 * 29 * 29 * 29)))
            l2yf[L, 1] = int(_rnd(f32(base) * (f32(16) / f32(116) + f32(L * 5) / f32(3 * 17 * 29))))
        else:
            fy = f32(L * 100 * base) / f32(255 * 116) + f32(16 * base) / f32(116)
            l2yf[L, 1] = int(_rnd(fy))
            l2yf[L, 0] = int(_rnd(fy * fy * fy / f32(base * base)))
    # abToXZ_b[v - minABvalue]: inverse of f, C integer division (truncating)
    k = np.arange(MIN_AB, base * 9 // 4 + MIN_AB, dtype=np.int64)
    trunc = np.where(k * 108 < 0, -((-k * 108) // 841), (k * 108) // 841)
    ab2xz = np.where(k <= 3390, trunc - base * 16 // 116 * 108 // 841, (k * k) // base * k // base)
    # sRGBInvGammaTab_b[i] = cvRound(255 * applyInvGamma(i / 4096))
    xi = ((f32(1) / f32(1 << INV_GAMMA_SHIFT)) * np.arange(1 << INV_GAMMA_SHIFT).astype(f32)).astype(np.float64)
    inv = np.where(xi <= 0.0031308, xi * 12.92, np.power(xi, 1.0 / 2.4) * 1.055 - 0.055).astype(f32)
    invgamma = _rnd(f32(255) * inv)
    d65 = np.array([0.950456, 1.0, 1.088754])
    fwd = _rnd(np.array([0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227]).reshape(3, 3)
               / d65[:, None] * (1 << LAB_SHIFT))
    bwd = _rnd(np.array([3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311]).reshape(3, 3)
               * d65[None, :] * (1 << LAB_SHIFT))
    return dict(gamma=gamma.astype(np.uint16), cbrt=cbrt.astype(np.uint16), l2yf=l2yf.astype(np.uint16).reshape(-1),
                ab2xz=ab2xz.astype(np.int32), invgamma=invgamma.astype(np.uint8),
                fwd=np.ascontiguousarray(fwd.reshape(-1).astype(np.int32)), inv=np.ascontiguousarray(bwd.reshape(-1).astype(np.int32)))


_DEV = {}


def lab_tables(device) -> dict:
    """The five look-up tables on ``device`` (cached) + the two 3x3 integer matrices as host int32 arrays."""
    key = (device.type, device.index)
    if key not in _DEV:
        t = lab_tables_np()
        _DEV[key] = {k: (torch.from_numpy(v).to(device) if k not in ("fwd", "inv") else v) for k, v in t.items()}
    return _DEV[key]


def tile_edge(S: int, tiles: int = TILES) -> int:
    """tile edge of ``CLAHE_Impl::apply``: the plane is padded (reflect-101, right / bottom) to a multiple of the grid"""
    return S // tiles if S % tiles == 0 else (S + tiles - S % tiles) // tiles


def clip_limit_int(clip: float, S: int, tiles: int = TILES) -> int:
    """``clipLimit = max(static_cast<int>(clipLimit_ * tileSizeTotal / histSize), 1)``"""
    ts = tile_edge(S, tiles)
    return max(int(float(clip) * (ts * ts) / 256), 1)
