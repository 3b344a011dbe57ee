"""Input pipeline: host-side table builder vs the numpy restatement (CPU), and the HIP kernel vs the
restatement bit for bit (GPU)."""
import numpy as np
import pytest
import torch

from adaptersis_amd.tools import augment as A
from adaptersis_amd.tools.dataset import Robomis, collate_u8
from oracle import augment_ref as R


def _data(B, S, seed=0):
    r = np.random.RandomState(seed)
    img = r.randint(0, 256, (B, S, S, 3)).astype(np.uint8)
    # smooth-ish structure as well, so interpolation matters
    img[:, ::2] = (img[:, ::2].astype(np.int32) // 2 + 60).astype(np.uint8)
    mask = (r.random_sample((B, S, S)) > 0.7).astype(np.uint8)
    return img, mask


def test_resize_tables_match_restatement():
    for (s0, n, S) in [(0, 294, 588), (17, 401, 588), (100, 488, 588), (3, 57, 64), (0, 64, 64)]:
        ofs, coef, near = A.resize_tables(s0, n, S, S)
        o2, c0, c1, n2 = R._axis_tables(n, S)
        assert np.array_equal(ofs - s0, o2) and np.array_equal(coef[:, 0], c0) and np.array_equal(coef[:, 1], c1)
        assert np.array_equal(near - s0, n2)
        assert int((coef.astype(np.int32).sum(1) - 2048).__abs__().max()) <= 1


def test_draw_distributions_and_luts():
    aug = A.TrainAugment(size=588, seed=3)
    ps = aug.draw(4000)
    frac = lambda f: sum(1 for p in ps if f(p)) / len(ps)
    assert abs(frac(lambda p: p["crop"] is not None) - 1 / 3) < 0.03          # OneOf([p=0.5, p=1.0]) normalised
    assert abs(frac(lambda p: p["flip"]) - 0.5) < 0.03
    assert abs(frac(lambda p: p["gamma"] is not None) - 0.8) < 0.03
    crops = [p["crop"] for p in ps if p["crop"] is not None]
    assert min(c[3] for c in crops) >= 294 and max(c[3] for c in crops) == 588
    assert all(0 <= c[0] and c[0] + c[2] <= 588 and 0 <= c[1] and c[1] + c[3] <= 588 for c in crops)
    assert all(0.8 <= p["alpha"] <= 1.2 and -0.2 <= p["beta"] <= 0.2 for p in ps)
    assert np.array_equal(A.brightness_contrast_lut(1.0, 0.0), np.arange(256, dtype=np.uint8))
    assert A.gamma_lut(1.0)[255] == 255 and A.gamma_lut(1.2)[128] < 128 < A.gamma_lut(0.8)[128]
    assert abs(frac(lambda p: p["clahe"] is not None) - 0.8) < 0.03            # A.CLAHE(p=0.8), train.py:161
    clips = [p["clahe"] for p in ps if p["clahe"] is not None]
    assert 1.0 <= min(clips) < 1.1 and 3.9 < max(clips) <= 4.0                  # clip_limit=4.0 -> U(1, 4)


def test_clahe_tables_and_restatement():
    """Product-side OpenCV Lab tables (tools/clahe.py) == the independent restatement's (oracle/augment_ref.py); the
    restatement behaves like CLAHE: greys stay grey, a flat tile histogram is the identity up to rounding, contrast grows."""
    from adaptersis_amd.tools import clahe as C
    t, o = C.lab_tables_np(), R.lab_tables()
    assert np.array_equal(t["gamma"], o["gamma"]) and np.array_equal(t["cbrt"], o["cbrt"])
    assert np.array_equal(t["l2yf"].reshape(256, 2), o["l2yf"]) and np.array_equal(t["ab2xz"], o["ab2xz"])
    assert np.array_equal(t["invgamma"], o["invgamma"])
    assert np.array_equal(t["fwd"].reshape(3, 3), o["rgb2xyz"]) and np.array_equal(t["inv"].reshape(3, 3), o["xyz2rgb"])
    assert t["fwd"].reshape(3, 3).sum(1).tolist() == [4096, 4096, 4096]
    assert C.tile_edge(588) == 74 and C.tile_edge(64) == 8 and C.clip_limit_int(2.0, 588) == int(2.0 * 74 * 74 / 256)
    assert C.clip_limit_int(2.0, 588) == R.clahe_clip_limit(2.0, 74 * 74) and C.clip_limit_int(0.001, 64) == 1
    grey = np.stack([np.arange(256, dtype=np.uint8)] * 3, -1)[None].repeat(4, 0)
    lab = R.rgb2lab_u8(grey)
    assert lab[0, 0].tolist() == [0, 128, 128] and lab[0, 255].tolist() == [255, 128, 128] and lab[0, 128, 0] == 137
    assert np.abs(R.lab2rgb_u8(lab).astype(int) - grey.astype(int)).max() <= 1
    r = np.random.RandomState(1)
    low = (r.randint(100, 140, (96, 96, 3))).astype(np.uint8)                     # low-contrast image
    out = R.clahe_rgb_u8(low, 4.0)
    assert out.std() > 1.5 * low.std() and out.shape == low.shape
    assert np.abs(R.clahe_rgb_u8(low, 1.0).astype(int) - low.astype(int)).mean() < np.abs(out.astype(int) - low.astype(int)).mean()


def test_dataset_npy_and_png(tmp_path):
    from PIL import Image
    img, mask = _data(3, 40)
    (tmp_path / "images" / "training").mkdir(parents=True)
    (tmp_path / "annotations" / "training").mkdir(parents=True)
    for i in range(3):
        Image.fromarray(img[i]).save(tmp_path / "images" / "training" / f"f{i}.png")
        Image.fromarray(mask[i] * 255).save(tmp_path / "annotations" / "training" / f"f{i}.png")
    ds = Robomis(str(tmp_path), "training", transform=None, imsize=None)
    assert len(ds) == 3
    im0, m0, idx = ds[1]
    assert im0.dtype == torch.uint8 and tuple(im0.shape) == (40, 40, 3) and torch.equal(im0, torch.from_numpy(img[1]))
    assert torch.equal(m0, torch.from_numpy(mask[1])) and idx == 1
    ds32 = Robomis(str(tmp_path), "training", imsize=32)
    assert tuple(ds32[0][0].shape) == (32, 32, 3) and set(ds32[0][1].unique().tolist()) <= {0, 1}
    b = collate_u8([ds[0], ds[2]])
    assert tuple(b[0].shape) == (2, 40, 40, 3) and b[2].tolist() == [0, 2]
    # albumentations-protocol transform on the host keeps the reference's item format
    tds = Robomis(str(tmp_path), "training", transform=lambda image, mask: {"image": image[:, ::-1], "mask": mask[:, ::-1]})
    x, y, _ = tds[0]
    assert x.dtype == torch.float32 and tuple(x.shape) == (3, 40, 40) and y.dtype == torch.int64
    assert torch.equal(x, torch.from_numpy(img[0][:, ::-1].transpose(2, 0, 1).copy()) / 255.0)
    # decode-free arrays
    (tmp_path / "npy" / "training").mkdir(parents=True)
    np.save(tmp_path / "npy" / "training" / "images.npy", img)
    np.save(tmp_path / "npy" / "training" / "masks.npy", mask)
    nds = Robomis(str(tmp_path / "npy"), "training")
    assert len(nds) == 3 and torch.equal(nds[2][0], torch.from_numpy(img[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("S,B", [(64, 6), (588, 4)])
def test_gpu_augment_bit_exact_vs_restatement(dev, S, B):
    img, mask = _data(B, S, seed=S)
    aug = A.TrainAugment(size=S, seed=11)
    params = aug.draw(B)
    # make sure every branch is exercised in the batch
    params[0].update(crop=None, flip=False, rotk=0, clahe=None, alpha=1.0, beta=0.0, gamma=None)          # identity
    params[1].update(crop=(3, 5, S - 9, S - 9), flip=True, rotk=1, clahe=3.7)
    params[2].update(crop=(0, 0, S // 2, S // 2), flip=False, rotk=3, clahe=1.0, alpha=1.17, beta=-0.11, gamma=0.83)
    params[3].update(crop=(S // 2, S // 2 - 1, S // 2, S // 2), flip=True, rotk=2, clahe=None, gamma=1.19)
    out, mout = aug(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev), params)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and tuple(out.shape) == (B, 3, S, S) and mout.dtype == torch.int64
    for b in range(B):
        ro, rm = R.apply(img[b], mask[b], params[b], S)
        assert np.array_equal(out[b].cpu().numpy(), ro), (b, params[b])
        assert np.array_equal(mout[b].cpu().numpy(), rm), (b, params[b])
    assert torch.equal(out[0].cpu(), torch.from_numpy(img[0].transpose(2, 0, 1).copy()) / 255.0)


@pytest.mark.gpu
def test_gpu_augment_feeds_the_engine(dev):
    """uint8 batch -> TrainAugment -> SegEngine.train_step: shapes / dtypes / value range are what the step expects."""
    from tests.test_gpu_step import build_engine
    eng, _ = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
    img, mask = _data(2, 224, seed=5)
    aug = A.TrainAugment(size=224, seed=2)
    x, y = aug(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev))
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    loss = eng.train_step(x, y)
    assert torch.isfinite(loss).item()


@pytest.mark.gpu
@pytest.mark.parametrize("S", [64, 588, 100])
def test_gpu_clahe_bit_exact_vs_restatement(dev, S):
    """CLAHE alone (no geometry, identity tables) at a size with whole tiles (64), at the reference's 588 (74-pixel tiles of a
    plane padded 588 -> 592 by reflection) and at 100 (13-pixel tiles, 4 padded rows); a batch without any CLAHE sample takes
    the single fused kernel and equals the three-kernel path with every flag off."""
    img, mask = _data(3, S, seed=S + 1)
    img[1] = (img[1] // 4 + 90).astype(np.uint8)                  # a low-contrast sample: the clip limit binds
    aug = A.TrainAugment(size=S, seed=1)
    ident = dict(crop=None, flip=False, rotk=0, alpha=1.0, beta=0.0, gamma=None)
    params = [dict(ident, clahe=1.0), dict(ident, clahe=4.0), dict(ident, clahe=2.3)]
    out, mout = aug(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev), params)
    torch.cuda.synchronize()
    for b in range(3):
        ref = R.clahe_rgb_u8(img[b], params[b]["clahe"]).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        got = out[b].cpu().numpy()
        assert np.array_equal(got, ref), (S, b, float(np.abs(got - ref).max()) * 255)
        assert np.array_equal(mout[b].cpu().numpy(), mask[b].astype(np.int64))
    none = [dict(ident, clahe=None)] * 3
    t = aug.tables(none, dev)
    a, _ = __import__("adaptersis_amd").ops.augment(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev), t)
    t["clahe_any"] = True                                          # force the three-kernel path with every flag off
    b_, _ = __import__("adaptersis_amd").ops.augment(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev), t)
    assert torch.equal(a, b_)
