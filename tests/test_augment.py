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
    with pytest.raises(ValueError):
        A.TrainAugment(clahe_p=0.8)


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
    params[0].update(crop=None, flip=False, rotk=0, alpha=1.0, beta=0.0, gamma=None)          # identity
    params[1].update(crop=(3, 5, S - 9, S - 9), flip=True, rotk=1)
    params[2].update(crop=(0, 0, S // 2, S // 2), flip=False, rotk=3, alpha=1.17, beta=-0.11, gamma=0.83)
    params[3].update(crop=(S // 2, S // 2 - 1, S // 2, S // 2), flip=True, rotk=2, gamma=1.19)
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
