"""Lab: how much of the decode-head part of the reference_exact step hides under the NEXT batch's frozen trunk when the two
run on different streams (cross-step pipelining)?  python scripts/pipeline_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from adaptersis_amd.utils import weights as W


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    eng = bench.build_engine("vit_large", dev, lr=1e-3)
    img, tgt = W.synthetic_batch(12, 588)
    img, tgt = img.to(dev), tgt.to(dev)
    for _ in range(3):
        eng.train_step(img, tgt)
    torch.cuda.synchronize()

    def timeit(f, n=8):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_full = timeit(lambda: eng.train_step(img, tgt))
    feats = eng.features
    t_trunk = timeit(lambda: feats(img, None, None))
    cat = feats(img, None, None)
    torch.cuda.synchronize()
    eng.features = lambda *a, **k: cat            # decoder-only step
    t_dec = timeit(lambda: eng.train_step(img, tgt))
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def both():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            feats(img, None, None)                 # next batch's trunk
        eng.train_step(img, tgt)                   # this batch's decoder (features patched to the cached tensor)
        main_s.wait_stream(side)
    t_both = timeit(both)
    print(f"full step {t_full:.2f} ms | trunk alone {t_trunk:.2f} | decode head + backward + SGD alone {t_dec:.2f} | "
          f"sum {t_trunk + t_dec:.2f} | both concurrently {t_both:.2f} ms -> {12 / t_both * 1e3:.1f} img/s")


if __name__ == "__main__":
    main()
