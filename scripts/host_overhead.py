"""How long does the host take to ENQUEUE one training step (Python + ctypes + HIP launches) against the GPU's 77 ms?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
eng = bench.build_engine("vit_large", dev, lr=0.01)
img, tgt = bench.synthetic(12, 588, 0, dev)
for _ in range(3): eng.train_step(img, tgt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.train_step(img, tgt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/5:.1f} ms/step; until GPU done {1e3*(t2-t0)/5:.1f} ms/step")
