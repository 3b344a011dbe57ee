"""The handful of helpers from the reference's `utils.py` that the training scripts call (SURVEY.md §2a:
`init_distributed_mode` :467-499, `MetricLogger`/`SmoothedValue` :224-400, `restart_from_checkpoint` :152-184,
`bool_flag` :187-199, `is_main_process` :432-433).  Same behaviour, no DINO-v1 baggage."""
from __future__ import annotations

import argparse
import datetime
import os
import sys
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


def bool_flag(s):
    FALSY, TRUTHY = {"off", "false", "0"}, {"on", "true", "1"}
    if s.lower() in FALSY:
        return False
    if s.lower() in TRUTHY:
        return True
    raise argparse.ArgumentTypeError("invalid value for a boolean flag")


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """`utils.py:467-499`: RANK/WORLD_SIZE/LOCAL_RANK env (torchrun) or single GPU; exits without a GPU."""
    if not torch.cuda.is_available():
        print("Does not support training without GPU.")
        sys.exit(1)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        print("Will run the code on one GPU.")
        args.rank, args.gpu, args.world_size = 0, 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(args.gpu)
    if args.world_size > 1:
        dist.init_process_group(backend="nccl", init_method=getattr(args, "dist_url", "env://"),
                                world_size=args.world_size, rank=args.rank, device_id=torch.device("cuda", args.gpu))
        dist.barrier()
    if args.rank != 0:  # mute non-main ranks like utils.py:452-464
        import builtins
        bp = builtins.print
        builtins.print = lambda *a, force=False, **k: bp(*a, **k) if force else None


class SmoothedValue:
    """`utils.py:224-285`."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.6f} ({global_avg:.6f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, value=self.value)


class MetricLogger:
    """`utils.py:308-400` (log_every prints iter/data time and max memory)."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            self.meters[k].update(v.item() if isinstance(v, torch.Tensor) else float(v))

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=""):
        start = end = time.time()
        it = SmoothedValue(fmt="{avg:.6f}")
        n = len(iterable)
        for i, obj in enumerate(iterable):
            yield obj
            it.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(it.global_avg * (n - i))))
                mem = torch.cuda.max_memory_allocated() / 2 ** 20 if torch.cuda.is_available() else 0
                print(f"{header} [{i}/{n}] eta: {eta} {self} time: {it} max mem: {mem:.0f}")
            end = time.time()
        print(f"{header} Total time: {datetime.timedelta(seconds=int(time.time() - start))}")


def restart_from_checkpoint(ckp_path, run_variables=None, **kwargs):
    """`utils.py:152-184`: load ``state_dict`` / ``optimizer`` / ``scheduler`` entries if the file exists."""
    if not os.path.isfile(ckp_path):
        return
    print(f"Found checkpoint at {ckp_path}")
    checkpoint = torch.load(ckp_path, map_location="cpu")
    for key, value in kwargs.items():
        if key in checkpoint and value is not None:
            try:
                sd = checkpoint[key]
                if hasattr(value, "named_parameters"):  # decoder keys carry DDP's "module." prefix (train.py:250)
                    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
                    msg = value.load_state_dict(sd, strict=False)
                else:
                    msg = value.load_state_dict(sd)
                print(f"=> loaded '{key}' from checkpoint '{ckp_path}' with msg {msg}")
            except (TypeError, ValueError, KeyError, RuntimeError) as e:
                print(f"=> failed to load '{key}' from checkpoint: '{ckp_path}' ({e})")
        else:
            print(f"=> key '{key}' not found in checkpoint: '{ckp_path}'")
    if run_variables is not None:
        for name in run_variables:
            if name in checkpoint:
                run_variables[name] = checkpoint[name]


def load_pretrained_weights(model, pretrained_weights, checkpoint_key):
    """`dinov2/utils/utils.py:20-33`: load a DINOv2 ``.pth`` into the backbone — take ``checkpoint_key`` ("teacher") when the
    file holds it, strip the ``module.`` (DDP) and ``backbone.`` (multi-crop wrapper) prefixes, ``strict=False`` (the
    checkpoints also carry the DINO / iBOT heads).  Local files only: there is no network on this path (an URL raises).
    Returns torch's (missing_keys, unexpected_keys) message."""
    from urllib.parse import urlparse
    if urlparse(str(pretrained_weights)).scheme not in ("", "file"):
        raise ValueError("load_pretrained_weights: URL checkpoints need a network; download the file and pass its path")
    state_dict = torch.load(pretrained_weights, map_location="cpu")
    if checkpoint_key is not None and isinstance(state_dict, dict) and checkpoint_key in state_dict:
        print(f"Take key {checkpoint_key} in provided checkpoint dict")
        state_dict = state_dict[checkpoint_key]
    state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
    state_dict = {k.replace("backbone.", ""): v for k, v in state_dict.items()}
    msg = model.load_state_dict(state_dict, strict=False)
    print("Pretrained weights found at {} and loaded with msg: {}".format(pretrained_weights, msg))
    return msg
