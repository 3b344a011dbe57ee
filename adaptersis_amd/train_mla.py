"""Drop-in for the reference's `train_mla.py`: the `train.py` pipeline with the MLA decode head
(`backbones/decoders.py:48-89`), the block -> CACNN -> CAViT adapter order of `train_mla.py:300-383` (selected by the
engine from the decoder type) and its optimiser settings (`train_mla.py:178-184`: lr * batch * world / 16, momentum 0.9,
no weight decay).  Same CLI except the spelling ``--local_rank`` (`train_mla.py:609`).

    python -m adaptersis_amd.train_mla --arch vit_large --imsize 588 --batch_size_per_gpu 12 --data_path synthetic
"""
from __future__ import annotations

from . import train as _t

train = _t.train                          # `train_mla.py:260-407`: the loop body is SegEngine.train_step (MLA flow)
validate_network = _t.validate_network    # `train_mla.py:410-600`


def train_seg(args):
    return _t.train_seg(args, head="mla")


def get_args_parser():
    p = _t.get_args_parser()
    for a in list(p._actions):
        if "--local-rank" in a.option_strings:
            p._handle_conflict_resolve(None, [("--local-rank", a)])
    p.add_argument("--local_rank", default=0, type=int)
    return p


if __name__ == "__main__":
    train_seg(get_args_parser().parse_args())
