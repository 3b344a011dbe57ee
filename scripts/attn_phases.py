"""Per-phase cycle counts of one attention-forward wave (lab build: ASIS_ATTN_ABLATE=5 dumps s_memtime stamps)."""
import os, sys
PIPE = os.environ.get("ASIS_ATTN_PIPE", "0") != "0"
os.environ["ASIS_ATTN_ABLATE"] = "6" if PIPE else "5"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def main():
    dev = torch.device("cuda:0")
    B, H, N, D = 12, 16, 1764, 1024
    dt = torch.float16
    M = B * N
    qk = torch.randn(M, 2 * D, device=dev).to(dt)
    vt = torch.zeros(B, D, 1792, device=dev, dtype=dt)
    vt[:, :, :N] = torch.randn(B, D, N, device=dev).to(dt)
    o = torch.empty(M, D, device=dev, dtype=dt)
    lse = torch.zeros(B * H * N, device=dev)
    for _ in range(3):
        ops.attention_fwd(qk[:, :D], qk[:, D:], vt, B, H, N, 0.125, out=o, lse=lse)
    torch.cuda.synchronize()
    raw = lse.view(torch.int64)[: 2 * 64 * 8].view(2, 64, 8).cpu()
    ck = lse.view(torch.int64)[2 * 64 * 8: 2 * 64 * 8 + 4].cpu().double()
    for w in range(2):
        print(f"block {w}: {ck[2 * w]:.0f} s_memtime ticks in {ck[2 * w + 1]:.0f} s_memrealtime ticks (100 MHz) -> "
              f"{ck[2 * w] / ck[2 * w + 1] * 100:.0f} MHz")
    names = ["S(K reads+MFMA+max)", "softmax(exp,cvt)", "PV (to completion)", "store_tile(vm wait)", "barrier", "load_tile issue", "loop head->tile"]
    if PIPE:
        pn = ["DMA issue", "K reads + max(S(t))", "rescale", "S(t+1) MFMA + exp(t)", "P.V(t) to completion", "vmcnt(0)",
              "barrier"]
        for w in range(2):
            t = raw[w, :28].double()
            d = torch.stack([t[:, i + 1] - t[:, i] for i in range(7)], 1)
            per_tile = t[1:, 0] - t[:-1, 0]
            print(f"pipe kernel, block z={'0' if w == 0 else 'B/2'}: mean ticks per tile {per_tile[2:-1].mean():.0f}")
            for i in range(7):
                print(f"   {pn[i]:24s} {d[2:-1, i].mean():8.0f}  (min {d[2:-1, i].min():.0f} max {d[2:-1, i].max():.0f})")
        return
    for w in range(2):
        t = raw[w, :28].double()
        d = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4],
                         t[:, 7] - t[:, 6], t[:, 0] - t[:, 7]], 1)
        per_tile = (t[1:, 0] - t[:-1, 0])
        print(f"block z={'0' if w == 0 else 'B/2'}: mean s_memtime ticks per tile {per_tile[2:-1].mean():.0f}")
        for i in range(7):
            print(f"   {names[i]:22s} {d[2:-1, i].mean():8.0f}  (min {d[2:-1, i].min():.0f} max {d[2:-1, i].max():.0f})")


if __name__ == "__main__":
    main()
