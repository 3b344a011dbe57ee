"""CPU probe (not a test; run by hand: ``python tests/precision_probe.py [arch] [size]``): which 16-bit rounding sites of the
HIP step put how much error on the adapter-stream / pyramid features?  The fp32 oracle is re-run with fp16 rounding
emulated at one group of sites at a time (operands of every GEMM, stored 16-bit intermediates) — what the MFMA path does
with fp32 accumulation — and the rel-L2 of ``x_final`` / ``c_final`` against the exact oracle is printed per site group.
DESIGN.md §3 quotes its output; it decides where split-precision operands are worth their cost."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from adaptersis_amd.utils import weights as W  # noqa: E402
from oracle import ref_torch as O  # noqa: E402

SITES = ["pe", "vit.ln", "vit.w.qkv", "vit.w.proj", "vit.w.fc1", "vit.w.fc2", "vit.qkv", "vit.p", "vit.o", "vit.h",
         "msda.ln", "msda.w", "msda.value", "msda.oa", "msda.samp", "cffn.ln", "cffn.w", "cffn.g"]
ON = set()
DT = torch.float16


def r(x, site):
    return x.to(DT).float() if site in ON else x


def lin(x, w, b, sa, sw):
    return F.linear(r(x, sa), r(w, sw), b)


def attention(x, sd, p, num_heads):
    B, N, C = x.shape
    hd = C // num_heads
    qkv = r(lin(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias"), "vit.ln", "vit.w.qkv"), "vit.qkv")
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = r(((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1), "vit.p")
    o = r((attn @ v).transpose(1, 2).reshape(B, N, C), "vit.o")
    return F.linear(o, r(sd[p + ".proj.weight"], "vit.w.proj"), sd.get(p + ".proj.bias"))


def mlp(x, sd, p):
    if p + ".fc1.weight" in sd:
        h = r(F.gelu(lin(x, sd[p + ".fc1.weight"], sd.get(p + ".fc1.bias"), "vit.ln", "vit.w.fc1")), "vit.h")
        return F.linear(h, r(sd[p + ".fc2.weight"], "vit.w.fc2"), sd.get(p + ".fc2.bias"))
    x12 = lin(x, sd[p + ".w12.weight"], sd.get(p + ".w12.bias"), "vit.ln", "vit.w.fc1")
    x1, x2 = x12.chunk(2, dim=-1)
    return F.linear(r(F.silu(x1) * x2, "vit.h"), r(sd[p + ".w3.weight"], "vit.w.fc2"), sd.get(p + ".w3.bias"))


def ms_deform_attn(query, reference_points, feat, spatial_shapes, sd, p, n_heads, n_levels, n_points):
    N, Lq, C = query.shape
    _, Lin, _ = feat.shape
    value = r(lin(feat, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"], "msda.ln", "msda.w"), "msda.value")
    value = value.view(N, Lin, n_heads, C // n_heads)
    qa, wa = ("msda.ln", "msda.w") if "msda.oa" in ON else ("-", "-")
    off = lin(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"], qa, wa)
    off = off.view(N, Lq, n_heads, n_levels, n_points, 2)
    aw = lin(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"], qa, wa)
    aw = F.softmax(aw.view(N, Lq, n_heads, n_levels * n_points), -1).view(N, Lq, n_heads, n_levels, n_points)
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(query.dtype)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = r(O.ms_deform_attn_core(value, spatial_shapes, loc, aw), "msda.samp")
    return F.linear(out, r(sd[p + ".output_proj.weight"], "msda.w"), sd[p + ".output_proj.bias"])


def conv_ffn(x, sd, p, grids):
    x = lin(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"], "cffn.ln", "cffn.w")
    x = r(F.gelu(O.dwconv(x, sd, p + ".dwconv", grids)), "cffn.g")
    return F.linear(x, r(sd[p + ".fc2.weight"], "cffn.w"), sd[p + ".fc2.bias"])


def patch_embed(img, sd, patch=14):
    if "pe" not in ON:
        return _patch_embed(img, sd, patch)
    sd2 = dict(sd)
    sd2["patch_embed.proj.weight"] = r(sd["patch_embed.proj.weight"], "pe")
    return _patch_embed(r(img, "pe"), sd2, patch)


_patch_embed = O.patch_embed


HEAD = None   # "unet": also push x_final through the fp32 oracle UNet(D) (BASELINE config 2) and report its logits;
              # "mla": the `train_mla.py` flow (BASELINE config 5) -> DecoderMLA, 11 classes: MLA input 0 / output
ONLY = None   # --sites a,b,c: restrict the per-site loop


def run(img, sds, heads):
    taps = {}
    if HEAD == "mla":
        with torch.no_grad():
            maps = O.mla_forward(img, sds["vit"], {k: v.clone() for k, v in sds["enc"].items()}, sds["cv"], sds["cn"], heads)
            out = O.decoder_mla(*maps, sd={k: v.clone() for k, v in sds["mla"].items()}, img_size=img.shape[-1], update_bn=False)
        return maps[0], maps[3], out
    with torch.no_grad():
        cat = O.adapter_forward(img, sds["vit"], {k: v.clone() for k, v in sds["enc"].items()}, sds["cv"], sds["cn"], heads, taps=taps)
        x = taps["x_stage3"]
        if HEAD == "fdec":   # the headline head: `train.py:389-406` decoder input -> FeatureDecoder (train-mode BatchNorm) logits
            y = O.feature_decoder(cat, {k: v.clone() for k, v in sds["fdec"].items()}, update_bn=False)
            return x, taps["c_stage3"], y
        if HEAD == "unet":
            B, N, D = x.shape
            h = int(N ** 0.5)
            y = O.unet(x.transpose(1, 2).reshape(B, D, h, h), {k: v.clone() for k, v in sds["unet"].items()}, update_bn=False)
            return x, taps["c_stage3"], y
    return x, taps["c_stage3"], taps["feats"][-1]


def main():
    global DT, HEAD
    global ONLY
    if "--unet" in sys.argv:
        sys.argv.remove("--unet")
        HEAD = "unet"
    if "--mla" in sys.argv:
        sys.argv.remove("--mla")
        HEAD = "mla"
    if "--fdec" in sys.argv:
        sys.argv.remove("--fdec")
        HEAD = "fdec"
    if "--sites" in sys.argv:
        i = sys.argv.index("--sites")
        ONLY = sys.argv[i + 1].split(",")
        del sys.argv[i:i + 2]
    arch = sys.argv[1] if len(sys.argv) > 1 else "vit_base_d4"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    if len(sys.argv) > 3 and sys.argv[3] == "bf16":
        DT = torch.bfloat16
    D, depth, heads, _ = W.VIT_CONFIGS[arch]
    sds = dict(vit=W.make_vit_state_dict(arch, layerscale="kernel"), enc=W.make_encoder_state_dict(D),
               cv=W.make_cavit_state_dict(D, mode="kernel"), cn=W.make_cacnn_state_dict(D, mode="kernel"))
    if HEAD == "unet":
        sds["unet"] = W.make_unet_state_dict(D, 2)
    if HEAD == "mla":
        sds["mla"] = W.make_decoder_mla_state_dict(D, 128, 11)
    if HEAD == "fdec":
        sds["fdec"] = W.make_feature_decoder_state_dict(D, 2, features=(D, 512, 256, 128, 64))
    img, _ = W.synthetic_batch(1, size)
    ref = run(img, sds, heads)
    O.attention, O.mlp, O.ms_deform_attn, O.conv_ffn, O.patch_embed = attention, mlp, ms_deform_attn, conv_ffn, patch_embed

    def err(tag):
        out = run(img, sds, heads)
        e = [float((a.double() - b.double()).norm() / b.double().norm()) for a, b in zip(out, ref)]
        names = {"unet": ("x_final", "c_final", "UNet logits"), "mla": ("MLA in0", "MLA in3", "MLA output"),
                 "fdec": ("x_final", "c_final", "FeatureDecoder logits")}.get(HEAD, ("x_final", "c_final", "passA feat"))
        print(f"{tag:34s} {names[0]} {e[0]:.2e}  {names[1]} {e[1]:.2e}  {names[2]} {e[2]:.2e}", flush=True)
        return e
    print(f"{arch} {size}x{size}, operands {DT}: rounding emulated at the listed sites only")
    ON.clear(); ON.update(SITES)
    err("ALL sites")
    ON.clear(); ON.update(s for s in SITES if s != "pe")
    err("ALL but pe (split-precision patch embed)")
    for s in (ONLY if ONLY is not None else SITES):
        ON.clear(); ON.update(s.split("+"))
        err("only " + s)
    if ONLY is not None:
        return
    for grp in ("vit", "msda", "cffn"):
        ON.clear(); ON.update(s for s in SITES if s.startswith(grp))
        err(f"group {grp}")
    ON.clear(); ON.update(s for s in SITES if s != "msda.oa")
    err("ALL but msda.oa (split offsets GEMM)")
    ON.clear(); ON.update(s for s in SITES if not s.startswith("vit.w"))
    err("ALL but the ViT weights")
    ON.clear(); ON.update(s for s in SITES if s.startswith("vit.w"))
    err("only the ViT weights")


if __name__ == "__main__":
    main()
