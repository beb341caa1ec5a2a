"""How much do cheaper products inside the ViT attention move the CPU oracle's logits / tokens?  (Not a pytest: run by hand,
    python tests/probe_vit_attention_precision.py      # ~5 minutes on 8 cores
Test infrastructure: it imports the oracle, so it lives under tests/.)  Variants: the softmax probabilities rounded to fp16 before
P.V (the split kernel would save the p_lo . v_hi product and half of the P-split arithmetic), V rounded to fp16 in P.V, K rounded
to fp16 in Q.K^T, Q and K both."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from _util import golden_inputs
from oracle import blip_ref as R

torch.set_num_threads(8)
g, meta, arch, sd, px = golden_inputs("blip_base256")
N = 48
px = px[:N]
base = R.greedy_generate(sd, arch, px, 20)
orig = R.vision_layer
h16 = lambda t: t.half().float()


def make(qf, kf, pf, vf):
    def layer(sd, arch, i, x):
        p = f"vision_model.encoder.layers.{i}."
        B, NT, D = x.shape
        H, hd = arch.v_heads, arch.v_hidden // arch.v_heads
        h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], arch.v_eps)
        qkv = F.linear(h, sd[p + "self_attn.qkv.weight"], sd[p + "self_attn.qkv.bias"])
        qkv = qkv.reshape(B, NT, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = torch.matmul(qf(q), kf(k).transpose(-1, -2)) * (hd ** -0.5)
        a = torch.softmax(s, dim=-1)
        ctx = torch.matmul(pf(a), vf(v)).permute(0, 2, 1, 3).reshape(B, NT, D)
        x = x + F.linear(ctx, sd[p + "self_attn.projection.weight"], sd[p + "self_attn.projection.bias"])
        h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], arch.v_eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return layer


def run(name, layer):
    R.vision_layer = layer
    out = R.greedy_generate(sd, arch, px, 20)
    R.vision_layer = orig
    L = min(out["sequences"].shape[1], base["sequences"].shape[1])
    same = (out["sequences"][:, :L] == base["sequences"][:, :L]).all(dim=1).sum().item()
    errs = []
    for t in range(min(len(out["logits"]), len(base["logits"]))):
        agree = (out["sequences"][:, :t + 1] == base["sequences"][:, :t + 1]).all(dim=1)
        if agree.any():
            errs.append((out["logits"][t][agree] - base["logits"][t][agree]).abs().max().item())
    emb = (out["image_embeds"] - base["image_embeds"]).abs().max().item()
    print(f"{name:34s} rows identical {same}/{N}  max |dlogit| {max(errs):.3e}  max |d image_embeds| {emb:.3e}", flush=True)


ident = lambda t: t
run("restated layer, nothing rounded", make(ident, ident, ident, ident))
run("P fp16 in P.V", make(ident, ident, h16, ident))
run("V fp16 in P.V", make(ident, ident, ident, h16))
run("P and V fp16", make(ident, ident, h16, h16))
run("K fp16 in Q.K^T", make(ident, h16, ident, ident))
run("Q and K fp16", make(h16, h16, ident, ident))
