"""How much does quantising the cross-attention K/V cache move the CPU oracle's logits / tokens?  (Not a pytest: run by hand,
    python tests/probe_kv_quant.py        # ~6 minutes on 8 cores
The output of the run that picked the KV16 layout is committed as profiles/r03_kv_quant_probe.txt.)  Test infrastructure: it imports
the oracle, so it lives under tests/."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import golden_inputs
from oracle import blip_ref as R

torch.set_num_threads(8)
g, meta, arch, sd, px = golden_inputs("blip_base256")
N = 48
px = px[:N]
emb = R.encode_image(sd, arch, px)
base = R.greedy_generate(sd, arch, px, 20, image_embeds=emb)
orig = R.cross_kv

def q_kv24(x):
    i = x.contiguous().view(torch.int32)
    i = ((i + 0x80) >> 8) << 8
    return i.view(torch.float32)
def q_fp16(x): return x.half().float()
def q_bf16(x): return x.bfloat16().float()
def q_int16(x):
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 32767.0
    return torch.round(x / s) * s
def q_int12(x):
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 2047.0
    return torch.round(x / s) * s

def run(fk, fv, name):
    def patched(sd_, arch_, image_embeds, state, repeat=1):
        orig(sd_, arch_, image_embeds, state, repeat)
        for i in range(arch_.t_layers):
            state.cross_k[i] = fk(state.cross_k[i]); state.cross_v[i] = fv(state.cross_v[i])
    R.cross_kv = patched
    out = R.greedy_generate(sd, arch, px, 20, image_embeds=emb)
    R.cross_kv = orig
    L = min(out["sequences"].shape[1], base["sequences"].shape[1])
    same = (out["sequences"][:, :L] == base["sequences"][:, :L]).all(dim=1).sum().item()
    # logit error on the steps where the prefixes still agree (step 0 always)
    errs = []
    for t in range(min(len(out["logits"]), len(base["logits"]))):
        agree = (out["sequences"][:, :t + 1] == base["sequences"][:, :t + 1]).all(dim=1)
        if agree.any():
            errs.append((out["logits"][t][agree] - base["logits"][t][agree]).abs().max().item())
    print(f"{name:28s} rows identical {same}/{N}  max |dlogit| {max(errs):.3e}  step0 {errs[0]:.3e}", flush=True)

ident = lambda x: x
run(q_kv24, q_kv24, "kv24 / kv24")
run(q_int16, q_int16, "int16-block / int16-block")
run(q_kv24, q_int16, "kv24 K / int16-block V")
run(q_fp16, q_fp16, "fp16 / fp16")
run(q_int12, q_int12, "int12-block")
run(q_bf16, q_bf16, "bf16 / bf16")
# margins of the baseline for context
m = []
for t, lg in enumerate(base["logits"]):
    top2 = lg.topk(2, dim=-1).values
    m.append((top2[:, 0] - top2[:, 1]))
m = torch.stack(m)
print("baseline min top-2 margin over steps/rows:", m.min().item(), " 1st percentile:", m.flatten().kthvalue(max(1, m.numel() // 100)).values.item())
