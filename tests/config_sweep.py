"""Engine vs oracle on a sweep of network configurations (fp32 exact mode + bf16): shapes with partial tiles, batch 1,
other activations, wider first level, anisotropic levels, multi-channel inputs / outputs.  GPU box only.

    python tests/config_sweep.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from biapy_amd.engine import NetConfig, ResUNetEngine
from oracle import net_oracle

DEV = "cuda"
CASES = [
    # (fm, patch, B, in_ch, out_ch, act, z_down)
    ([16, 32, 64], (16, 48, 80), 1, 1, (1,), "elu", None),
    ([16, 32, 64], (24, 40, 72), 2, 1, (1,), "elu", [1, 2]),
    ([32, 64], (20, 36, 44), 2, 1, (2,), "relu", None),
    ([16, 32, 64, 128], (32, 32, 48), 1, 16, (1,), "silu", [2, 1, 2]),
    ([16, 48, 96], (16, 32, 32), 2, 1, (3,), "elu", None),
    ([32, 32, 64], (8, 64, 64), 1, 1, (1,), "elu", [1, 1]),
    ([16, 32], (64, 64, 64), 1, 1, (1,), "elu", None),          # >= 64^3: lean persistent conv kernels
    ([16, 32], (66, 70, 68), 1, 1, (1,), "elu", None),          # ragged at 64^3+ (partial tiles in the lean kernels)
    ([48, 64, 80, 96], (32, 64, 64), 1, 1, (2,), "elu", [1, 1, 1]),   # Ovarian-Reserve template widths: GEMM-fed head (48 features), non-power-of-two levels
    ([52, 68, 84], (16, 64, 64), 2, 1, (2,), "elu", [1, 1]),          # CartoCell template widths: zero-padded to [64, 80, 96] inside the engine
    ([20, 36], (16, 32, 32), 1, 1, (1,), "gelu", None),                # padded widths under an activation with act(0) == 0 ...
    ([20, 36], (16, 32, 32), 1, 1, (1,), "sigmoid", None),             # ... and one with act(0) != 0 (the padded channels are read through zero weights)
]


def run(fm, patch, B, in_ch, out_ch, act, zd, dtype):
    torch.manual_seed(1)
    sd = net_oracle.init_state_dict(in_ch, fm, out_channels=out_ch, z_down=zd, seed=3) if "z_down" in net_oracle.init_state_dict.__code__.co_varnames \
        else net_oracle.init_state_dict(in_ch, fm, seed=3)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, in_ch, *patch, generator=g)
    tgt = (torch.rand(B, sum(out_ch), *patch, generator=g) > 0.5).float()
    kw = dict(feature_maps=fm, z_down=zd, activation=act) if "activation" in net_oracle.resunet_forward.__code__.co_varnames else dict(feature_maps=fm, z_down=zd)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo_ref = net_oracle.resunet_forward(params, x, **kw)
    loss_ref = F.binary_cross_entropy_with_logits(lo_ref, tgt)
    gref = torch.autograd.grad(loss_ref, list(params.values()))
    eng = ResUNetEngine(NetConfig(in_ch=in_ch, feature_maps=fm, out_channels=out_ch, activation=act, z_down=zd), dtype)
    P = {k: v.to(DEV) for k, v in sd.items()}
    xd = x.to(DEV)
    if in_ch > 1:
        xd = xd.contiguous(memory_format=torch.channels_last_3d)
    logits, ctx = eng.forward(P, xd, head_act=0, save=True)
    lg = logits.detach().clone().requires_grad_(True)
    F.binary_cross_entropy_with_logits(lg, tgt.to(DEV)).backward()
    G = eng.backward(P, ctx, lg.grad)
    torch.cuda.synchronize()
    e_lo = (logits.cpu() - lo_ref.detach()).abs().max().item() / lo_ref.abs().max().item()
    worst, wn = 0.0, ""
    gmax = max(gr.norm().item() for gr in gref)
    for (k, _), gr in zip(params.items(), gref):
        d = gr.norm().item()
        if d < 1e-5 * gmax:   # conv biases in front of an InstanceNorm: the true gradient is exactly zero, both sides hold rounding noise
            continue
        e = (G[k].cpu() - gr).norm().item() / d
        if e > worst:
            worst, wn = e, k
    return e_lo, worst, wn


bad = 0
for case in CASES:
    fm, patch, B, in_ch, out_ch, act, zd = case
    for dtype, tl, tg in ((torch.float32, 2e-4, 2e-3), (torch.bfloat16, 8e-2, 0.2)):
        try:
            e_lo, e_g, wn = run(fm, patch, B, in_ch, out_ch, act, zd, dtype)
            ok = e_lo < tl and e_g < tg
        except Exception as ex:  # noqa: BLE001
            e_lo, e_g, wn, ok = float("nan"), float("nan"), f"{type(ex).__name__}: {ex}", False
        bad += not ok
        print(f"{'ok  ' if ok else 'FAIL'} fm={fm} patch={patch} B={B} in={in_ch} out={out_ch} act={act} zd={zd} {str(dtype)[6:]:8s} logits {e_lo:.2e} grads {e_g:.2e} {wn}")
print("failures:", bad)
sys.exit(1 if bad else 0)
