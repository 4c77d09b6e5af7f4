"""Where does the bf16 mode's Dice difference come from?  A CPU emulation of the device data path (runs without a GPU).

The device kernels keep fp32 accumulators and fp32 InstanceNorm statistics; what bf16 mode rounds is (a) every tensor written to
HBM (raw conv outputs, pooled tensors, transposed-conv outputs, the last block's features), (b) the MFMA operands: the
normalised + activated value ``ELU(scale*x+shift)`` is rounded to bf16 again while it is staged, and the weights are packed as
bf16.  This script restates the ResUNet forward (oracle/net_oracle.py) with a rounding function at each of those points, trains
a small network on synthetic blobs on the CPU (the scenario of tests/kernel_checks.py::check_dice_parity_trained), and prints the
Dice difference to the fp32 forward with individual rounding points switched off - the error budget DESIGN.md section 5 quotes.

    python scripts/bf16_error_budget.py [--steps 120] [--seeds 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import net_oracle  # noqa: E402


def rb(x, kind):
    if kind == "f32":
        return x
    return x.to(torch.bfloat16 if kind == "bf16" else torch.float16).to(torch.float32)


def inorm_rec(acc, gamma, beta, eps=1e-5):
    """Statistics of the fp32 accumulators (what the producer kernel's epilogue sums) -> per-(n,c) scale / shift."""
    dims = tuple(range(2, acc.dim()))
    mean = acc.mean(dims, keepdim=True)
    var = acc.var(dims, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    shp = (1, -1) + (1,) * (acc.dim() - 2)
    scale = gamma.view(shp) * rstd
    return scale, beta.view(shp) - mean * scale


def forward(sd, x, fm, store="bf16", operand="bf16", weights="bf16", last_store=None, stats_of="acc"):
    """store: rounding of tensors written to HBM; operand: rounding of the activated MFMA operand; weights: rounding of packed
    weights; last_store: rounding of the final block's features (None = same as store); stats_of: "acc" (fp32 accumulators, as the
    kernels do) or "stored" (statistics of the rounded tensor)."""
    depth = len(fm) - 1
    last_store = store if last_store is None else last_store
    W = lambda k: rb(sd[k], weights)  # noqa: E731

    def conv3(a, key):
        return F.conv3d(a, W(key + ".weight"), sd[key + ".bias"], padding=1)

    def block(xs, rec, prefix, first, out_kind):
        """xs: stored raw input (or the fp32 image); rec: (scale, shift) of the pre-norm, None for the first block."""
        i = 0 if first else 2
        a = xs if rec is None else rb(F.elu(rec[0] * xs + rec[1]), operand)
        acc1 = conv3(a, f"{prefix}.block.{i}.block.0")
        h = rb(acc1, store)
        r1 = inorm_rec(acc1 if stats_of == "acc" else h, sd[f"{prefix}.block.{i}.block.1.weight"], sd[f"{prefix}.block.{i}.block.1.bias"])
        a2 = rb(F.elu(r1[0] * h + r1[1]), operand)
        acc2 = conv3(a2, f"{prefix}.block.{i + 1}.block.0") + F.conv3d(xs, W(f"{prefix}.shortcut.0.weight"), sd[f"{prefix}.shortcut.0.bias"])
        return acc2, rb(acc2, out_kind)

    skips = []
    cur, rec = x, None
    for i in range(depth):
        acc, out = block(cur, rec, f"down_path.{i}", i == 0, store)
        skips.append((acc, out))
        pooled = F.max_pool3d(out, 2)                        # max commutes with the rounding
        nxt = "bottleneck" if i == depth - 1 else f"down_path.{i + 1}"
        pacc = F.max_pool3d(acc, 2) if stats_of == "acc" else pooled
        # the fused pooling epilogue sums the ROUNDED pooled values (they are what the next conv reads)
        rec = inorm_rec(pooled if stats_of != "acc" else pooled, sd[f"{nxt}.block.0.weight"], sd[f"{nxt}.block.0.bias"])
        del pacc
        cur = pooled
    acc, cur = block(cur, rec, "bottleneck", False, store)
    for j, i in enumerate(range(depth - 1, -1, -1)):
        up_acc = F.conv_transpose3d(cur, W(f"up_paths.0.{j}.up.weight"), sd[f"up_paths.0.{j}.up.bias"], stride=2)
        up = rb(up_acc, store)
        sacc, sout = skips[i]
        cat = torch.cat([up, sout], 1)
        cat_acc = torch.cat([up_acc, sacc], 1) if stats_of == "acc" else cat
        pre = f"up_paths.0.{j}.conv_block"
        rec = inorm_rec(cat_acc, sd[f"{pre}.block.0.weight"], sd[f"{pre}.block.0.bias"])
        acc, cur = block(cat, rec, pre, False, last_store if i == 0 else store)
    return F.conv3d(cur, sd["heads.0.weight"], sd["heads.0.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--seeds", type=int, default=3)
    a = ap.parse_args()
    fm = [16, 32, 64]
    rows = {}
    for seed in range(a.seeds):
        g = torch.Generator().manual_seed(100 + seed)

        def batch(B):
            n = torch.randn(B, 1, 32, 32, 32, generator=g)
            t = (F.avg_pool3d(n, 5, stride=1, padding=2) > 0.05).float()
            return t * 1.5 + 0.8 * torch.randn(B, 1, 32, 32, 32, generator=g), t

        sd = net_oracle.init_state_dict(1, fm, seed=seed)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.AdamW(list(params.values()), lr=2e-3)
        for it in range(a.steps):
            x, t = batch(4)
            opt.zero_grad()
            loss = net_oracle.bce_with_logits(net_oracle.resunet_forward(params, x, fm), t)
            loss.backward()
            opt.step()
        sd = {k: v.detach() for k, v in params.items()}
        x, t = batch(8)
        with torch.no_grad():
            ref = net_oracle.resunet_forward(sd, x, fm)
            d_ref = net_oracle.dice(torch.sigmoid(ref), t)
            variants = {
                "f32 everywhere (emulator == oracle)": dict(store="f32", operand="f32", weights="f32"),
                "bf16 mode as built (storage + operands + weights)": dict(),
                "  ... statistics of the ROUNDED tensors instead of the accumulators": dict(stats_of="stored"),
                "  ... fp32 weights": dict(weights="f32"),
                "  ... fp32 operands (no second rounding after ELU)": dict(operand="f32"),
                "  ... fp32 storage (operands + weights still bf16)": dict(store="f32"),
                "  ... last block's features kept fp32 (head fused into the conv epilogue)": dict(last_store="f32"),
                "  ... only the weights rounded": dict(store="f32", operand="f32"),
                "fp16 storage + operands + weights (same 16 bits, 11-bit mantissa)": dict(store="f16", operand="f16", weights="f16"),
            }
            for name, kw in variants.items():
                lo = forward(sd, x, fm, **kw)
                d = net_oracle.dice(torch.sigmoid(lo), t)
                flips = ((lo > 0) != (ref > 0)).float().mean().item()
                rel = ((lo - ref).abs().max() / ref.abs().max()).item()
                rows.setdefault(name, []).append((abs(d - d_ref), flips, rel))
        print(f"seed {seed}: loss {loss.item():.4f}  dice_ref {d_ref:.5f}", flush=True)
    print(f"\n{'variant':86s} |Dice delta| (mean, max)   label flips   logits rel err")
    for name, v in rows.items():
        dd = [r[0] for r in v]
        print(f"{name:86s} {sum(dd) / len(dd):9.2e} {max(dd):9.2e}   {sum(r[1] for r in v) / len(v):9.2e}   {max(r[2] for r in v):9.2e}")


if __name__ == "__main__":
    main()
