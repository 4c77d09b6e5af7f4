"""``optimizer.step()`` of the training loop (reference: biapy/engine/train_engine.py:173-177) for ``torch.optim.Adam`` / ``AdamW`` through
``bpx_adam_step``: the same update, in the arithmetic order of torch's fused kernel, written into the optimizer's OWN state tensors
(``exp_avg``, ``exp_avg_sq``, ``step``) - ``state_dict()`` / ``load_state_dict()``, schedulers and a later plain ``optimizer.step()`` see
nothing unusual.  torch's multi-tensor launch gives one 64 K-element chunk to a block: the 6.7 M parameters of cfg 2 are ~200 blocks on 256 CUs
and three launches of 45 us; ``bpx_adam_step`` uses 4096-element blocks (two launches plus the step increment).

``fused_step(optimizer)`` returns False - and does nothing - for anything it does not reproduce exactly (another optimizer class, amsgrad,
maximize, host-side ``step`` counters, tensor-valued betas / eps / weight decay, registered step hooks, state not yet initialised, non-fp32 or
non-contiguous tensors, sparse gradients, a missing gradient) - decided for every group before the first launch, so a step is never half done:
the caller then runs ``optimizer.step()`` itself.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L

lib = L.lib

_ENABLED = os.environ.get("BPX_FUSED_ADAM", "1") != "0"


def _group_ok(opt, g) -> bool:
    if g.get("amsgrad", False) or g.get("maximize", False) or g.get("differentiable", False):
        return False
    if not g.get("capturable", False):       # host-side step counters: torch's own path
        return False
    if any(torch.is_tensor(v) for v in (*g["betas"], g["eps"], g["weight_decay"])):
        return False
    lr = g["lr"]
    if torch.is_tensor(lr) and lr.is_cuda and (lr.dtype != torch.float32 or lr.numel() != 1):
        return False
    for p in g["params"]:
        if p.grad is None:
            return False
        st = opt.state.get(p)
        if not st or "exp_avg" not in st or "exp_avg_sq" not in st or not torch.is_tensor(st.get("step")):
            return False
        ts = (p, p.grad, st["exp_avg"], st["exp_avg_sq"])
        if any((not t.is_cuda) or t.dtype != torch.float32 or not t.is_contiguous() or t.is_sparse for t in ts):
            return False
        if st["step"].dtype != torch.float32 or not st["step"].is_cuda or st["step"].numel() != 1:
            return False
        if any(t.numel() != p.numel() for t in ts):
            return False
    return True


def _has_step_hooks(opt) -> bool:
    import torch.optim.optimizer as O

    return bool(getattr(opt, "_optimizer_step_pre_hooks", None) or getattr(opt, "_optimizer_step_post_hooks", None)
                or getattr(O, "_global_optimizer_pre_hooks", None) or getattr(O, "_global_optimizer_post_hooks", None))


@torch.no_grad()
def fused_step(optimizer: torch.optim.Optimizer) -> bool:
    """One optimizer step through ``bpx_adam_step``; False (nothing done) when the optimizer is not an Adam(W) this kernel reproduces."""
    if not _ENABLED or type(optimizer) not in (torch.optim.Adam, torch.optim.AdamW):
        return False
    if _has_step_hooks(optimizer):            # hooks hang on optimizer.step(): torch's own path runs them
        return False
    groups = optimizer.param_groups
    if not all(_group_ok(optimizer, g) for g in groups):     # every refusal is decided BEFORE the first launch: no partial step
        return False
    st = L.stream_ptr()
    for g in groups:
        decoupled = 1 if isinstance(optimizer, torch.optim.AdamW) or g.get("decoupled_weight_decay", False) else 0
        ps = [p for p in g["params"]]
        if not ps:
            continue
        arr = (L.AdamTensor * len(ps))()
        for i, p in enumerate(ps):
            s = optimizer.state[p]
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()
            arr[i].step, arr[i].numel = s["step"].data_ptr(), p.numel()
        lr = g["lr"]
        lr_d, lr_h = (lr.data_ptr(), 0.0) if torch.is_tensor(lr) and lr.is_cuda else (None, float(lr))
        b1, b2 = g["betas"]
        L.check(lib.bpx_adam_step(len(ps), arr, lr_d, lr_h, float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), decoupled, st))
    optimizer._opt_called = True             # what lr_scheduler's wrapper of optimizer.step() records (its "scheduler before optimizer" warning reads it)
    return True


def step(optimizer: torch.optim.Optimizer) -> None:
    """``optimizer.step()``, through the HIP kernel where it applies."""
    if not fused_step(optimizer):
        optimizer.step()
