"""Step drivers of the training path: ``train_one_epoch`` / ``evaluate``.

Host-side mirror of ``biapy/engine/train_engine.py:25-207`` (train) and ``:210-330`` (evaluate) for the single-loss,
single-optimizer case of the hot path.  Same loop semantics - ``optimizer.zero_grad()`` before the loop, per step: shape check
against ``DATA.PATCH_SIZE`` (same ``ValueError``), forward through ``model_call_func(batch, is_train=True)``, loss, backward,
optional ``clip_grad_norm_``, ``optimizer.step()``, a one-cycle scheduler stepped per iteration, ``zero_grad()``; a non-finite
loss stops training with ``sys.exit(1)``; the return value is ``({name: epoch average}, last step index)`` with the averages
taken over all ranks (``MetricLogger.synchronize_between_processes``).

What is different, because the step is ~12 ms on an MI355X and a host round trip per step would show:
  * the loss stays on the device; it is accumulated there and read back every ``sync_every`` steps (the reference's print
    frequency, 10), which is also when finiteness is checked - a NaN stops the run at most ``sync_every - 1`` steps later than
    the reference would;
  * with ``graph="auto"|"on"`` and a fixed batch shape the step is replayed from HIP graphs
    (``graphs.GraphedTrainStep``; ``graphs.DataParallelTrainStep`` when a process group is up - the model is then used
    unwrapped and its gradients are averaged by one flat all-reduce per step, which is what the DDP wrap of
    ``base_workflow.py:952-958`` does for the reference).  A ragged last batch, gradient clipping or a per-step scheduler fall
    back to the eager step.
The reference's ``cfg`` is not required: pass ``patch_size`` / ``gradient_clip_norm`` / ``lr_scheduler_name``, or a ``cfg``
object with ``DATA.PATCH_SIZE``, ``TRAIN.GRADIENT_CLIP_NORM`` and ``TRAIN.LR_SCHEDULER.NAME`` from which they are read.
Contrastive memory banks, multiple losses/optimizers and the warm-up-cosine per-step schedule stay on the reference's loop.
"""
from __future__ import annotations

import math
import sys
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.nn.utils import clip_grad_norm_


def _cfg_get(cfg, path: str, default):
    cur = cfg
    for part in path.split("."):
        if cur is None or not hasattr(cur, part):
            return default
        cur = getattr(cur, part)
    return cur


def to_pytorch_format(x: torch.Tensor, device) -> torch.Tensor:
    """(B,[Z,]Y,X,C) -> float32 (B,C,[Z,]Y,X) on ``device`` (biapy/utils/misc.py:689-713; a permuted view, channels-last strides)."""
    nd = x.dim()
    return x.to(device, non_blocking=True).to(torch.float32).permute(0, nd - 1, *range(1, nd - 1))


def _default_call(model, device):
    def call(batch, is_train=True):
        return model(to_pytorch_format(batch, device))

    return call


def _default_targets(device):
    def prep(targets, batch):
        return to_pytorch_format(targets, device)

    return prep


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _check_finite(acc: torch.Tensor, n: int) -> float:
    val = acc.item()                                   # the only host synchronisation of the loop
    if not math.isfinite(val):
        print("Loss is {}, stopping training".format(val / max(n, 1)))
        sys.exit(1)
    return val


def train_one_epoch(
    model: torch.nn.Module,
    loss_function: Callable,
    data_loader,
    optimizer: torch.optim.Optimizer,
    device: torch.device,
    epoch: int,
    cfg=None,
    model_call_func: Optional[Callable] = None,
    metric_function: Optional[Callable] = None,
    prepare_targets: Optional[Callable] = None,
    lr_scheduler=None,
    patch_size: Optional[Sequence[int]] = None,
    gradient_clip_norm: Optional[float] = None,
    lr_scheduler_name: Optional[str] = None,
    loss_name: str = "loss",
    graph: str = "auto",
    sync_every: int = 10,
    verbose: bool = False,
) -> Tuple[Dict[str, float], int]:
    patch_size = tuple(patch_size if patch_size is not None else _cfg_get(cfg, "DATA.PATCH_SIZE", ()))
    clip = float(gradient_clip_norm if gradient_clip_norm is not None else _cfg_get(cfg, "TRAIN.GRADIENT_CLIP_NORM", 0.0))
    sched_name = lr_scheduler_name if lr_scheduler_name is not None else _cfg_get(cfg, "TRAIN.LR_SCHEDULER.NAME", "")
    if sched_name in ("warmupcosine", "warmupreduceonplateau"):
        raise NotImplementedError("per-iteration warm-up schedules stay on the reference's train_one_epoch")
    device = torch.device(device)
    inner = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    call = model_call_func or _default_call(model, device)
    prep = prepare_targets or _default_targets(device)
    per_step_sched = lr_scheduler is not None and sched_name == "onecycle"
    capturable = all(g.get("capturable", False) for g in optimizer.param_groups)
    want_graph = graph == "on" or (graph == "auto" and device.type == "cuda" and capturable)
    can_graph = want_graph and device.type == "cuda" and clip <= 0 and not per_step_sched and model_call_func is None and metric_function is None
    if graph == "on" and not can_graph:
        raise ValueError("graph='on' needs a CUDA/HIP device, a capturable optimizer, no gradient clipping, no per-step scheduler and "
                         "the default model_call_func / metric_function")
    model.train()
    optimizer.zero_grad()
    gstep, gshape = None, None
    acc = torch.zeros((), dtype=torch.float64, device=device)          # running sum of the losses since the last read-back
    total, count, pending, step = 0.0, 0, 0, -1
    for step, (batch, targets) in enumerate(data_loader):
        if patch_size and tuple(batch.shape[1:-1]) != tuple(patch_size[:-1]):
            raise ValueError(
                "Trying to input data with different shape than 'DATA.PATCH_SIZE'. Check your configuration."
                f" Input: {batch.shape[1:-1]} vs PATCH_SIZE: {patch_size[:-1]}"
            )
        if can_graph:
            x, t = to_pytorch_format(batch, device), prep(targets, batch)
            if gstep is None:
                from . import graphs

                key = (id(optimizer), id(loss_function), tuple(x.shape), tuple(t.shape), _world())
                cached = getattr(inner, "_bpx_graph_step", None)
                if cached is not None and cached[0] == key:              # later epochs replay the graphs captured in the first one
                    gstep, gshape = cached[1], (tuple(x.shape), tuple(t.shape))
            if gstep is None:
                multi = _world() > 1
                if multi and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
                    graphs.broadcast_parameters_from_rank0(inner.parameters())    # a DDP wrap has done this already
                snap = _snapshot(inner, optimizer)                      # capture warms up with real optimizer steps: undo them
                if multi:
                    gstep = graphs.DataParallelTrainStep(inner, loss_function, optimizer, x, t, broadcast_parameters=False)
                else:
                    gstep = graphs.GraphedTrainStep(inner, loss_function, optimizer, x, t)
                gshape = (tuple(x.shape), tuple(t.shape))
                _restore(inner, optimizer, snap)
                inner._bpx_graph_step = (key, gstep)
            if (tuple(x.shape), tuple(t.shape)) == gshape:
                loss = gstep(x, t)
            else:                                                      # ragged last batch: same three phases, eagerly
                loss = _eager_step(inner, loss_function, optimizer, x, t)
        else:
            t = prep(targets, batch)
            outputs = call(batch, is_train=True)
            loss = loss_function(outputs, t)
            if metric_function is not None:
                metric_function(outputs, t)
            loss.backward()
            if clip > 0:
                clip_grad_norm_([p for g in optimizer.param_groups for p in g["params"]], max_norm=clip)
            optimizer.step()
            if per_step_sched:
                lr_scheduler.step()
            optimizer.zero_grad()
        acc += loss.detach().to(torch.float64)
        pending += 1
        if pending == sync_every:
            total += _check_finite(acc, pending)
            count += pending
            acc.zero_()
            pending = 0
            if verbose:
                print("Epoch: [{}]  [{}/{}]  {}: {:.4f}".format(epoch + 1, step + 1, len(data_loader), loss_name, total / count))
    if pending:
        total += _check_finite(acc, pending)
        count += pending
    stats = torch.tensor([total, float(count)], dtype=torch.float64, device=device)
    if _world() > 1:
        dist.all_reduce(stats)
    max_lr = max(float(g["lr"]) for g in optimizer.param_groups)
    out = {loss_name: (stats[0] / torch.clamp(stats[1], min=1.0)).item(), "lr": max_lr}
    print("[Train] averaged stats:", "  ".join(f"{k}: {v:.6f}" for k, v in out.items()))
    return out, step


def _snapshot(model, optimizer):
    params = [p.detach().clone() for p in model.parameters()]
    state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in optimizer.state.items()}
    return params, state


@torch.no_grad()
def _restore(model, optimizer, snap) -> None:
    """Puts parameters and optimizer state back IN PLACE (the captured graphs hold their addresses).  State that did not
    exist before the warm-up (a fresh optimizer) is zeroed, which is its initial value for Adam(W) / momentum SGD."""
    params, state = snap
    for p, s in zip(model.parameters(), params):
        p.copy_(s)
    for p, st in optimizer.state.items():
        old = state.get(id(p))
        for k, v in st.items():
            if torch.is_tensor(v):
                if old is not None and torch.is_tensor(old.get(k)):
                    v.copy_(old[k])
                else:
                    v.zero_()
    torch.cuda.synchronize()


def _eager_step(model, loss_function, optimizer, x, t):
    """One eager step for a batch the captured graphs do not fit (ragged last batch); gradients averaged over the ranks."""
    optimizer.zero_grad(set_to_none=True)                              # p.grad may still alias a graph's private gradient buffers
    loss = loss_function(model(x), t)
    loss.backward()
    if _world() > 1:
        grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        pack = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(pack)
        pack.mul_(1.0 / _world())
        off = 0
        for g in grads:
            g.copy_(pack[off:off + g.numel()].view_as(g))
            off += g.numel()
    optimizer.step()
    optimizer.zero_grad()
    return loss


@torch.no_grad()
def evaluate(
    model: torch.nn.Module,
    loss_function: Callable,
    data_loader,
    device: torch.device,
    epoch: int,
    cfg=None,
    model_call_func: Optional[Callable] = None,
    metric_function: Optional[Callable] = None,
    prepare_targets: Optional[Callable] = None,
    lr_scheduler=None,
    lr_scheduler_name: Optional[str] = None,
    loss_name: str = "loss",
) -> Dict[str, float]:
    """Validation pass (train_engine.py:210-330): eval mode, loss (and ``metric_function`` values, a dict of 0-d tensors or
    floats per batch) averaged over the loader and over the ranks; steps a ``ReduceLROnPlateau`` scheduler with the loss."""
    device = torch.device(device)
    sched_name = lr_scheduler_name if lr_scheduler_name is not None else _cfg_get(cfg, "TRAIN.LR_SCHEDULER.NAME", "")
    call = model_call_func or _default_call(model, device)
    prep = prepare_targets or _default_targets(device)
    model.eval()
    sums: Dict[str, torch.Tensor] = {}
    n = 0
    for batch, targets in data_loader:
        t = prep(targets, batch)
        outputs = call(batch, is_train=True)
        vals = {loss_name: loss_function(outputs, t)}
        if metric_function is not None:
            vals.update(metric_function(outputs, t) or {})
        for k, v in vals.items():
            v = torch.as_tensor(v, device=device).detach().to(torch.float64)
            sums[k] = sums[k] + v if k in sums else v.clone()
        n += 1
    keys = sorted(sums)
    stats = torch.stack([sums[k] for k in keys] + [torch.tensor(float(n), dtype=torch.float64, device=device)]) if keys else torch.zeros(1, dtype=torch.float64, device=device)
    if _world() > 1:
        dist.all_reduce(stats)
    host = stats.tolist()                                              # one read-back for the whole pass
    cnt = max(host[-1], 1.0)
    out = {k: host[i] / cnt for i, k in enumerate(keys)}
    if loss_name in out and not math.isfinite(out[loss_name]):
        print("Loss is {}, stopping training".format(out[loss_name]))
        sys.exit(1)
    print("[Val] averaged stats:", "  ".join(f"{k}: {v:.6f}" for k, v in out.items()))
    if lr_scheduler is not None and sched_name == "reduceonplateau" and loss_name in out:
        lr_scheduler.step(out[loss_name])
    return out
