"""Step drivers of the training path: ``train_one_epoch`` / ``evaluate``.

Call-compatible with ``biapy/engine/train_engine.py:25-43`` (train) and ``:210-224`` (evaluate): the unchanged call sites of
``Base_Workflow.train`` (``biapy/engine/base_workflow.py:1070-1088`` and ``:1114-1126``) work against this module - positional
``cfg``, ``optimizer`` / ``lr_scheduler`` as lists, ``log_writer``, ``memory_bank``, ``total_iters``, ``contrast_warmup_iters``,
``loss_names``; the return value is ``({meter name: epoch average}, last step index)`` with one meter per loss name, one per
learning-rate name (``loss`` -> ``lr``, :92) and whatever ``metric_function(outputs, targets, metric_logger=...)`` records,
averaged over all ranks (``MetricLogger.synchronize_between_processes``).

Same loop semantics as the reference: ``zero_grad()`` on every optimizer before the loop; per step the per-iteration warm-up
schedules (:118-121), ``prepare_targets``, the shape check against ``DATA.PATCH_SIZE`` (same ``ValueError``),
``model_call_func(batch, is_train=True)``, the loss (a tensor, or ``{"losses": [...], "metrics": {...}}``), for every loss
``backward`` -> optional ``clip_grad_norm_`` -> ``optimizer[i].step()`` -> one-cycle scheduler -> ``zero_grad()``; a
non-finite loss stops training with ``sys.exit(1)``.

What is different, because the step is ~10 ms on an MI355X and a host round trip per step would show:
  * the losses stay on the device; they are accumulated there and read back every ``sync_every`` steps (the reference's print
    frequency, 10), which is also when finiteness is checked and the ``log_writer`` is fed (with the window mean) - a NaN stops
    the run at most ``sync_every - 1`` steps later than the reference would; INTEGRATION.md says what that means for
    checkpoints written from an exit hook;
  * with ``graph="auto"|"on"`` the step of a biapy_amd model is replayed from HIP graphs (``graphs.GraphedTrainStep``;
    ``graphs.DataParallelTrainStep`` when a process group is up - the model is then used unwrapped and its gradients are
    averaged by one flat all-reduce per step, which is what the DDP wrap of ``base_workflow.py:952-958`` does for the
    reference).  The replayed step bypasses ``model_call_func``, which is only legal where that function is the identity around
    the model in training mode (``to_pytorch_format`` -> model -> no resize, no training-time activation:
    ``base_workflow.py:855-892`` with ``ce_sigmoid`` / ``ce_softmax`` / ``linear`` heads, :1427; or a loss that applies the model's head
    activations inside its own kernel, ``losses.InstanceChannelsLoss`` built with the same list) - anything else, a ragged last
    batch, gradient clipping, per-step schedulers, several losses or a memory bank run the eager step.  The optimizer's ``lr``
    is turned into a device scalar so that scheduler updates between epochs reach the captured optimizer step.
Contrastive memory banks stay on the reference's loop (``NotImplementedError`` here, as the model classes raise for ``contrast``).
"""
from __future__ import annotations

import math
import sys
from collections import defaultdict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.nn.utils import clip_grad_norm_
from torch.optim.lr_scheduler import OneCycleLR, ReduceLROnPlateau


def _cfg_get(cfg, path: str, default):
    cur = cfg
    for part in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            if part not in cur:
                return default
            cur = cur[part]
        elif hasattr(cur, part):
            cur = getattr(cur, part)
        else:
            return default
    return cur


def to_pytorch_format(x: torch.Tensor, device) -> torch.Tensor:
    """(B,[Z,]Y,X,C) -> float32 (B,C,[Z,]Y,X) on ``device`` (biapy/utils/misc.py:689-713; a permuted view, channels-last strides)."""
    nd = x.dim()
    return x.to(device, non_blocking=True).to(torch.float32).permute(0, nd - 1, *range(1, nd - 1))


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class SmoothedValue:
    """Series total / count (biapy/utils/misc.py:890-961; only the global average is used by the epoch drivers)."""

    def __init__(self, window_size: int = 20, fmt: Optional[str] = None):
        self.total, self.count, self.last = 0.0, 0, 0.0
        self.fmt = fmt or "{global_avg:.4f}"

    def update(self, value, n: int = 1):
        value = float(value)
        self.last = value
        self.count += n
        self.total += value * n

    @property
    def global_avg(self) -> float:
        return self.total / (self.count + sys.float_info.epsilon)

    @property
    def value(self) -> float:
        return self.last

    def __str__(self):
        return self.fmt.format(global_avg=self.global_avg, value=self.last, median=self.last, avg=self.global_avg, max=self.last)


class MetricLogger:
    """The part of biapy/utils/misc.py:1001-1150 the epoch drivers and the workflows' ``metric_calculation`` use: ``meters``,
    ``add_meter``, ``update(**{name: value})`` and the cross-rank synchronisation of totals and counts (ONE all-reduce for all
    meters; the reference does one per meter)."""

    def __init__(self, delimiter: str = "  ", verbose: bool = False):
        self.meters = defaultdict(SmoothedValue)      # workflows write ``metric_logger.meters[name].update(v)`` (semantic_seg.py:392)
        self.delimiter, self.verbose = delimiter, verbose

    def add_meter(self, name: str, meter: SmoothedValue):
        self.meters[name] = meter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if torch.is_tensor(v):
                v = v.item()
            self.meters[k].update(v)

    def synchronize_between_processes(self, device=None):
        if _world() == 1 or not self.meters:
            return
        names = list(self.meters)
        t = torch.tensor([[self.meters[k].count, self.meters[k].total] for k in names], dtype=torch.float64,
                         device=device if device is not None else "cpu")
        dist.all_reduce(t)
        for k, (c, tot) in zip(names, t.tolist()):
            self.meters[k].count, self.meters[k].total = int(c), tot

    def __str__(self):
        return self.delimiter.join("{}: {}".format(k, str(m)) for k, m in self.meters.items())


def _as_list(v):
    if v is None:
        return []
    return list(v) if isinstance(v, (list, tuple)) else [v]


def _losses_of(result) -> Tuple[List[torch.Tensor], Dict]:
    """train_engine.py:152-158: a loss function returns a tensor or {"losses": [...], "metrics": {...}}."""
    if isinstance(result, dict):
        return list(result.get("losses", [])), dict(result.get("metrics", {}))
    return [result], {}


_GRAPH_SAFE_ACTS = ("ce_sigmoid", "ce_softmax", "linear")


def _check_fused_activations(model_call_func, loss_function) -> None:
    """A loss that applies the head activations inside its kernel (``losses.InstanceChannelsLoss``: tanh of the 'D' channel, ...) takes the model's
    RAW output.  The reference's ``model_call_func`` applies those activations itself in training (``apply_model_activations``,
    base_workflow.py:1403-1457), so the pair would apply them twice - silently.  Refused unless the function declares that it returns raw logits
    (attribute ``returns_raw_logits = True``) or is left to the default (``None``: to_pytorch_format -> model)."""
    fused = getattr(loss_function, "fused_head_activations", None)
    if model_call_func is None or not fused or getattr(model_call_func, "returns_raw_logits", False):
        return
    if any(str(a).lower() not in _GRAPH_SAFE_ACTS for a in fused):
        raise ValueError("the loss applies the head activations %s itself and needs the model's raw output, but a model_call_func was given (the reference's "
                         "applies them too): pass model_call_func=None, or mark yours with returns_raw_logits = True" % ([str(a) for a in fused],))


def _graphable_model(inner, criterion=None) -> bool:
    """A biapy_amd drop-in whose training-time ``model_call_func`` is the identity around the model (see the module docstring): every
    head is linear at training time, or the criterion applies the model's head activations itself (``InstanceChannelsLoss``: the
    'D' channel's tanh is part of the loss kernel) and was built with the same list."""
    if not getattr(inner, "_bpx_dropin", False):
        return False
    acts = [str(a).lower() for a in getattr(inner, "head_activations", ["ce_sigmoid"])]
    if all(a in _GRAPH_SAFE_ACTS for a in acts):
        return True
    fused = getattr(criterion, "fused_head_activations", None)
    return fused is not None and [str(a).lower() for a in fused] == acts


class _Window:
    """Device-side accumulation of the losses between two read-backs."""

    def __init__(self, n_losses: int, device):
        self.acc = torch.zeros(n_losses, dtype=torch.float64, device=device)
        self.pending = 0

    def add(self, losses: Sequence[torch.Tensor]):
        for i, l in enumerate(losses):
            self.acc[i] += l.detach().to(torch.float64)
        self.pending += 1

    def flush(self, logger: MetricLogger, loss_names: Sequence[str], log_writer=None):
        """The only host synchronisation of the loop: reads the sums, stops on a non-finite loss (train_engine.py:160-164)."""
        if not self.pending:
            return
        vals = self.acc.tolist()
        n = self.pending
        for name, v in zip(loss_names, vals):
            if not math.isfinite(v):
                print("Loss is {}, stopping training".format(v / n))
                sys.exit(1)
            logger.meters[name].update(v / n, n)
            if log_writer:
                log_writer.update(head="loss", **{name: v / n})
        self.acc.zero_()
        self.pending = 0


def train_one_epoch(
    cfg,
    model: torch.nn.Module,
    model_call_func: Optional[Callable],
    loss_function: Callable,
    metric_function: Optional[Callable],
    prepare_targets: Optional[Callable],
    data_loader,
    optimizer,
    device: torch.device,
    epoch: int,
    log_writer=None,
    lr_scheduler=None,
    verbose: bool = False,
    memory_bank=None,
    total_iters: int = 0,
    contrast_warmup_iters: int = 0,
    loss_names: Optional[List[str]] = None,
    *,
    graph: str = "auto",
    sync_every: int = 10,
) -> Tuple[Dict[str, float], int]:
    if memory_bank is not None:
        raise NotImplementedError("contrastive training (memory_bank) stays on the reference's train_one_epoch")
    device = torch.device(device)
    optimizers = _as_list(optimizer)
    loss_names = list(loss_names) if loss_names else ["loss"]
    schedulers = _as_list(lr_scheduler) or [None] * len(optimizers)
    lr_names = [n.replace("loss", "lr", 1) for n in loss_names]
    patch_size = tuple(_cfg_get(cfg, "DATA.PATCH_SIZE", ()) or ())
    clip = float(_cfg_get(cfg, "TRAIN.GRADIENT_CLIP_NORM", 0.0) or 0.0)
    sched_name = _cfg_get(cfg, "TRAIN.LR_SCHEDULER.NAME", "") or ""
    per_iter_warmup = sched_name in ("warmupcosine", "warmupreduceonplateau")
    inner = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    _check_fused_activations(model_call_func, loss_function)
    if model_call_func is None:
        def model_call_func(batch, is_train=True):  # noqa: E306 - the default of a stand-alone caller
            return model(to_pytorch_format(batch, device))
    if prepare_targets is None:
        def prepare_targets(targets, batch):  # noqa: E306
            return to_pytorch_format(targets, device)

    per_step_sched = sched_name == "onecycle" and any(isinstance(s, OneCycleLR) for s in schedulers if s is not None)
    capturable = all(g.get("capturable", False) for o in optimizers for g in o.param_groups)
    single = len(optimizers) == 1 and len(loss_names) == 1
    can_graph = (device.type == "cuda" and capturable and single and clip <= 0 and not per_step_sched and not per_iter_warmup
                 and _graphable_model(inner, loss_function))
    if graph == "on" and not can_graph:
        raise ValueError("graph='on' needs a CUDA/HIP device, a biapy_amd model with training-time-linear heads (or a loss fusing them), ONE capturable "
                         "optimizer and loss, no gradient clipping and no per-step scheduler")
    use_graph = can_graph and graph in ("on", "auto")

    model.train(True)
    logger = MetricLogger(delimiter="  ", verbose=verbose)
    for name in loss_names:
        logger.add_meter(name, SmoothedValue())
    for opt in optimizers:
        opt.zero_grad()
    win = _Window(len(loss_names), device)
    lr_sum = [0.0] * len(optimizers)
    lr_cnt = 0
    gstep, gshape = None, None
    step = -1
    n_steps = len(data_loader) if hasattr(data_loader, "__len__") else 0
    for step, (batch, targets) in enumerate(data_loader):
        if per_iter_warmup:                                            # per-iteration schedules (train_engine.py:118-121)
            for sched, opt in zip(schedulers, optimizers):
                if sched is not None and hasattr(sched, "adjust_learning_rate"):
                    sched.adjust_learning_rate(opt, step / max(n_steps, 1) + epoch)
        targets = prepare_targets(targets, batch)
        if patch_size and tuple(batch.shape[1:-1]) != tuple(patch_size[:-1]):
            raise ValueError(
                "Trying to input data with different shape than 'DATA.PATCH_SIZE'. Check your configuration."
                f" Input: {batch.shape[1:-1]} vs PATCH_SIZE: {patch_size[:-1]}"
            )
        outputs = None
        if use_graph:
            x = to_pytorch_format(batch, device)
            if gstep is None:
                gstep, gshape = _graph_step(inner, model, loss_function, optimizers[0], x, targets)
                if gstep is None:                                      # the loss function is not a plain tensor loss: eager epoch
                    use_graph = False
            if use_graph:
                if (tuple(x.shape), tuple(targets.shape)) == gshape:
                    losses = [gstep(x, targets)]
                    outputs = gstep.outputs
                else:                                                  # ragged last batch: same three phases, eagerly
                    loss, outputs = _eager_step(inner, loss_function, optimizers[0], x, targets)
                    losses = [loss]
                if metric_function is not None:
                    metric_function(outputs, targets, metric_logger=logger)
        if not use_graph:
            outputs = model_call_func(batch, is_train=True)
            losses, pre = _losses_of(loss_function(outputs, targets))
            if pre:
                for m_name, m_val in pre.items():
                    logger.update(**{m_name: m_val})
            elif metric_function is not None:
                metric_function(outputs, targets, metric_logger=logger)
            for i, loss_tensor in enumerate(losses):
                loss_tensor.backward()
                if clip > 0:
                    clip_grad_norm_([p for g in optimizers[i].param_groups for p in g["params"]], max_norm=clip)
                optimizers[i].step()
                if schedulers[i] is not None and isinstance(schedulers[i], OneCycleLR) and sched_name == "onecycle":
                    schedulers[i].step()
                optimizers[i].zero_grad()
        win.add(losses)
        if per_step_sched or per_iter_warmup:                          # the learning rate moves inside the epoch: sample it per step
            for i, opt in enumerate(optimizers):
                lr_sum[i] += max(float(g["lr"]) for g in opt.param_groups)
            lr_cnt += 1
        if win.pending >= sync_every:
            win.flush(logger, loss_names, log_writer)
            if verbose:
                print("Epoch: [{}]  [{}/{}]  {}".format(epoch + 1, step + 1, n_steps, str(logger)))
    win.flush(logger, loss_names, log_writer)
    steps_done = step + 1
    for i, opt in enumerate(optimizers):                               # train_engine.py:193-202: a meter of the max lr per step
        if steps_done == 0:
            break
        name = lr_names[i] if i < len(lr_names) else f"lr_{i}"
        m = SmoothedValue(window_size=1, fmt="{value:.6f}")
        if lr_cnt:
            m.total, m.count, m.last = lr_sum[i], lr_cnt, lr_sum[i] / lr_cnt
        else:
            lr = max(float(g["lr"]) for g in opt.param_groups)         # constant inside the epoch: one read (a device scalar under graphs)
            m.total, m.count, m.last = lr * steps_done, steps_done, lr
        logger.add_meter(name, m)
        if log_writer:
            log_writer.update(head="opt", **{name: m.last})
    logger.synchronize_between_processes(device)
    print("[Train] averaged stats:", logger)
    return {k: meter.global_avg for k, meter in logger.meters.items()}, step


def _graph_step(inner, model, loss_function, optimizer, x, t):
    """The captured step of (model, loss, optimizer, shapes), built once and cached on the model across epochs."""
    from . import graphs

    key = (id(optimizer), id(loss_function), tuple(x.shape), tuple(t.shape), _world())
    cached = getattr(inner, "_bpx_graph_step", None)
    if cached is not None and cached[0] == key:                        # later epochs replay the graphs captured in the first one
        return cached[1], (tuple(x.shape), tuple(t.shape))
    with torch.no_grad():
        probe = loss_function(inner(x), t)                             # a dict-returning loss cannot be captured as ONE backward
    if isinstance(probe, dict):
        return None, None
    del probe
    if not optimizer.state and not _fresh_state_is_zero(optimizer):
        return None, None                                              # its fresh state is not all-zero: the warm-up could not be undone
    multi = _world() > 1
    if multi and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
        graphs.broadcast_parameters_from_rank0(inner.parameters())     # a DDP wrap has done this already
    snap = _snapshot(inner, optimizer)                                 # capture warms up with real optimizer steps: undo them
    if multi:
        gstep = graphs.DataParallelTrainStep(inner, loss_function, optimizer, x, t, broadcast_parameters=False)
    else:
        gstep = graphs.GraphedTrainStep(inner, loss_function, optimizer, x, t)
    _restore(inner, optimizer, snap)
    inner._bpx_graph_step = (key, gstep)
    return gstep, (tuple(x.shape), tuple(t.shape))


# optimizers whose freshly created per-parameter state is all zeros (step counters, moments, accumulators): what _restore relies on
# for state that did not exist before the capture warm-up.  ASGD (eta = lr, mu = 1), Rprop (step_size = lr), Adagrad
# (initial_accumulator_value) and NAdam (mu_product starts at 1 and is only ever multiplied: zeroed it stays 0 and every bias correction of
# the run is wrong - ADVICE r3) do not qualify and keep the eager step when they arrive without state; neither does SGD with momentum and a
# non-zero dampening (torch seeds the momentum buffer with the first gradient, a zeroed buffer gives (1 - dampening) * grad).
_ZERO_INIT_OPTIMIZERS = (torch.optim.Adam, torch.optim.AdamW, torch.optim.RAdam, torch.optim.Adamax, torch.optim.Adadelta, torch.optim.RMSprop, torch.optim.SGD)


def _fresh_state_is_zero(optimizer) -> bool:
    if not isinstance(optimizer, _ZERO_INIT_OPTIMIZERS):
        return False
    if isinstance(optimizer, torch.optim.SGD):
        return all(not (g.get("momentum", 0) and g.get("dampening", 0)) for g in optimizer.param_groups)
    return True


def _snapshot(model, optimizer):
    params = [p.detach().clone() for p in model.parameters()]
    buffers = [b.detach().clone() for b in model.buffers()]
    state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in optimizer.state.items()}
    return params, buffers, state


@torch.no_grad()
def _restore(model, optimizer, snap) -> None:
    """Puts parameters, module buffers and optimizer state back IN PLACE (the captured graphs hold their addresses).  State that did
    not exist before the warm-up (a fresh optimizer) is zeroed - its initial value for the classes of ``_ZERO_INIT_OPTIMIZERS``,
    the only ones ``_graph_step`` lets through without state.  The loss function must be pure: a loss that keeps running state
    (moving averages, counters) has seen the warm-up batches and is not rolled back."""
    from .engine import bump_weights_epoch

    params, buffers, state = snap
    for p, s in zip(model.parameters(), params):
        p.copy_(s)
    for b, s in zip(model.buffers(), buffers):
        b.copy_(s)
    for p, st in optimizer.state.items():
        old = state.get(id(p))
        for k, v in st.items():
            if torch.is_tensor(v):
                if old is not None and torch.is_tensor(old.get(k)):
                    v.copy_(old[k])
                else:
                    v.zero_()
    bump_weights_epoch()
    torch.cuda.synchronize()


def _eager_step(model, loss_function, optimizer, x, t):
    """One eager step for a batch the captured graphs do not fit (ragged last batch); gradients averaged over the ranks.
    Replays do not depend on ``p.grad`` (the graphs hold raw addresses), so dropping the gradients here is safe."""
    optimizer.zero_grad(set_to_none=True)                              # p.grad may still alias a graph's private gradient buffers
    outputs = model(x)
    loss = loss_function(outputs, t)
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
    optimizer.zero_grad(set_to_none=True)
    return loss, outputs.detach()


@torch.no_grad()
def evaluate(
    cfg,
    model: torch.nn.Module,
    model_call_func: Optional[Callable],
    loss_function: Callable,
    metric_function: Optional[Callable],
    prepare_targets: Optional[Callable],
    epoch: int,
    data_loader,
    lr_scheduler=None,
    memory_bank=None,
    loss_names: Optional[List[str]] = None,
    *,
    device=None,
    eval_dtype: Optional[torch.dtype] = None,
) -> Dict[str, float]:
    """``eval_dtype``: storage type of the drop-in model for the duration of the pass (restored afterwards), e.g. ``torch.float16`` - the
    inference mode whose outputs agree with the fp32 reference to Dice 1e-4 - for a model that trains in bfloat16.  None: unchanged.

    Validation pass (train_engine.py:210-330): eval mode, the losses and whatever ``metric_function`` records averaged over
    the loader and over the ranks; steps ``ReduceLROnPlateau`` schedulers with their loss (:323-328).  ``model_call_func`` is
    called with ``is_train=True`` exactly as the reference does (:276).  The losses are accumulated on the device and read back
    once, at the end of the pass."""
    if memory_bank is not None:
        raise NotImplementedError("contrastive validation (memory_bank) stays on the reference's evaluate")
    loss_names = list(loss_names) if loss_names else ["loss"]
    schedulers = _as_list(lr_scheduler)
    sched_name = _cfg_get(cfg, "TRAIN.LR_SCHEDULER.NAME", "") or ""
    if device is None:
        p0 = next(iter(model.parameters()), None)
        device = p0.device if p0 is not None else torch.device("cpu")
    device = torch.device(device)
    _check_fused_activations(model_call_func, loss_function)
    if model_call_func is None:
        def model_call_func(batch, is_train=True):  # noqa: E306
            return model(to_pytorch_format(batch, device))
    if prepare_targets is None:
        def prepare_targets(targets, batch):  # noqa: E306
            return to_pytorch_format(targets, device)
    logger = MetricLogger(delimiter="  ")
    for name in loss_names:
        logger.add_meter(name, SmoothedValue())
    model.eval()
    inner = getattr(model, "module", model)                           # DistributedDataParallel keeps the drop-in under .module
    keep_dtype = getattr(inner, "compute_dtype", None)
    switch = eval_dtype is not None and keep_dtype is not None and keep_dtype != eval_dtype
    if switch:
        from .engine import set_compute_dtype
        set_compute_dtype(inner, eval_dtype)
    try:
        return _evaluate_pass(cfg, model, model_call_func, loss_function, metric_function, prepare_targets, epoch, data_loader, schedulers, sched_name,
                              loss_names, logger, device)
    finally:
        if switch:
            inner.compute_dtype = keep_dtype


def _evaluate_pass(cfg, model, model_call_func, loss_function, metric_function, prepare_targets, epoch, data_loader, schedulers, sched_name, loss_names,
                   logger, device):
    win = _Window(len(loss_names), device)
    for batch in data_loader:
        images, targets = batch[0], batch[1]
        targets = prepare_targets(targets, images)
        outputs = model_call_func(images, is_train=True)
        losses, pre = _losses_of(loss_function(outputs, targets))
        if pre:
            for m_name, m_val in pre.items():
                logger.update(**{m_name: m_val})
        elif metric_function is not None:
            metric_function(outputs, targets, metric_logger=logger)
        win.add(losses)
    win.flush(logger, loss_names)                                      # one read-back for the whole pass (+ the finiteness check)
    logger.synchronize_between_processes(device)
    print("[Val] averaged stats:", logger)
    if schedulers and sched_name == "reduceonplateau":
        for i, sched in enumerate(schedulers):
            if sched is not None and isinstance(sched, ReduceLROnPlateau):
                sched.step(logger.meters[loss_names[i]].global_avg, epoch=epoch)
    return {k: meter.global_avg for k, meter in logger.meters.items()}
