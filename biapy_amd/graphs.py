"""HIP-graph replay of a whole training / inference step.

A 3D ResUNet step is ~170 kernel launches of 10-700 us each; issued eagerly from Python they leave the GPU idle between
dependent launches and, on a slow host, make the step host-bound.  Capturing the step once (``torch.cuda.graphs`` records the
engine's raw HIP launches like any other stream work) and replaying it costs one launch per step.

The reference has no equivalent (its step is ``base_workflow.py:1068-1137`` + ``train_engine.py:178-231`` run eagerly); this is
an addition of the MI355X path, used by ``bench.py`` and offered to callers who train on fixed-size patches - which is what
BiaPy does (``DATA.PATCH_SIZE`` is fixed per run, ``TRAIN.BATCH_SIZE`` with ``drop_last``).

Constraints (those of ``torch.cuda.graphs``): static shapes; the optimizer must be built with ``capturable=True``; no host
synchronisation inside the step; no autograd graph of an earlier eager backward may still be referenced when a graphed step is
built (``del loss`` first - the constructors raise a ``RuntimeError`` otherwise, see ``_warm``).  ``GraphedTrainStep`` is single-process; ``DataParallelTrainStep`` is the multi-GPU form
(one process per GPU): the same two replays with ONE flat-gradient RCCL all-reduce between them.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


_STALE_GRAPH = (
    "an autograd graph from an earlier backward is still alive (typically the last `loss` tensor of an eager loop): its "
    "AccumulateGrad nodes are bound to the stream they ran on, and capturing a backward that has to synchronise with that stream "
    "kills the process on ROCm.  Delete those references (`del loss`) before building a graphed step."
)


class _LrTensors:
    """A capturable Adam(W) given a Python-float ``lr`` bakes it (and ``1 - lr * weight_decay``) into the captured kernels, so a
    scheduler stepping between epochs would be ignored by the replays.  The learning rates therefore become 0-d device
    tensors before capture: PyTorch's schedulers update tensor learning rates in place (``lr_scheduler._update_param_group_val``),
    and a caller (or one of the reference's warm-up schedules) that ASSIGNS ``group["lr"] = value`` is caught by ``sync()``,
    which every replay calls: the value is copied into the captured tensor and the tensor is put back.
    ``weight_decay`` and the betas stay compile-time constants of the captured step."""

    def __init__(self, optimizer: torch.optim.Optimizer, device):
        self.opt = optimizer
        self.lrs = []
        for g in optimizer.param_groups:
            lr = g["lr"]
            if not torch.is_tensor(lr) or lr.device != torch.device(device) or lr.dtype != torch.float32 or lr.numel() != 1:
                lr = torch.tensor(float(lr), dtype=torch.float32, device=device)      # also a host / fp64 tensor lr: the captured kernels read a device float
                g["lr"] = lr
            self.lrs.append(lr)

    def sync(self) -> None:
        for g, t in zip(self.opt.param_groups, self.lrs):
            cur = g["lr"]
            if cur is not t:
                t.fill_(float(cur))                    # host scalar -> device tensor, no synchronisation
                g["lr"] = t


def _opt_step(optimizer) -> None:
    """``optimizer.step()``; Adam / AdamW through ``bpx_adam_step`` once their state exists (optim.py)."""
    from .optim import step

    step(optimizer)


def _bump() -> None:
    from .engine import bump_weights_epoch

    bump_weights_epoch()


def _warm(fn, iters: int = 3, side=None) -> None:
    """Eager warm-up on the stream the capture will use.  PyTorch warns when a parameter's AccumulateGrad node belongs to
    another stream; followed by a capture that situation dumped core (measured), so it is turned into an error here."""
    import warnings

    side = side if side is not None else torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    always = torch.is_warn_always_enabled()
    torch.set_warn_always(True)                       # the warning is a warn-once one: without this only the first hazard of a process is seen
    try:
        with warnings.catch_warnings():
            warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
            try:
                with torch.cuda.stream(side):
                    for _ in range(iters):
                        fn()
            except UserWarning:
                torch.cuda.synchronize()
                raise RuntimeError(_STALE_GRAPH) from None
    finally:
        torch.set_warn_always(always)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


class GraphedTrainStep:
    """``loss = step(x, target)`` == ``opt.zero_grad(); loss = loss_fn(model(x), target); loss.backward(); opt.step()``.

    ``x`` / ``target`` are copied into static device buffers (pass ``None`` to reuse what is already there); the returned loss
    is a static tensor that the next call overwrites.
    """

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: torch.optim.Optimizer, x: torch.Tensor,
                 target: torch.Tensor, warmup: int = 3):
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("build the optimizer with capturable=True to capture its step")
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.x, self.target = x.clone(), target.clone()
        self._lr = _LrTensors(optimizer, x.device)
        self._out = None

        def eager():
            optimizer.zero_grad(set_to_none=True)
            self._out = model(self.x)
            loss = loss_fn(self._out, self.target)
            loss.backward()
            _opt_step(optimizer)
            return loss

        self.eager = eager
        _warm(eager, warmup)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = eager()
        torch.cuda.synchronize()
        _bump()

    @property
    def outputs(self) -> torch.Tensor:
        """The model output of the last replay (static memory of the graph: valid until the next call)."""
        return self._out.detach()

    def __call__(self, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self._lr.sync()
        self.graph.replay()
        _bump()                                        # parameters changed without a version bump: packed-weight caches are stale
        return self.loss


@torch.no_grad()
def broadcast_parameters_from_rank0(params, group=None) -> None:
    """What ``DistributedDataParallel`` does when it wraps a module (base_workflow.py:952-958): every rank starts from rank 0's
    parameters.  One packed broadcast."""
    params = list(params)
    pack = torch.cat([p.reshape(-1) for p in params])
    dist.broadcast(pack, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    off = 0
    for p in params:
        p.copy_(pack[off:off + p.numel()].view_as(p))
        off += p.numel()


class DataParallelTrainStep:
    """Data-parallel step, one process per GPU: ``[zero grads, forward, loss, backward]`` -> all-reduce -> ``[mean, optimizer]``.

    What ``DistributedDataParallel`` does for the reference (``base_workflow.py:952-958``: parameters broadcast from rank 0 at
    construction, gradients averaged over ranks every step), laid out for replay: every ``p.grad`` is a view into ONE flat fp32
    buffer (6.69 M elements = 26.8 MB for cfg 2), so a step is two HIP-graph replays with a single ring all-reduce over xGMI
    between them - no per-bucket hooks, no copies, ~3 host calls per step.
    InstanceNorm has no cross-rank statistics and there are no buffers to synchronise.

    ``overlap`` (round 4): with a drop-in ResUNet the step drives the engine directly (forward, loss, the loss's own small
    autograd graph for d loss / d logits, the engine's hand-written backward) and splits the backward where its LAST stretch begins - the
    backward of the first encoder block, ~0.8 ms of kernels at cfg 2.  Every other parameter gradient is final in the flat slab by then (the
    engine flushes its queued weight-gradient reductions at that point), so their all-reduce (all of the 26.8 MB but the first block's few
    kilobytes) is started there with ``async_op`` and runs on RCCL's stream beside the rest of the backward; the first block's segment follows.
    Three graph replays per step (forward + backward head | backward tail | optimizer) instead of two.  This is what DDP's 25 MB buckets
    firing during backward buy the reference (base_workflow.py:952-958).  ``overlap="auto"`` takes it when the model qualifies; models without
    an engine (or with dict outputs), and models carrying forward / backward hooks (the overlapped form does not go through ``model.forward``),
    keep the serial form, as does ``overlap=False``.  The eager overlapped form re-reads the module's training flag at every step; a captured
    graph (either form) freezes the mode - dropout on or off - it was captured in.

    ``graph=False`` runs the same three phases eagerly (any device / backend; this is what the gloo tests drive); that form
    needs ``p.grad`` to stay views of ``self.flat_grad`` and re-binds them at every call, so an ``optimizer.zero_grad()`` by the
    caller is harmless.  The replayed form does not depend on ``p.grad`` at all (the graphs hold raw addresses and
    ``self.flat_grad`` keeps the slab alive).
    """

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: torch.optim.Optimizer, x: torch.Tensor,
                 target: torch.Tensor, group=None, graph: bool = True, warmup: int = 3, broadcast_parameters: bool = True, overlap="auto"):
        self.model, self.loss_fn, self.opt, self.group = model, loss_fn, optimizer, group
        self.overlapped = False
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("model has no trainable parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("DataParallelTrainStep expects fp32 master parameters on one device")
        if graph:
            if not x.is_cuda:
                raise RuntimeError("graph=True needs CUDA/HIP tensors")
            for g in optimizer.param_groups:
                if not g.get("capturable", False):
                    raise ValueError("build the optimizer with capturable=True to capture its step")
        if self.world > 1 and broadcast_parameters:                      # DDP's construction-time broadcast from rank 0
            broadcast_parameters_from_rank0(self.params, group)
        self.x, self.target = x.clone(), target.clone()
        self._lr = _LrTensors(optimizer, dev) if graph else None
        self._out = None
        inv = 1.0 / self.world
        if overlap and self._overlap_ok(model, x):
            self._init_overlapped(model, loss_fn, optimizer, graph, warmup, inv)
            return
        if overlap is True:
            raise ValueError("overlap=True needs a drop-in ResUNet (biapy_amd.resunet.ResUNet without super-resolution / class heads) on a HIP device")
        self.flat_grad = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

        def fwd_bwd():
            self.flat_grad.zero_()
            self._out = model(self.x)
            loss = loss_fn(self._out, self.target)
            loss.backward()                                              # AccumulateGrad adds in place: grads stay in flat_grad
            return loss

        def update():
            if self.world > 1:
                self.flat_grad.mul_(inv)
            _opt_step(optimizer)

        self._fwd_bwd, self._update = fwd_bwd, update
        self.graphs = None
        self.adopted = False
        if graph:
            side = torch.cuda.Stream()                                   # warm-up and capture on ONE stream: the parameters'
            #                                                              AccumulateGrad nodes remember the stream they were made on
            # Zero-copy form: when autograd leaves the gradients as consecutive views of ONE allocation in parameter order
            # (the MI355X engine writes every parameter gradient into one slab), that slab IS the flat buffer - no zero-fill
            # and no per-parameter accumulate kernels (98 small launches, ~0.3 ms per step for cfg 2).
            def fwd_bwd_adopt():
                optimizer.zero_grad(set_to_none=True)
                self._out = model(self.x)
                loss = loss_fn(self._out, self.target)
                loss.backward()
                return loss

            def probe():
                fwd_bwd_adopt()

            _warm(probe, 1, side)
            self.adopted = self._adopted_flat() is not None
            if self.adopted:
                def eager():
                    fwd_bwd_adopt()
                    self.flat_grad = self._adopted_flat()
                    self._all_reduce()
                    update()
            else:
                off = 0
                for p in self.params:                                    # the probe replaced the views
                    p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
                    off += p.numel()

                def eager():
                    fwd_bwd()
                    self._all_reduce()
                    update()

            _warm(eager, warmup, side)
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=side, capture_error_mode="thread_local"):
                self.loss = fwd_bwd_adopt() if self.adopted else fwd_bwd()
            if self.adopted:
                self.flat_grad = self._adopted_flat()                    # the slab of the captured run: static across replays
                if self.flat_grad is None:
                    raise RuntimeError("gradient layout changed between warm-up and capture")
            with torch.cuda.graph(g2, pool=g1.pool(), stream=side, capture_error_mode="thread_local"):
                update()
            torch.cuda.synchronize()
            self._check_views()
            self.graphs = (g1, g2)
            _bump()

    # ---- overlapped form -------------------------------------------------------------------------------------------------------------
    def _overlap_ok(self, model, x) -> bool:
        inner = model.module if hasattr(model, "module") and not hasattr(model, "engine") else model
        if not (x.is_cuda and hasattr(inner, "engine") and hasattr(inner, "_named") and hasattr(inner, "_finish_outputs")):
            return False
        if getattr(inner, "sr_pre", 0) or getattr(inner, "return_class", False) or getattr(inner, "explicit_activations", False):
            return False
        # the overlapped form calls engine.forward / engine.backward itself: module forward hooks would be skipped, so a hooked model keeps the
        # serial form (which goes through model.forward) under overlap="auto"
        hooked = ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")
        for top in {id(model): model, id(inner): inner}.values():
            for m_ in top.modules():      # submodules too: a hook on a block would be skipped just the same
                if any(getattr(m_, h, None) for h in hooked):
                    return False
        names, params = inner._named()
        if [id(p) for p in params] != [id(p) for p in self.params]:
            return False
        from .engine import needs_lift
        if getattr(getattr(inner, "cfg", None), "true_feature_maps", None) is not None or any(needs_lift(p) for p in params):
            return False      # zero-padded widths / 2D or anisotropic kernels: the engine's flat gradient slab has the lifted shapes, not the parameters'
        first = [n.startswith("down_path.0.") for n in names]
        k = sum(first)
        return 0 < k < len(names) and all(first[:k]) and not any(first[k:])      # the first block's parameters are a prefix of the slab

    def _init_overlapped(self, model, loss_fn, optimizer, graph, warmup, inv):
        inner = model.module if hasattr(model, "module") and not hasattr(model, "engine") else model
        eng = inner.engine()
        names, params = inner._named()
        self._n_first = sum(p.numel() for n, p in zip(names, params) if n.startswith("down_path.0."))
        self.overlapped = True
        self._works = []
        state = {"capturing": None, "open": None}

        def reduce_rest():          # at the start of the backward's last stretch: everything but the first block's gradients is final
            cap = state["capturing"]
            if cap is not None:     # capture: close the first graph here and open the second - nothing is exchanged while capturing
                cap[0].capture_end()
                state["open"] = None
                cap[1].capture_begin(pool=cap[0].pool(), capture_error_mode="thread_local")
                state["open"] = cap[1]
                return
            if self.world > 1:
                self._works.append(dist.all_reduce(eng.last_flat_grad[self._n_first:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        def fwd_bwd():
            # inner.engine() again: it refreshes engine.drop_active from the module's CURRENT training flag (the eager form follows a later
            # model.eval() / model.train(); a captured graph freezes the mode it was captured in, as it freezes everything else)
            eng_ = inner.engine()
            if eng_ is not eng:
                raise RuntimeError("DataParallelTrainStep: the model's compute_dtype changed after the step was built")
            P = {n: p.detach() for n, p in zip(names, params)}
            logits, ctx = eng.forward(P, self.x.to(torch.float32), head_act=0, save=True)
            self._out = logits
            with torch.enable_grad():
                lg = logits.detach().requires_grad_(True)
                loss = loss_fn(lg, self.target)
                (dl,) = torch.autograd.grad(loss, lg)
            G = eng.backward(P, ctx, dl, on_last_block=reduce_rest)
            self.flat_grad = eng.last_flat_grad
            for n, p in zip(names, params):
                p.grad = G[n]
            return loss.detach()

        def finish_reduce():
            if self.world > 1:
                self._works.append(dist.all_reduce(self.flat_grad[:self._n_first], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                for w in self._works:
                    w.wait()
                self._works.clear()

        def update():
            if self.world > 1:
                self.flat_grad.mul_(inv)
            _opt_step(optimizer)

        def eager():
            loss = fwd_bwd()
            finish_reduce()
            update()
            return loss

        self._eager_overlapped = eager
        self._finish_reduce, self._update = finish_reduce, update
        self.graphs = None
        self.adopted = True
        if not graph:
            return
        side = torch.cuda.Stream()
        with torch.no_grad():
            _warm(eager, warmup, side)
            g1a, g1b, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                state["capturing"] = (g1a, g1b)
                try:
                    g1a.capture_begin(capture_error_mode="thread_local")
                    state["open"] = g1a
                    self.loss = fwd_bwd()                                   # reduce_rest() switches from g1a to g1b inside
                    if state["open"] is not g1b:
                        raise RuntimeError("DataParallelTrainStep: the engine's backward never reached its last block (on_last_block was not called)")
                    g1b.capture_end()
                    state["open"] = None
                except BaseException:
                    # leave the stream out of capture mode before the error travels on (a loss_fn error or an unsupported shape inside fwd_bwd
                    # would otherwise leave every later HIP call of the process failing with a capture error)
                    if state["open"] is not None:
                        try:
                            state["open"].capture_end()
                        except Exception:  # noqa: BLE001 - the capture is already invalid; ending it is best effort
                            pass
                        state["open"] = None
                    raise
                finally:
                    state["capturing"] = None
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(g2, pool=g1a.pool(), stream=side, capture_error_mode="thread_local"):
                update()
        torch.cuda.synchronize()
        self._check_views()
        self.graphs = (g1a, g1b, g2)
        _bump()

    def _adopted_flat(self) -> Optional[torch.Tensor]:
        """The gradients as one flat tensor if they are consecutive contiguous fp32 views of one allocation, in parameter order."""
        g0 = self.params[0].grad
        if g0 is None:
            return None
        store, base, off = g0.untyped_storage(), g0.data_ptr(), 0
        for p in self.params:
            g = p.grad
            if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != store.data_ptr()
                    or g.data_ptr() != base + 4 * off):
                return None
            off += p.numel()
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(store, g0.storage_offset(), (off,), (1,))

    @property
    def outputs(self) -> torch.Tensor:
        """The model output of the last step (static memory under graphs: valid until the next call)."""
        return self._out.detach()

    def _bind_views(self):
        """Make every ``p.grad`` a view of ``self.flat_grad`` again (the eager form accumulates into them)."""
        base, off = self.flat_grad.data_ptr(), 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    def _check_views(self):
        base = self.flat_grad.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                raise RuntimeError("a gradient left the flat buffer (was optimizer.zero_grad(set_to_none=True) called?)")
            off += p.numel()

    def _all_reduce(self):
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def __call__(self, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if self.overlapped:
            if self.graphs is None:
                with torch.no_grad():
                    loss = self._eager_overlapped()
                _bump()
                return loss
            self._lr.sync()
            self.graphs[0].replay()                                      # forward, loss, backward up to its last stretch
            if self.world > 1:                                           # ... whose gradients travel while the last stretch runs
                self._works.append(dist.all_reduce(self.flat_grad[self._n_first:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.graphs[1].replay()                                      # backward of the first encoder block
            self._finish_reduce()
            self.graphs[2].replay()
            _bump()
            return self.loss
        if self.graphs is None:
            self._bind_views()                                           # the caller may have dropped or replaced p.grad
            loss = self._fwd_bwd()
            self._all_reduce()
            self._update()
            return loss
        self._lr.sync()
        self.graphs[0].replay()
        self._all_reduce()
        self.graphs[1].replay()
        _bump()
        return self.loss


class GraphedInference:
    """``y = infer(x)`` == ``fn(x)`` (e.g. ``model.predict_proba``) for a fixed input shape; ``y`` is a static tensor."""

    def __init__(self, fn: Callable, x: torch.Tensor, warmup: int = 3):
        if not x.is_cuda:
            raise RuntimeError("GraphedInference needs CUDA/HIP tensors")
        self.fn = fn
        self.x = x.clone()
        with torch.no_grad():
            _warm(lambda: fn(self.x), warmup)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.y = fn(self.x)
        torch.cuda.synchronize()

    def __call__(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.y
