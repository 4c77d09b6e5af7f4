"""HIP-graph replay of a whole training / inference step.

A 3D ResUNet step is ~170 kernel launches of 10-700 us each; issued eagerly from Python they leave the GPU idle between
dependent launches and, on a slow host, make the step host-bound.  Capturing the step once (``torch.cuda.graphs`` records the
engine's raw HIP launches like any other stream work) and replaying it costs one launch per step.

The reference has no equivalent (its step is ``base_workflow.py:1068-1137`` + ``train_engine.py:178-231`` run eagerly); this is
an addition of the MI355X path, used by ``bench.py`` and offered to callers who train on fixed-size patches - which is what
BiaPy does (``DATA.PATCH_SIZE`` is fixed per run, ``TRAIN.BATCH_SIZE`` with ``drop_last``).

Constraints (those of ``torch.cuda.graphs``): static shapes; the optimizer must be built with ``capturable=True``; no host
synchronisation inside the step; single process (``DistributedDataParallel`` all-reduces are left eager - use the plain step).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def _warm(fn, iters: int = 3) -> None:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(iters):
            fn()
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

        def eager():
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn(model(self.x), self.target)
            loss.backward()
            optimizer.step()
            return loss

        self.eager = eager
        _warm(eager, warmup)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = eager()
        torch.cuda.synchronize()

    def __call__(self, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
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
