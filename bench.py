#!/usr/bin/env python
"""bench.py - voxels/s of the 3D ResUNet hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3                 # train step (fwd+bwd+AdamW), cfg 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                      # data parallel over RCCL, weak scaling
    python bench.py --mode infer                                    # forward only (sigmoid head fused)
    python bench.py --breakdown                                     # per-kernel event timing table (not timed run)

Workload = BASELINE.json configs[1]: 3D ResUNet (feature maps 16-32-64-128-256, InstanceNorm, ELU), 128^3
1-channel patches, batch 4 per GPU, bf16 storage / fp32 accumulate, synthetic data, random-init weights.
One step = one pass over one batch.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

FM = [16, 32, 64, 128, 256]
FLOP_PER_VOXEL_FWD = 144832       # BASELINE.md section 3 (2*MAC of all Conv3d/ConvTranspose3d), per input voxel
MFMA_PEAK_BF16 = 2.5e15           # dense bf16 peak, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def synth_batch(B, P, device, seed):
    """x ~ N(0,1) (zero-mean/unit-var normalised input), target = smoothed-noise blobs (~50 % fg)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(B, P, P, P, 1, generator=g, device=device)              # (B,Z,Y,X,C) as the data loader hands it over
    x = x.permute(0, 4, 1, 2, 3)                                            # to_pytorch_format: channels_last_3d view
    n = torch.randn(B, 1, P, P, P, generator=g, device=device)
    t = (F.avg_pool3d(n, 9, stride=1, padding=4) > 0).to(torch.float32)
    return x, t


def wgrad_k(key):
    """Kernel size of a bpx_conv3d_wgrad profile key: the int that follows the second tensor argument."""
    pos = [i for i, k in enumerate(key) if isinstance(k, str)]
    return key[pos[1] + 1]


def conv_flops(name, key):
    """Algorithmic FLOPs of one conv launch from its profile key (dtype,N,D,H,W,'Cx',...)."""
    if name not in ("bpx_conv3d_fwd", "bpx_conv3d_dgrad", "bpx_conv3d_wgrad"):
        return 0
    ints = [k for k in key if isinstance(k, int)]
    cs = [int(k[1:]) for k in key if isinstance(k, str)]
    N, D, H, W = ints[1:5]
    vox = N * D * H * W
    if name == "bpx_conv3d_fwd":      # x, sc, y
        cin, csc, cout = cs[0], cs[1], cs[2]
        return 2 * vox * (27 * cin + csc) * cout
    if name == "bpx_conv3d_dgrad":    # dy, t, g
        return 2 * vox * 27 * cs[0] * cs[2]
    if name == "bpx_conv3d_wgrad":    # x, act, dy, k[, small workspace size]
        k = wgrad_k(key)
        return 2 * vox * (k ** 3) * cs[0] * cs[1]
    return 0


def conv_bytes(name, key, es):
    """Algorithmic HBM bytes of one conv launch: every activation operand read once, the result written once."""
    if name not in ("bpx_conv3d_fwd", "bpx_conv3d_dgrad", "bpx_conv3d_wgrad"):
        return 0
    ints = [k for k in key if isinstance(k, int)]
    cs = [int(k[1:]) for k in key if isinstance(k, str)]
    N, D, H, W = ints[1:5]
    vox = N * D * H * W
    if name == "bpx_conv3d_fwd":      # read x, read shortcut input, write y
        return vox * (cs[0] + cs[1] + cs[2]) * es
    if name == "bpx_conv3d_dgrad":    # read dy, read the pre-activation t, write g
        return vox * (cs[0] + 2 * cs[2]) * es
    return vox * (cs[0] + cs[1]) * es  # wgrad: read x and dy


def cpu_baseline(P, train):
    """The oracle (plain PyTorch CPU fp32 restatement of the reference graph) on this host's cores, batch 1."""
    from oracle import net_oracle

    torch.manual_seed(0)
    sd = net_oracle.init_state_dict(1, FM, seed=0)
    x = torch.randn(1, 1, P, P, P)
    tgt = (torch.rand(1, 1, P, P, P) > 0.5).float()
    params = {k: v.clone().requires_grad_(train) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3) if train else None

    def step():
        if train:
            opt.zero_grad(set_to_none=True)
            loss = net_oracle.bce_with_logits(net_oracle.resunet_forward(params, x, FM), tgt)
            loss.backward()
            opt.step()
        else:
            with torch.no_grad():
                net_oracle.resunet_forward(params, x, FM)

    t0 = time.time(); step(); warm = time.time() - t0
    reps = 1 if warm > 12 else 2
    t0 = time.time()
    for _ in range(reps):
        step()
    dt = (time.time() - t0) / reps
    return dict(value=P ** 3 / dt, unit="voxels/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{reps} {'train steps (fwd+BCE+bwd+AdamW)' if train else 'forwards'} of one {P}^3 patch, batch 1, fp32, after 1 warm-up "
                       f"({dt:.2f} s each)")


def synth_volume(V, dev, seed=3):
    """SURVEY.md 8(d) cfg 3: closed-form sin/cos lattice + seeded noise, generated on the device (no 4.3 GB transfer)."""
    ax = torch.arange(V, device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(seed)
    vol = torch.sin(ax * 0.11)[:, None, None] * torch.cos(ax * 0.07)[None, :, None] + torch.sin(ax * 0.05)[None, None, :]
    vol = vol + 0.3 * torch.randn(V, V, V, generator=g, device=dev)
    return vol.unsqueeze(-1).contiguous()


def bench_sliding(a, model, dev, rank, world):
    """cfg 3: crop -> forward -> blend of one volume; patches sharded over the ranks (strong scaling)."""
    from biapy_amd.workflow import SlidingWindowPredictor

    model.eval()
    V = a.vol
    vol = synth_volume(V, dev)
    sw = SlidingWindowPredictor(model, (a.patch,) * 3, (0.5, 0.5, 0.5), (0, 0, 0), batch_size=a.batch)
    out = None
    for _ in range(a.warmup):
        out = sw.predict(vol, rank=rank, world=world, gather="rank0")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = sw.predict(vol, rank=rank, world=world, gather="rank0")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    if rank == 0:
        from biapy_amd import tiling
        plan = tiling.MergePlan((V, V, V), (a.patch,) * 3, (0.5, 0.5, 0.5), (0, 0, 0), dev)
        pv = plan.n_patches * a.patch ** 3
        print(json.dumps(dict(
            metric="voxels/sec 3D ResUNet 128^3 patch (sliding-window inference: crop+forward+blend, patch voxels)",
            value=pv * a.steps / elapsed, unit="voxels/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps,
            higher_is_better=True, scaling="strong", vs_baseline=None, dtype=a.dtype, data="synthetic",
            config=dict(workload="cfg3: %d^3 volume, %d^3 patches, 50%% overlap, %d patches sharded over %d GPU(s)" % (V, a.patch, plan.n_patches, world),
                        patches=plan.n_patches, volume=V, parallelism="z-slab x%d" % world),
            output_voxels_per_s=V ** 3 * a.steps / elapsed, checksum=float(out.double().mean().item()) if out is not None else None)))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["train", "infer", "sliding"], default="train")
    ap.add_argument("--vol", type=int, default=1024, help="sliding mode: edge of the synthetic volume (cfg 3 = 1024)")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel timing table of one step and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="testing aid: take the multi-GPU code path (RCCL process group) even with one rank")
    ap.add_argument("--dp", choices=["flat", "ddp"], default="flat",
                    help="multi-GPU gradient exchange: flat = two HIP-graph replays around ONE flat-gradient all-reduce "
                         "(biapy_amd.graphs.DataParallelTrainStep); ddp = torch DistributedDataParallel with eager hooks")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from a captured HIP graph (auto: single-GPU runs; the ~230 launches of a step are host-bound otherwise)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if os.environ.get("BPX_BENCH_ONE_DEVICE") == "1":   # testing aid: all ranks on cuda:0 (a 1-GPU box can then run the N > 1 launch line)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or a.force_ddp
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # "nccl" IS RCCL on ROCm; BPX_BENCH_BACKEND=gloo is a testing aid (gloo moves device tensors too, so two ranks can share a GPU)
        dist.init_process_group(os.environ.get("BPX_BENCH_BACKEND", "nccl"), init_method="env://")

    from biapy_amd import _lib as L
    from biapy_amd.resunet import ResUNet

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = ResUNet(image_shape=(a.patch,) * 3 + (1,), activation="elu", feature_maps=FM, drop_values=[0.0] * 5, normalization="in",
                    yx_down=[2] * 4, z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5, compute_dtype=dtype).to(dev)
    if a.mode == "sliding":
        return bench_sliding(a, model, dev, rank, world)
    train = a.mode == "train"
    net = model
    x, tgt = synth_batch(a.batch, a.patch, dev, seed=rank)
    if train:
        model.train()
        use_ddp = multi and (a.dp == "ddp" or a.graph == "off" or a.breakdown)
        if use_ddp:
            if a.graph != "off":
                # forward and backward as two HIP-graph replays below the autograd boundary; DDP's hooks, the RCCL all-reduce
                # and the optimizer stay eager (ResUNet.capture_graphs)
                try:
                    model.capture_graphs(x)
                except Exception as e:
                    print(f"[bench] rank {rank}: graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                    model.release_graphs()
            # one 27 MB bucket (a single ring all-reduce over xGMI) whose views ARE the .grad tensors: no copy-back kernels
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False,
                                                            gradient_as_bucket_view=True, bucket_cap_mb=int(os.environ.get("BPX_DDP_BUCKET_MB", "64")))
        want_graph = (a.graph == "on" or (a.graph == "auto" and not a.breakdown)) and not use_ddp
        try:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True, capturable=want_graph)
        except Exception:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=want_graph)
    else:
        model.eval()
        want_graph = a.graph == "on" or (a.graph == "auto" and not a.breakdown)   # no collective inside an inference step
    from biapy_amd.losses import BCEWithLogitsLoss
    loss_fn = BCEWithLogitsLoss()                                      # LOSS.TYPE="CE" -> BCEWithLogits (metrics.py:543-544), fused HIP passes

    def step():
        if train:
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(net(x), tgt)
            loss.backward()
            opt.step()
            return loss
        return model.predict_proba(x)

    eager_step = step
    graphed = False
    if want_graph:
        # HIP-graph capture of the whole step (forward, loss, backward, AdamW): one launch per step instead of ~170
        # (biapy_amd/graphs.py).
        try:
            from biapy_amd.graphs import GraphedInference, GraphedTrainStep

            if train and multi:
                from biapy_amd.graphs import DataParallelTrainStep

                gstep = DataParallelTrainStep(net, loss_fn, opt, x, tgt)
            elif train:
                gstep = GraphedTrainStep(net, loss_fn, opt, x, tgt)
            else:
                gstep = GraphedInference(model.predict_proba, x)

            def step():  # noqa: F811
                return gstep()

            graphed = True
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            step = eager_step
            torch.cuda.synchronize()
            if train and multi:                                        # still exchange gradients: same three phases, eager
                from biapy_amd.graphs import DataParallelTrainStep

                estep = DataParallelTrainStep(net, loss_fn, opt, x, tgt, graph=False, broadcast_parameters=False)
                step = lambda: estep()  # noqa: E731

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if train and not torch.isfinite(out.detach()).all():
        raise SystemExit("non-finite loss in warm-up")

    if a.breakdown:
        prof = L.Profile()
        L.lib.prof = prof
        step()
        torch.cuda.synchronize()
        L.lib.prof = None
        rows = sorted(prof.summary().items(), key=lambda kv: -kv[1][1])
        tot = sum(v[1] for _, v in rows)
        print(f"# per-call event timing of one {a.mode} step (B={a.batch}, {a.patch}^3, {a.dtype}); sum = {tot:.3f} ms")
        for (name, key), (cnt, ms) in rows:
            fl = conv_flops(name, key) * cnt
            tf = f"{fl / (ms * 1e-3) / 1e12:8.1f} TF/s" if fl else " " * 13
            print(f"{ms:9.3f} ms {100 * ms / tot:5.1f}% x{cnt:<3d} {tf} {name} {key}")
        return

    prof = L.Profile(names=("bpx_conv3d_fwd", "bpx_conv3d_dgrad", "bpx_conv3d_wgrad", "bpx_wgrad_defer_flush"))
    if not graphed:
        L.lib.prof = prof
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.lib.prof = None
    prof_steps = a.steps
    fb_graphs = train and getattr(model, "_graphs", None) is not None
    if fb_graphs:
        model.release_graphs()
    if graphed or fb_graphs:
        # a graph replay has no per-launch events: time the same launches on the same stream in eager steps right after
        # the timed region (same kernels, same shapes; `value` above is NOT taken from these steps)
        prof_steps = min(a.steps, 5)
        L.lib.prof = prof
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize()
        L.lib.prof = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    vox_step = world * a.batch * a.patch ** 3
    value = vox_step * a.steps / elapsed
    if rank == 0:
        # dominant kernel = the conv entry point with the largest summed event time inside the timed region
        summ = prof.summary()
        per = {}
        for (name, key), (cnt, ms) in summ.items():
            if name == "bpx_wgrad_defer_flush":      # the batched reduction of the step's partial slabs is part of the wgrad calls' time
                per.setdefault("bpx_conv3d_wgrad", [0.0, 0.0, 0, 0.0])[1] += ms
                continue
            d = per.setdefault(name, [0.0, 0.0, 0, 0.0])
            d[0] += conv_flops(name, key) * cnt
            d[1] += ms
            d[2] += cnt
            d[3] += conv_bytes(name, key, 2 if a.dtype == "bf16" else 4) * cnt
        dom = max(per.items(), key=lambda kv: kv[1][1]) if per else None
        roofline = None
        if dom:
            name, (fl, ms, cnt, by) = dom
            peak = MFMA_PEAK_BF16 / 1e12 if a.dtype == "bf16" else 157.3
            ach_f = fl / (ms * 1e-3) / 1e12                 # TFLOP/s
            ach_b = by / (ms * 1e-3) / 1e9                  # GB/s of algorithmic bytes
            # the binding roof is the one whose minimum time (work / peak) is larger for this kernel's launches
            hbm_bound = by / HBM_PEAK > fl / (peak * 1e12)
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, scripts/pmc_traffic.py
            try:
                traffic = (json.load(open(tf)).get(name) or {}).get("total_bytes")
            except (OSError, ValueError):
                traffic = None
            roofline = dict(bound="hbm" if hbm_bound else "mfma", kernel=name,
                            achieved=round(ach_b if hbm_bound else ach_f, 2), peak=HBM_PEAK / 1e9 if hbm_bound else peak,
                            unit="GB/s" if hbm_bound else "TFLOP/s",
                            frac=round(ach_b / (HBM_PEAK / 1e9) if hbm_bound else ach_f / peak, 4),
                            traffic=traffic, algorithmic_bytes_per_launch=round(by / cnt), flops_per_launch=round(fl / cnt),
                            launches=cnt, avg_launch_ms=round(ms / cnt, 4), tflops=round(ach_f, 2), algorithmic_GBps=round(ach_b, 1),
                            all={k: dict(tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 2), GBps=round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                                         ms_per_step=round(v[1] / prof_steps, 3)) for k, v in per.items()},
                            timed_on="eager steps right after the timed region (the timed region replays HIP graphs)" if (graphed or fb_graphs) else "the timed region")
        mult = 3 if train else 1
        line = dict(
            metric="voxels/sec 3D ResUNet 128^3 patch (%s)" % ("train: fwd+bwd+AdamW" if train else "inference forward"),
            value=value, unit="voxels/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.dtype, data="synthetic",
            config=dict(workload="cfg2: 3D ResUNet fm=16-32-64-128-256 IN+ELU, %d^3x1 patches, batch %d/GPU, %s" % (a.patch, a.batch, a.mode),
                        global_batch=world * a.batch, patch=a.patch, parallelism="dp%d" % world, mode=a.mode),
            launch=("hip-graph replays (forward+loss+backward | optimizer) around one flat-gradient RCCL all-reduce" if multi else
                    "hip-graph replay (whole step)") if graphed else ("hip-graph replay (forward, backward) + eager DDP/optimizer"
                                                                    if fb_graphs else "eager"),
            mfma_frac_end_to_end=round(value * FLOP_PER_VOXEL_FWD * mult / (world * MFMA_PEAK_BF16), 5),
            roofline=roofline,
        )
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.patch, train)
        print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
