#!/usr/bin/env python
"""bench.py - voxels/s of the 3D ResUNet hot path on MI355X (BASELINE.json metric: "train+infer" at 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3                 # train step (fwd+bwd+AdamW) of cfg 2, then inference + cfg 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                      # data parallel over RCCL (weak scaling); sliding window sharded
    python bench.py --mode infer | --mode sliding                   # one section only (its own JSON line)
    python bench.py --breakdown                                     # per-kernel event timing table (not a timed run)

ONE JSON line on rank 0.  ``value`` = the TRAIN throughput (the timed region the driver's clock brackets: W warm-up steps, then
exactly K steps between barriers + synchronize, max over ranks).  The same process then measures, each bracketed the same way
and reported as a sub-record with its own roofline entry:
  ``infer``   - forward + fused sigmoid of the same batch (weak scaling: every rank its own batch);
  ``sliding`` - cfg 3, crop -> forward -> blend of a 1024^3 synthetic volume, 4096 patches sharded over the ranks by Z-slab, each
                rank holding only its input slab; strong scaling (N = 1 runs the whole 1024^3 volume on one GPU; --vol 512 is one GPU's
                share of the 8-GPU job).
Workload = BASELINE.json configs[1]: 3D ResUNet (feature maps 16-32-64-128-256, InstanceNorm, ELU), 128^3 1-channel patches,
batch 4 per GPU, bf16 storage / fp32 accumulate, synthetic data, random-init weights.  One step = one pass over one batch.
"""
import argparse
import json
import os
import sys
import threading
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
# `dtype` of the JSON line: the arithmetic types of the timed path (accumulation is fp32 everywhere)
DTYPE_NAMES = {"mix16": "f16 forward/activations + bf16 gradients (MFMA f16 / bf16, fp32 accumulate)", "bf16": "bf16", "f32": "f32"}
CONV_ENTRIES = ("bpx_conv3d_fwd", "bpx_conv3d_dgrad", "bpx_conv3d_wgrad", "bpx_conv3d_wgrad_db2", "bpx_conv3d_bwd_fused", "bpx_wgrad_defer_flush")
_CONV_NAMES = ("bpx_conv3d_fwd", "bpx_conv3d_dgrad", "bpx_conv3d_wgrad", "bpx_conv3d_bwd_fused")


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
    if name not in _CONV_NAMES:
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
    if name == "bpx_conv3d_bwd_fused":  # dy, t, g: the dgrad AND the wgrad of the conv (each 2 * 27 * Cdy * Ct per voxel)
        return 2 * 2 * vox * 27 * cs[0] * cs[2]
    k = wgrad_k(key)                  # wgrad: x, act, dy, k[, small workspace size]
    return 2 * vox * (k ** 3) * cs[0] * cs[1]


def conv_bytes(name, key, es):
    """Algorithmic HBM bytes of one conv launch: every activation operand read once, the result written once."""
    if name not in _CONV_NAMES:
        return 0
    ints = [k for k in key if isinstance(k, int)]
    cs = [int(k[1:]) for k in key if isinstance(k, str)]
    N, D, H, W = ints[1:5]
    vox = N * D * H * W
    if name == "bpx_conv3d_fwd":      # read x, read shortcut input, write y
        return vox * (cs[0] + cs[1] + cs[2]) * es
    if name in ("bpx_conv3d_dgrad", "bpx_conv3d_bwd_fused"):    # read dy, read the pre-activation t, write g (the fused backward reads nothing more for dW)
        return vox * (cs[0] + 2 * cs[2]) * es
    return vox * (cs[0] + cs[1]) * es  # wgrad: read x and dy


# ------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (PyTorch CPU fp32 restatement of the reference graph, NumPy restatement of crop / merge) on this host
# ------------------------------------------------------------------------------------------------------------------------
def _cpu_net_leg(P, train, threads, reps=3):
    from oracle import net_oracle

    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
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
        if warm > 30:                       # a very slow host: one timed repetition keeps the default run within minutes
            reps = 1
        times = []
        for _ in range(reps):               # >= 3 timed repetitions after the warm-up; mean, spread stated
            t0 = time.time()
            step()
            times.append(time.time() - t0)
        dt = sum(times) / len(times)
    finally:
        torch.set_num_threads(old)
    sd = (sum((t - dt) ** 2 for t in times) / len(times)) ** 0.5
    return dict(value=P ** 3 / dt, unit="voxels/s", cores=threads, kind="port", reps=len(times), seconds_per_rep=[round(t, 3) for t in times],
                rel_std=round(sd / dt, 4),
                sample=f"{len(times)} {'train steps (fwd+BCE+bwd+AdamW)' if train else 'forwards'} of one {P}^3 patch, batch 1, fp32, after 1 warm-up "
                       f"(mean {dt:.2f} s, min {min(times):.2f} s, max {max(times):.2f} s)")


def _cpu_tiling_leg(V=256, P=128):
    """crop + merge of a V^3 volume (256^3, 64 patches) with the NumPy oracle: single-threaded, as the reference is."""
    import numpy as np

    from oracle import tiling_oracle

    vol = np.random.RandomState(0).rand(V, V, V, 1).astype(np.float32)
    t0 = time.time()
    p, _ = tiling_oracle.crop(vol, (P, P, P, 1), (0.5, 0.5, 0.5))
    t1 = time.time()
    tiling_oracle.merge(p, vol.shape, overlap=(0.5, 0.5, 0.5))
    t2 = time.time()
    n = p.shape[0]
    return dict(value=n * P ** 3 / (t2 - t0), unit="patch voxels/s", cores=1, kind="port",
                sample=f"crop ({t1 - t0:.2f} s) + merge ({t2 - t1:.2f} s) of one {V}^3 volume, {n} patches of {P}^3, 50 % overlap, fp32",
                crop_s=round(t1 - t0, 3), merge_s=round(t2 - t1, 3), patches=n)


def cpu_thread_settings(ncpu):
    """BiaPy's default (min(4, ncpu), biapy/_biapy.py:333-346), 16, 32, 64 and every core of this host - clipped to the host, ascending, unique."""
    return sorted({min(t, ncpu) for t in (4, 16, 32, 64, ncpu)})


def cpu_baseline(P, quick=False):
    """the reference's CPU path beside the GPU numbers, on this host, core count stated.  ONE shape for every network leg (the
    benched P^3 patch, batch 1): the train step and the inference forward at BiaPy's default ``min(4, ncpu)`` threads, 16, 32, 64 and all cores
    (round 4 compared 128 threads at 128^3 with 4 threads at 64^3, and the 128-thread leg it quoted was the SLOWER one).
    Every leg is reported; the headline ``value`` is the FASTEST train leg and ``cores`` its thread count.  crop + merge single-threaded as the
    reference is.  Bounded: 1 warm-up + 3 repetitions per leg, about two minutes in total."""
    ncpu = torch.get_num_threads()
    settings = [ncpu] if quick else cpu_thread_settings(ncpu)
    legs = {}
    for t in settings:
        legs[f"train_{t}_threads"] = _cpu_net_leg(P, True, t, reps=1 if quick else 3)
    if not quick:
        for t in settings:
            legs[f"infer_{t}_threads"] = _cpu_net_leg(P, False, t, reps=3)
        legs["crop_merge_numpy"] = _cpu_tiling_leg()
    best = max((k for k in legs if k.startswith("train_")), key=lambda k: legs[k]["value"])
    head = dict(legs[best])
    head["headline_leg"] = best
    head["biapy_default_threads"] = min(4, ncpu)
    head["host_threads"] = ncpu
    head["legs"] = legs
    return head


# ------------------------------------------------------------------------------------------------------------------------
def synth_volume(V, dev, z0=0, z1=None, seed=3):
    """SURVEY.md 8(d) cfg 3: closed-form sin/cos lattice + seeded noise, generated on the device slice by slice (one generator
    seed per z slice), so that a rank can build exactly its input slab [z0, z1) of the V^3 volume."""
    z1 = V if z1 is None else z1
    ax = torch.arange(V, device=dev, dtype=torch.float32)
    az = torch.arange(z0, z1, device=dev, dtype=torch.float32)
    vol = torch.sin(az * 0.11)[:, None, None] * torch.cos(ax * 0.07)[None, :, None] + torch.sin(ax * 0.05)[None, None, :]
    g = torch.Generator(device=dev)
    for k in range(z1 - z0):
        g.manual_seed(seed * 100003 + z0 + k)
        vol[k] += 0.3 * torch.randn(V, V, generator=g, device=dev)
    return vol.unsqueeze(-1).contiguous()


def _timed(fn, steps, world, dev):
    """The contract's timed region: barrier + synchronize on both sides, MAX over ranks."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    return elapsed


def run_sliding(a, model, dev, rank, world, V, warmup, steps):
    """cfg 3: crop -> forward -> blend of one volume; patches sharded over the ranks (strong scaling), slab-only inputs."""
    from biapy_amd import _lib as L
    from biapy_amd import tiling
    from biapy_amd.workflow import SlidingWindowPredictor

    model.eval()
    sw = SlidingWindowPredictor(model, (a.patch,) * 3, (0.5, 0.5, 0.5), (0, 0, 0), batch_size=a.batch)
    z_lo, z_hi = sw.input_slab((V, V, V), rank, world)
    slab = synth_volume(V, dev, z_lo, z_hi) if z_hi > z_lo else torch.zeros((1, V, V, 1), device=dev)
    out = [None]

    def once():
        out[0] = sw.predict(slab, rank=rank, world=world, gather="rank0", z_offset=z_lo, full_z=V)

    for _ in range(warmup):
        once()
    elapsed = _timed(once, steps, world, dev)
    # per-kernel times of the blend and the gather of the last pass's geometry, on this rank (events on the launch stream)
    prof = L.Profile(names=("bpx_merge3d_blend", "bpx_crop3d_gather"))
    L.lib.prof = prof
    once()
    torch.cuda.synchronize()
    L.lib.prof = None
    ms = {"bpx_merge3d_blend": 0.0, "bpx_crop3d_gather": 0.0}
    cnt = {"bpx_merge3d_blend": 0, "bpx_crop3d_gather": 0}
    for (name, _), (c, t) in prof.summary().items():
        ms[name] += t
        cnt[name] += c
    from biapy_amd.workflow import plan_slabs

    plan = tiling.MergePlan((V, V, V), (a.patch,) * 3, (0.5, 0.5, 0.5), (0, 0, 0), dev)
    mine = plan_slabs([plan.row_start(i) for i in range(plan.grid[0].n)], a.patch, V, world)[rank]
    n_mine = (mine.rows[1] - mine.rows[0]) * plan.grid[1].n * plan.grid[2].n
    pv = plan.n_patches * a.patch ** 3
    # algorithmic bytes of this rank's blend: its predictions read once + the slices it produces written once (fp32, C = 1);
    # the boundary partial sums (numerator + weights, written and re-read) are overhead, not algorithmic bytes
    merge_bytes = n_mine * a.patch ** 3 * 4 + (mine.own[1] - mine.own[0]) * V * V * 4
    crop_bytes = 2 * n_mine * a.patch ** 3 * 4
    rec = None
    if rank == 0:
        mb = merge_bytes / (ms["bpx_merge3d_blend"] * 1e-3) / 1e9 if ms["bpx_merge3d_blend"] else None
        cb = crop_bytes / (ms["bpx_crop3d_gather"] * 1e-3) / 1e9 if ms["bpx_crop3d_gather"] else None
        rec = dict(
            metric="voxels/sec 3D ResUNet 128^3 patch (sliding-window inference: crop+forward+blend, patch voxels)",
            value=pv * steps / elapsed, unit="voxels/s", n_gpus=world, steps=steps, warmup=warmup, ms_per_step=1e3 * elapsed / steps,
            scaling="strong", output_voxels_per_s=V ** 3 * steps / elapsed,
            config=dict(workload="cfg3%s: %d^3 volume, %d^3 patches, 50%% overlap, %d patches sharded over %d GPU(s), slab-only inputs"
                                 % ("" if V == 1024 else " (one GPU's share of the 8-GPU job)", V, a.patch, plan.n_patches, world),
                        patches=plan.n_patches, volume=V, parallelism="z-slab x%d" % world, input_slices_rank0=[z_lo, z_hi]),
            mfma_frac_end_to_end=round(pv * steps / elapsed * FLOP_PER_VOXEL_FWD / (world * MFMA_PEAK_BF16), 5),
            roofline=dict(bound="hbm", kernel="bpx_merge3d_blend", achieved=round(mb, 1) if mb else None, peak=HBM_PEAK / 1e9, unit="GB/s",
                          frac=round(mb / (HBM_PEAK / 1e9), 4) if mb else None, **_tiling_traffic(V, world, plan.n_patches),
                          algorithmic_bytes_per_pass=merge_bytes, launches=cnt["bpx_merge3d_blend"], ms_per_pass=round(ms["bpx_merge3d_blend"], 3),
                          crop=dict(kernel="bpx_crop3d_gather", GBps=round(cb, 1) if cb else None, launches=cnt["bpx_crop3d_gather"],
                                    ms_per_pass=round(ms["bpx_crop3d_gather"], 3), algorithmic_bytes_per_pass=crop_bytes),
                          timed_on="one extra pass after the timed region, rank 0's share"),
            checksum=float(out[0].double().mean().item()) if out[0] is not None else None)
    return rec


def _tiling_traffic(V, world, n_patches):
    """HBM bytes per blend / gather pass from the committed PMC passes (profiles/pmc_traffic_tiling.json: 512 patches of 128^3 <-> 512^3,
    tests/bench_kernels.py merge_rows under rocprofv3 --pmc); only quoted for that shape - the default 1-GPU sub-record."""
    src = os.path.join("profiles", "pmc_traffic_tiling.json")
    if not (V == 512 and world == 1 and n_patches == 512):
        return dict(traffic=None)
    try:
        from biapy_amd._lib import source_digest

        d = json.load(open(os.path.join(ROOT, src)))
        stamp = (d.get("_meta") or {}).get("tiling_sha256")
        if stamp != source_digest(only=("tiling.hip", "bpx_common.h")):      # the same refusal as for the conv counters
            print(f"[bench] {src} is STALE: collected for other tiling kernel sources; sliding.traffic is null until scripts/refresh_profiles.sh is re-run", file=sys.stderr)
            return dict(traffic=None, traffic_error=f"{src} is stale (collected for other kernel sources); re-run scripts/refresh_profiles.sh")
        return dict(traffic=(d.get("bpx_merge3d_blend") or {}).get("total_bytes"), crop_traffic=(d.get("bpx_crop3d_gather") or {}).get("total_bytes"),
                    traffic_source=src + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tests/bench_kernels.py merge_rows, committed; "
                                         "bytes per whole-volume pass: the gather is issued per batch in the predictor)")
    except (OSError, ValueError):
        return dict(traffic=None)


def pmc_traffic(entry, src=os.path.join("profiles", "pmc_traffic.json")):
    """HBM bytes per call of a C-ABI entry from the committed rocprofv3 --pmc passes (scripts/pmc_traffic.py) - quoted only while the file's
    stamp (sha256 of biapy_amd/csrc) equals the tree's: counters of other kernels than the ones being timed are refused, loudly."""
    try:
        from biapy_amd._lib import source_digest

        d = json.load(open(os.path.join(ROOT, src)))
        stamp = (d.get("_meta") or {}).get("csrc_sha256")
        if stamp != source_digest():
            print(f"[bench] {src} is STALE: it was collected for other kernel sources (stamp {str(stamp)[:12]} != tree {source_digest()[:12]}); "
                  f"roofline.traffic is null until scripts/refresh_profiles.sh is re-run", file=sys.stderr)
            return dict(traffic=None, traffic_error=f"{src} is stale (collected for other kernel sources); re-run scripts/refresh_profiles.sh")
        return dict(traffic=(d.get(entry) or {}).get("total_bytes"),
                    traffic_source=src + " (separate rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes over this command, committed with the stamp "
                                         "of the kernel sources; per call of the entry point, not measured in this run)")
    except (OSError, ValueError) as e:
        return dict(traffic=None, traffic_error=f"{src}: {e}")


def _shape_name(name, key):
    """'bpx_conv3d_fwd[4x128^3 C48+0->C16]' from a profile key (dtype, N, D, H, W, 'Cx', ...)."""
    ints = [k for k in key if isinstance(k, int)]
    cs = [int(k[1:]) for k in key if isinstance(k, str)]
    N, D, H, W = ints[1:5]
    vol = f"{D}^3" if D == H == W else f"{D}x{H}x{W}"
    if name == "bpx_conv3d_fwd":
        return f"{name}[{N}x{vol} C{cs[0]}" + (f"+sc{cs[1]}" if cs[1] else "") + f"->C{cs[2]}]"
    if name in ("bpx_conv3d_dgrad", "bpx_conv3d_bwd_fused"):
        return f"{name}[{N}x{vol} dy C{cs[0]}->g C{cs[2]}]"
    return f"{name}[{N}x{vol} C{cs[0]}->C{cs[1]} k{wgrad_k(key)}]"


def conv_roofline(prof, prof_steps, dtype, timed_on):
    """Roofline entries from per-launch HIP events on the launch stream: `roofline` = the launch SHAPE (entry point + tensor shape = one
    kernel instance) with the most time per step, `top3` the three heaviest shapes, `families` the per-entry-point totals of round 2.
    Algorithmic bytes / flops per launch follow from the shape; the binding roof is the larger of bytes / 8 TB/s
    and flops / the dense MFMA peak."""
    summ = prof.summary()
    es = 4 if dtype == "f32" else 2
    peak = 157.3 if dtype == "f32" else MFMA_PEAK_BF16 / 1e12      # dense fp16 MFMA peak = dense bf16 peak
    fam, shapes = {}, []
    flush_ms = 0.0
    for (name, key), (cnt, ms) in summ.items():
        if name == "bpx_conv3d_wgrad_db2":       # the same entry with a second bias-gradient destination (same key layout)
            name = "bpx_conv3d_wgrad"
        if name == "bpx_wgrad_defer_flush":      # the batched reduction of the step's partial slabs is part of the wgrad calls' time
            flush_ms += ms
            continue
        fl, by = conv_flops(name, key) * cnt, conv_bytes(name, key, es) * cnt
        d = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
        d[0] += fl; d[1] += ms; d[2] += cnt; d[3] += by
        if cnt and fl:
            shapes.append((ms, name, key, cnt, fl, by))
    if "bpx_conv3d_wgrad" in fam:
        fam["bpx_conv3d_wgrad"][1] += flush_ms
    fam = {k: v for k, v in fam.items() if v[2] > 0}
    if not shapes:
        return None, None

    def entry(ms, name, key, cnt, fl, by):
        ach_f, ach_b = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        hbm = by / HBM_PEAK > fl / (peak * 1e12)
        return dict(bound="hbm" if hbm else "mfma", kernel=_shape_name(name, key), achieved=round(ach_b if hbm else ach_f, 2),
                    peak=HBM_PEAK / 1e9 if hbm else peak, unit="GB/s" if hbm else "TFLOP/s", frac=round(ach_b / (HBM_PEAK / 1e9) if hbm else ach_f / peak, 4),
                    algorithmic_bytes_per_launch=round(by / cnt), flops_per_launch=round(fl / cnt), launches=cnt, avg_launch_ms=round(ms / cnt, 4),
                    ms_per_step=round(ms / prof_steps, 3), tflops=round(ach_f, 2), algorithmic_GBps=round(ach_b, 1), mfma_frac=round(ach_f / peak, 4),
                    hbm_frac=round(ach_b / (HBM_PEAK / 1e9), 4))

    shapes.sort(key=lambda t: -t[0])
    top = [entry(*t) for t in shapes[:3]]
    head = dict(top[0])
    head.update(pmc_traffic(shapes[0][1]))
    head["traffic_is_for"] = "the average call of entry point %s (all its shapes), not this shape alone" % shapes[0][1]
    head["families"] = {k: dict(tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 2), GBps=round(v[3] / (v[1] * 1e-3) / 1e9, 1), ms_per_step=round(v[1] / prof_steps, 3),
                                hbm_frac=round(v[3] / (v[1] * 1e-3) / HBM_PEAK, 4)) for k, v in fam.items()}
    head["timed_on"] = timed_on
    return head, top


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["all", "train", "infer", "sliding"], default="all",
                    help="all = train (the headline value) + the infer and sliding sub-records; one name = that section only")
    ap.add_argument("--vol", type=int, default=None, help="edge of the synthetic sliding-window volume (default: cfg 3's 1024 at every N; 512 = one GPU's share of the "
                                                          "8-GPU job, the round 1-5 default at N = 1)")
    ap.add_argument("--feed", choices=["host", "device"], default="device",
                    help="device (default, as rounds 1-5; labelled in the record): one device-resident batch, H2D excluded; host: every timed train step takes "
                         "its batch from pinned host memory (async H2D on a copy stream + a device-to-device copy into the step's inputs), as the reference's "
                         "loop does.  Whichever is the headline, the other is timed right after it over the same K steps and reported as a sub-field")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--dtype", choices=["mix16", "bf16", "f32"], default="mix16",
                    help="mix16 (default, compute_dtype=torch.float16): fp16 forward pass and stored activations, bf16 gradients and backward MFMA "
                         "operands, fp32 accumulation - the training mode whose forward meets Dice delta < 1e-4 against the fp32 reference; "
                         "bf16: the all-bf16 mode of rounds 1-2 (A/B baseline); f32: the exact mode")
    ap.add_argument("--infer-dtype", choices=["f16", "bf16", "f32", "same"], default="f16",
                    help="storage type of the inference forward and of the sliding window (sub-records `infer`, `sliding`; --mode infer / sliding): "
                         "f16 = the fp16 forward that meets the Dice < 1e-4 bar at the speed of bf16 (the forward of the mixed training mode); "
                         "same = --dtype")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel timing table of one step and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="skip the eager per-launch event steps after a graph-replayed timed region (no roofline entry): a rocprofv3 kernel trace "
                         "of the command then holds the capture warm-up and graph-replayed steps only")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="only the all-core train leg")
    ap.add_argument("--force-ddp", action="store_true", help="testing aid: take the multi-GPU code path (RCCL process group) even with one rank")
    ap.add_argument("--dp", choices=["flat", "ddp"], default="flat",
                    help="multi-GPU gradient exchange: flat = two HIP-graph replays around ONE flat-gradient all-reduce "
                         "(biapy_amd.graphs.DataParallelTrainStep); ddp = torch DistributedDataParallel with eager hooks")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from a captured HIP graph (auto: on; the ~230 launches of a step are host-bound otherwise)")
    ap.add_argument("--arch", choices=["resunet", "resunetpp"], default="resunet",
                    help="resunetpp = cfg 4 (3D instance segmentation, B/C/D channels, ResUNet++ fm 16-32-64-128-256, 80^3 patches): its own JSON "
                         "line, train mode only - a second-tier configuration, not the headline")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the ResUNet++ (cfg 4) sub-record of the single-GPU line")
    ap.add_argument("--no-bf16-record", action="store_true", help="skip the pure-bf16 train sub-record (`train_bf16`) of the single-GPU mix16 line")
    ap.add_argument("--self-check", action="store_true",
                    help="verify (also with one rank under --force-ddp) that every rank's post-all-reduce gradient / parameter checksum agrees and that "
                         "the gathered cfg-3 volume has the checksum of the single-GPU run committed in profiles/sliding_checksums.json; at N > 1 the "
                         "checks run by default")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the RCAN x4 super-resolution (cfg 5) sub-record of the single-GPU line")
    ap.add_argument("--sliding-timeout", type=float, default=240.0,
                    help="N > 1: seconds after which a hung sliding-window section is abandoned (the line is printed without it)")
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
        # Bring the communicator up NOW and drain C stdio on every rank: RCCL writes a version banner through stdio when its first communicator is
        # created, and a redirected stdout only flushes that buffer at exit - from a rank other than 0 possibly AFTER rank 0 has printed the JSON line.
        # Ranks other than 0 leave through os._exit (finish()), which drops whatever C stdio still holds.
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass

    from biapy_amd import _lib as L
    from biapy_amd.resunet import ResUNet

    dtype = {"mix16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    if a.arch == "resunetpp":
        return run_resunetpp(a, dev, rank, world, multi, dtype)
    torch.manual_seed(0)
    model = ResUNet(image_shape=(a.patch,) * 3 + (1,), activation="elu", feature_maps=FM, drop_values=[0.0] * 5, normalization="in",
                    yx_down=[2] * 4, z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5, compute_dtype=dtype).to(dev)
    init_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}     # the sliding-window sections run with THESE weights at every N
    V = a.vol if a.vol is not None else 1024     # cfg 3's own size at every N (one GPU holds it: 4.3 GB in, 4.3 GB out; a timed pass is ~2.6 s)
    inf_name = {"mix16": "f16"}.get(a.dtype, a.dtype) if (a.infer_dtype == "same" or a.dtype == "f32") else a.infer_dtype
    inf_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[inf_name]
    if a.mode == "sliding":
        model.compute_dtype = inf_dtype
        rec = run_sliding(a, model, dev, rank, world, V, a.warmup, a.steps)
        if rank == 0:
            rec.update(higher_is_better=True, vs_baseline=None, dtype=inf_name, data="synthetic")
        finish(rec, multi, rank)
        return

    x, tgt = synth_batch(a.batch, a.patch, dev, seed=rank)
    from biapy_amd.losses import BCEWithLogitsLoss
    loss_fn = BCEWithLogitsLoss()                                      # LOSS.TYPE="CE" -> BCEWithLogits (metrics.py:543-544), fused HIP passes
    line = None

    # ======================================================= train =======================================================
    if a.mode in ("all", "train") or a.breakdown and a.mode != "infer":
        model.train()
        net = model
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

        def eager_step():
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(net(x), tgt)
            loss.backward()
            opt.step()
            return loss

        step = eager_step
        graphed = False
        gstep = None
        if want_graph:
            # HIP-graph capture of the whole step (forward, loss, backward, AdamW): one launch per step instead of ~170
            try:
                from biapy_amd.graphs import DataParallelTrainStep, GraphedTrainStep

                gstep = DataParallelTrainStep(net, loss_fn, opt, x, tgt) if multi else GraphedTrainStep(net, loss_fn, opt, x, tgt)
                step = lambda: gstep()  # noqa: E731
                graphed = True
            except Exception as e:  # capture is an optimisation, never a requirement
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
                if multi:                                              # still exchange gradients: same three phases, eager
                    from biapy_amd.graphs import DataParallelTrainStep

                    estep = DataParallelTrainStep(net, loss_fn, opt, x, tgt, graph=False, broadcast_parameters=False)
                    step = lambda: estep()  # noqa: E731
        feed_note = "device-resident batch, H2D excluded (the host-fed step is timed beside it: host_fed_ms_per_step)"
        inner_step, fed_step = step, None
        if not a.breakdown:
            # The reference copies every batch to the device (train_engine.py:116, 125: two H2D copies per step).  Here: the batch lies in PINNED host
            # memory, a copy stream brings the NEXT step's batch into a staging pair while this step computes, and the step begins with a
            # device-to-device copy of the staged pair into its (graph-static) input tensors - all inside the timed region.
            xs_h, ts_h = x.cpu().pin_memory(), tgt.cpu().pin_memory()       # (channels_last_3d strides kept: a plain byte copy)
            xs_d, ts_d = torch.empty_like(x), torch.empty_like(tgt)
            copy_stream = torch.cuda.Stream(device=dev)
            ev_staged, ev_taken = torch.cuda.Event(), torch.cuda.Event()

            def stage():
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ev_taken)
                    xs_d.copy_(xs_h, non_blocking=True)
                    ts_d.copy_(ts_h, non_blocking=True)
                    ev_staged.record(copy_stream)

            ev_taken.record(torch.cuda.current_stream())
            stage()

            def fed_step():
                main = torch.cuda.current_stream()
                main.wait_event(ev_staged)
                x.copy_(xs_d)
                tgt.copy_(ts_d)
                ev_taken.record(main)
                stage()
                return inner_step()

            host_note = ("pinned host batch (%.1f MB): async H2D on a copy stream beside the previous step, then a device-to-device copy into the step's static "
                         "inputs; both inside the timed region" % ((xs_h.numel() * xs_h.element_size() + ts_h.numel() * ts_h.element_size()) / 1e6))
            if a.feed == "host":
                step, feed_note = fed_step, host_note + " (the device-resident step is timed beside it: device_resident_ms_per_step)"
        for _ in range(a.warmup):
            out = step()
        torch.cuda.synchronize()
        if a.warmup and not torch.isfinite(out.detach()).all():
            raise SystemExit("non-finite loss in warm-up")

        if a.breakdown:
            return breakdown(a, L, step, "train")

        self_check = None
        if multi and (a.self_check or world > 1):
            # every rank must hold the SAME averaged gradients and the same parameters after a step (all-reduce + identical optimizer steps):
            # rank-local double checksums are gathered and compared on every rank; a mismatch fails the run instead of producing a number
            gsum = torch.zeros((), dtype=torch.float64, device=dev)
            for p_ in model.parameters():
                gsum += p_.detach().double().sum() * 1e-3
                if p_.grad is not None:
                    gsum += p_.grad.detach().double().abs().sum()
            allsums = [torch.zeros_like(gsum) for _ in range(world)]
            dist.all_gather(allsums, gsum)
            vals = [float(v) for v in allsums]
            agree = all(v == vals[0] for v in vals)
            self_check = dict(ranks=world, gradient_and_parameter_checksums_agree=agree, checksum=vals[0])
            if not agree:
                raise SystemExit(f"[bench] self-check FAILED: the ranks' gradient / parameter checksums differ after the all-reduce: {vals}")

        prof = L.Profile(names=CONV_ENTRIES)
        if not graphed:
            L.lib.prof = prof
        elapsed = _timed(step, a.steps, world, dev)
        L.lib.prof = None
        device_resident_ms = host_fed_ms = None
        if fed_step is not None:      # the other feed, the same K steps, as a sub-field beside the headline
            other = 1e3 * _timed(inner_step if a.feed == "host" else fed_step, a.steps, world, dev) / a.steps
            if a.feed == "host":
                device_resident_ms = other
            else:
                host_fed_ms = other
        prof_steps = a.steps
        fb_graphs = getattr(model, "_graphs", None) is not None
        if fb_graphs:
            model.release_graphs()
        if (graphed or fb_graphs) and not a.no_launch_events:
            # a graph replay has no per-launch events: time the same launches on the same stream in eager steps right after
            # the timed region (same kernels, same shapes; `value` above is NOT taken from these steps)
            prof_steps = min(a.steps, 5)
            L.lib.prof = prof
            for _ in range(prof_steps):
                eager_step()
            torch.cuda.synchronize()
            L.lib.prof = None
        value = world * a.batch * a.patch ** 3 * a.steps / elapsed
        if rank == 0:
            line = dict(
                metric="voxels/sec 3D ResUNet 128^3 patch (train: fwd+bwd+AdamW; sub-records: infer, sliding)",
                value=value, unit="voxels/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype=DTYPE_NAMES[a.dtype], data="synthetic", feed=feed_note, device_resident_ms_per_step=device_resident_ms, host_fed_ms_per_step=host_fed_ms,
                config=dict(workload="cfg2: 3D ResUNet fm=16-32-64-128-256 IN+ELU, %d^3x1 patches, batch %d/GPU, train" % (a.patch, a.batch),
                            global_batch=world * a.batch, patch=a.patch, parallelism="dp%d" % world, mode="train"),
                launch=(("hip-graph replays (forward+loss+backward up to the first encoder block | its backward | optimizer), the flat-gradient RCCL all-reduce "
                         "of everything but the first block running beside the second replay" if getattr(gstep, "overlapped", False) else
                         "hip-graph replays (forward+loss+backward | optimizer) around one flat-gradient RCCL all-reduce") if multi else
                        "hip-graph replay (whole step)") if graphed else ("hip-graph replay (forward, backward) + eager DDP/optimizer"
                                                                        if fb_graphs else "eager"),
                mfma_frac_end_to_end=round(value * FLOP_PER_VOXEL_FWD * 3 / (world * MFMA_PEAK_BF16), 5),
            )
            line["roofline"], line["roofline_top3"] = conv_roofline(prof, prof_steps, a.dtype,
                                                                    "eager steps right after the timed region (the timed region replays HIP graphs)"
                                                                    if (graphed or fb_graphs) else "the timed region")
            if self_check is not None:
                line["self_check"] = self_check
            # insurance for the first multi-GPU run: the train record is on stderr before the sliding-window exchange starts
            print("[bench] train record (the JSON line follows at the end): " + json.dumps({k: line[k] for k in ("value", "ms_per_step", "n_gpus", "self_check") if k in line}),
                  file=sys.stderr, flush=True)
        del step, eager_step, opt
        if graphed:
            del gstep
        net = None
        for p in model.parameters():
            p.grad = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    # ======================================================= infer =======================================================
    infer_rec = None
    if a.mode in ("all", "infer"):
        model.eval()
        model.compute_dtype = inf_dtype                 # the engine is rebuilt for the inference storage type (fp16 by default)
        want_graph = a.graph == "on" or (a.graph == "auto" and not a.breakdown)   # no collective inside an inference step
        eager_inf = lambda: model.predict_proba(x)  # noqa: E731
        step = eager_inf
        graphed = False
        if want_graph:
            try:
                from biapy_amd.graphs import GraphedInference

                ginf = GraphedInference(model.predict_proba, x)
                step = lambda: ginf()  # noqa: E731
                graphed = True
            except Exception as e:
                print(f"[bench] inference graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if a.breakdown:
            return breakdown(a, L, step, "infer")
        prof = L.Profile(names=CONV_ENTRIES)
        if not graphed:
            L.lib.prof = prof
        elapsed = _timed(step, a.steps, world, dev)
        L.lib.prof = None
        prof_steps = a.steps
        if graphed and not a.no_launch_events:
            prof_steps = min(a.steps, 5)
            L.lib.prof = prof
            for _ in range(prof_steps):
                eager_inf()
            torch.cuda.synchronize()
            L.lib.prof = None
        value = world * a.batch * a.patch ** 3 * a.steps / elapsed
        if rank == 0:
            infer_rec = dict(
                metric="voxels/sec 3D ResUNet 128^3 patch (inference forward + fused sigmoid)", value=value, unit="voxels/s", n_gpus=world,
                steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, scaling="weak",
                config=dict(workload="cfg2: 3D ResUNet fm=16-32-64-128-256 IN+ELU, %d^3x1 patches, batch %d/GPU, infer" % (a.patch, a.batch),
                            global_batch=world * a.batch, patch=a.patch, parallelism="replicas x%d" % world, mode="infer"),
                launch="hip-graph replay (weights packed inside the graph)" if graphed else "eager", dtype=inf_name,
                mfma_frac_end_to_end=round(value * FLOP_PER_VOXEL_FWD / (world * MFMA_PEAK_BF16), 5))
            infer_rec["roofline"], infer_rec["roofline_top3"] = conv_roofline(
                prof, prof_steps, inf_name, "eager forwards right after the timed region (the timed region replays a HIP graph)" if graphed else "the timed region")
        if graphed:
            del ginf
        del step
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if a.mode == "infer":
            if rank == 0:
                infer_rec.update(higher_is_better=True, vs_baseline=None, data="synthetic")
            finish(infer_rec, multi, rank)
            return

    # ====================================================== sliding ======================================================
    sliding_rec = None
    if a.mode == "all":
        # Never lose the train line to a hung exchange (the multi-GPU sliding path has not seen an 8-GPU node before the driver's
        # scaling run): a watchdog prints the line without the sliding record and ends the process.
        done = threading.Event()

        def watchdog():
            if not done.wait(a.sliding_timeout):
                if rank == 0 and line is not None:
                    line["infer"] = infer_rec
                    line["sliding"] = dict(error="timed out after %.0f s" % a.sliding_timeout)
                    print(json.dumps(line), flush=True)
                os._exit(0)

        if world > 1:
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            model.load_state_dict(init_sd)               # the initial weights (same seed on every rank): the checksum of the blended volume is then
            model.compute_dtype = inf_dtype              # comparable between N = 1 and N = 8 (profiles/sliding_checksums.json)
            sliding_rec = run_sliding(a, model, dev, rank, world, V, 1, max(1, min(a.steps, 2)))
            if sliding_rec is not None:
                sliding_rec["dtype"] = inf_name
        except Exception as e:  # the headline line must survive
            sliding_rec = dict(error=f"{type(e).__name__}: {e}")
        done.set()

    # BASELINE.json labels cfg 2 "bf16": the headline runs the mixed mode (fp16 forward inside the Dice bar, bf16 gradients); the pure-bf16 train
    # step - same kernels, same 16 bits per element, forward outside the Dice bar - is timed beside it so the labelled dtype has its own number
    bf16_rec = None
    if a.mode in ("all", "train") and a.dtype == "mix16" and world == 1 and not multi and not a.no_bf16_record and not a.breakdown:
        try:
            bf16_rec = run_train_bf16(a, dev)
        except Exception as e:  # noqa: BLE001
            bf16_rec = dict(error=f"{type(e).__name__}: {e}")
        if line is not None:
            line["train_bf16"] = bf16_rec
    # cfg 4 (ResUNet++ 80^3, B/C/D loss) as a sub-record of the single-GPU line: the second model family the path covers
    cfg4_rec = None
    if a.mode == "all" and world == 1 and not multi and not a.no_cfg4:
        try:
            del model
            torch.cuda.empty_cache()
            cfg4_rec = run_resunetpp(a, dev, rank, world, multi, dtype, as_record=True)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 - the headline line must survive
            cfg4_rec = dict(error=f"{type(e).__name__}: {e}")

    # cfg 5 (RCAN-3D x4, 64^3 -> 256^3, fp16 inference) as a sub-record of the single-GPU line
    cfg5_rec = None
    if a.mode == "all" and world == 1 and not multi and not a.no_cfg5:
        try:
            cfg5_rec = run_rcan_sr(a, dev)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 - the headline line must survive
            cfg5_rec = dict(error=f"{type(e).__name__}: {e}")

    if rank == 0 and line is not None:
        # the sliding-window checksum against the single-GPU run of the same volume (profiles/sliding_checksums.json, committed)
        if sliding_rec is not None and sliding_rec.get("checksum") is not None:
            try:
                ref = json.load(open(os.path.join(ROOT, "profiles", "sliding_checksums.json"))).get(str(V))
            except (OSError, ValueError):
                ref = None
            if ref is not None:
                ok = abs(sliding_rec["checksum"] - ref["checksum"]) <= 1e-9 * max(1.0, abs(ref["checksum"]))
                sliding_rec["checksum_matches_single_gpu_run"] = ok
                sliding_rec["checksum_reference"] = ref
                if not ok:
                    print(f"[bench] self-check: the gathered cfg-3 volume's checksum {sliding_rec['checksum']!r} differs from the single-GPU run's {ref['checksum']!r}",
                          file=sys.stderr)
        # flat sub-record values FIRST (a truncated tail of the line still carries them), the long objects behind
        flat = {}
        for tag, rec in (("train_bf16", bf16_rec), ("infer", infer_rec), ("sliding", sliding_rec), ("cfg4", cfg4_rec), ("cfg5", cfg5_rec)):
            if isinstance(rec, dict) and "value" in rec:
                flat[f"{tag}_value"] = rec["value"]
                flat[f"{tag}_ms_per_step"] = rec.get("ms_per_step")
                flat[f"{tag}_unit"] = rec.get("unit")
            elif isinstance(rec, dict) and "error" in rec:
                flat[f"{tag}_error"] = rec["error"]
        long_keys = ("roofline", "roofline_top3")
        out = {k: v for k, v in line.items() if k not in long_keys}
        out.update(flat)
        for k in long_keys:
            if k in line:
                out[k] = line[k]
        out["infer"], out["sliding"] = infer_rec, sliding_rec
        if cfg4_rec is not None:
            out["cfg4_resunetpp"] = cfg4_rec
        if cfg5_rec is not None:
            out["cfg5_rcan_sr"] = cfg5_rec
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.patch, quick=a.quick_cpu_baseline)
        finish(out, multi, rank)
        return
    finish(None, multi, rank)


def finish(rec, multi, rank):
    """The ONE JSON line as the LAST line of stdout.  RCCL writes a version banner through C stdio when the first communicator comes up; with stdout
    redirected that buffer is flushed at process exit, i.e. AFTER a line printed from Python (seen in profiles/r03_bench_train_dp_1rank.json before
    this function existed).  So: drain C stdio first, print the line, tear the process group down, and leave without running exit handlers that
    could print again."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    if rank == 0 and rec is not None:
        sys.stdout.write(json.dumps(rec) + "\n")
    sys.stdout.flush()
    sys.stderr.flush()
    if multi:
        import threading

        def teardown():
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
        t = threading.Thread(target=teardown, daemon=True)     # the line is out: a communicator teardown that hangs must not hold the launcher
        t.start()
        t.join(20.0)
        os._exit(0)


def run_train_bf16(a, dev):
    """The cfg-2 train step with compute_dtype = bfloat16 (what BASELINE.json's config line names), graph-replayed, a few steps: a sub-record."""
    from biapy_amd.graphs import GraphedTrainStep
    from biapy_amd.losses import BCEWithLogitsLoss
    from biapy_amd.resunet import ResUNet

    torch.manual_seed(0)
    m = ResUNet(image_shape=(a.patch,) * 3 + (1,), activation="elu", feature_maps=FM, drop_values=[0.0] * 5, normalization="in", yx_down=[2] * 4,
                z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5, compute_dtype=torch.bfloat16).to(dev).train()
    x, tgt = synth_batch(a.batch, a.patch, dev, seed=0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)
    gstep = GraphedTrainStep(m, BCEWithLogitsLoss(), opt, x, tgt)
    steps = max(3, min(a.steps, 10))
    for _ in range(2):
        gstep()
    elapsed = _timed(lambda: gstep(), steps, 1, dev)
    rec = dict(value=a.batch * a.patch ** 3 * steps / elapsed, unit="voxels/s", ms_per_step=1e3 * elapsed / steps, steps=steps, dtype="bf16",
               dice_status="pure bf16 storage: the forward does NOT meet Dice delta < 1e-4 against the fp32 reference (measured 0.8e-4 .. 1.4e-4 on a trained "
                           "model, DESIGN.md section 5); the headline `value` is the mixed mode, whose forward does")
    del gstep, opt, m
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return rec


def run_resunetpp(a, dev, rank, world, multi, dtype, as_record=False):
    """cfg 4: ResUNet++ (fm 16-32-64-128-256), 80^3 x 1 patches, three output channels (B, C: BCE on logits; D: MSE through tanh),
    batch --batch per GPU, one step = fwd + loss + bwd + AdamW; data parallel = DistributedDataParallel over RCCL (the module is an
    ordinary nn.Module with one autograd.Function)."""
    from biapy_amd.losses import InstanceChannelsLoss
    from biapy_amd.resunetpp import ResUNetPlusPlus

    P = 80 if a.patch == 128 else a.patch
    if dtype == torch.float16 and torch.float16 not in getattr(ResUNetPlusPlus, "supported_compute_dtypes", ()):
        dtype = torch.bfloat16                             # the tape engine's backward kernels are bf16 / f32
    torch.manual_seed(0)
    fm = [16, 32, 64, 128, 256]
    model = ResUNetPlusPlus(image_shape=(P, P, P, 1), activation="elu", feature_maps=fm, drop_values=[0.0] * 5, normalization="in", yx_down=[2] * 4,
                            z_down=[2] * 4, output_channels=[3], output_channel_info=["BCD"], head_activations=["ce_sigmoid", "ce_sigmoid", "tanh"],
                            isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5, compute_dtype=dtype).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(rank)
    x = torch.randn(a.batch, 1, P, P, P, generator=g, device=dev)
    n = torch.randn(a.batch, 2, P, P, P, generator=g, device=dev)
    tgt = torch.cat([(F.avg_pool3d(n, 9, stride=1, padding=4) > 0).float(), torch.rand(a.batch, 1, P, P, P, generator=g, device=dev) * 2 - 1], 1)
    loss_fn = InstanceChannelsLoss(channel_weights=(1, 1, 1), out_channels=["B", "C", "D"], losses_to_use=["bce", "bce", "mse"]).to(dev)
    want_graph = a.graph != "off" and a.dp != "ddp"
    use_ddp = multi and not want_graph
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], gradient_as_bucket_view=True, bucket_cap_mb=64) if use_ddp else model
    try:                                                     # the fused multi-tensor AdamW, as on the ResUNet line (the unfused capturable one
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True, capturable=want_graph)   # issues two strided divisions per parameter tensor)
    except Exception:  # noqa: BLE001
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=want_graph)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(net(x), tgt)
        loss.backward()
        opt.step()
        return loss

    launch = "eager (tape engine%s)" % ("; DistributedDataParallel" if use_ddp else "")
    if a.breakdown:
        from biapy_amd import _lib as L
        for _ in range(2):
            step()
        return breakdown(a, L, step, "ResUNet++ train")
    if want_graph:
        # the tape engine issues ~1500 launches per step from Python: replay them (same classes as the ResUNet bench line)
        try:
            from biapy_amd.graphs import DataParallelTrainStep, GraphedTrainStep
            g_step = (DataParallelTrainStep(model, loss_fn, opt, x, tgt, graph=True, warmup=max(2, a.warmup)) if multi
                      else GraphedTrainStep(model, loss_fn, opt, x, tgt, warmup=max(2, a.warmup)))
            step = lambda: g_step()
            launch = ("hip-graph replays (forward+loss+backward | optimizer) around one flat-gradient RCCL all-reduce" if multi
                      else "hip-graph replay (whole step)")
        except Exception as e:  # noqa: BLE001 - the eager tape is the fallback of the bench, never of the product
            if multi:
                raise
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if a.warmup and not torch.isfinite(out.detach()).all():
        raise SystemExit("non-finite loss in warm-up")
    elapsed = _timed(step, a.steps, world, dev)
    nparams = sum(p.numel() for p in model.parameters())
    value = world * a.batch * P ** 3 * a.steps / elapsed
    rec = dict(
        metric="voxels/sec 3D ResUNet++ %d^3 patch (train: fwd + B/C/D loss + bwd + AdamW)" % P, value=value, unit="voxels/s", n_gpus=world,
        steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
        dtype={torch.float16: DTYPE_NAMES["mix16"], torch.bfloat16: "bf16", torch.float32: "f32"}[dtype], data="synthetic", launch=launch,
        config=dict(workload="cfg4: 3D ResUNet++ fm=16-32-64-128-256 IN+ELU, %d^3x1 patches, 3 channels (B,C,D), batch %d/GPU, train" % (P, a.batch),
                    global_batch=world * a.batch, patch=P, parameters=nparams, parallelism="dp%d" % world, mode="train"),
        mfma_frac_end_to_end=round(value * 1044917 * 3 / (world * MFMA_PEAK_BF16), 5))
    if as_record:                                           # the sub-record of the default line (main): the caller prints
        return rec
    finish(rec, multi, rank)


def run_rcan_sr(a, dev):
    """cfg 5: 3D super-resolution x4 with RCAN-3D (10 groups x 20 RCABs, 16 filters), one 64^3 patch -> 256^3, fp16, inference: the
    trunk at 64^3, the up-scaling stage (conv 16 -> 1024 channels with the 3-D pixel shuffle fused into its store) and the last conv at
    256^3.  One step = one forward of one patch, replayed from a HIP graph (the trunk is a chain of ~800 dependent kernels)."""
    from biapy_amd.rcan import rcan

    torch.manual_seed(0)
    m = rcan(ndim=3, num_channels=1, filters=16, scale=4, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=True, out_channels=1,
             head_activations=["linear"], compute_dtype=torch.float16).to(dev).eval()
    x = torch.randn(1, 1, 64, 64, 64, device=dev)
    step, launch = (lambda: m(x)), "eager"
    with torch.no_grad():
        y = m(x)
        try:
            from biapy_amd.graphs import GraphedInference

            ginf = GraphedInference(lambda t: m(t), x)
            step, launch = (lambda: ginf()), "hip-graph replay"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] cfg5 graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        for _ in range(2):
            step()
        steps = max(3, min(a.steps, 10))
        elapsed = _timed(step, steps, 1, dev)
    out_vox = 256 ** 3
    # forward FLOPs: 401 convs 16->16 + sf (1->16) at 64^3, 16 -> 1024 at 64^3, 16 -> 1 at 256^3 (2 * 27 * Cin * Cout per voxel)
    flops = 2 * 27 * (64 ** 3 * (401 * 256 + 16 + 16 * 1024) + 256 ** 3 * 16)
    return dict(metric="voxels/sec 3D RCAN x4 super-resolution 64^3 -> 256^3 (inference forward, output voxels)", value=out_vox * steps / elapsed, unit="voxels/s",
                n_gpus=1, steps=steps, ms_per_step=1e3 * elapsed / steps, dtype="f16", launch=launch, input_voxels_per_s=64 ** 3 * steps / elapsed,
                config=dict(workload="cfg5: RCAN-3D x4 (10 x 20 RCABs, 16 filters), one 64^3 patch -> 256^3, fp16, inference", patch=64, scale=4),
                mfma_frac_end_to_end=round(flops * steps / elapsed / MFMA_PEAK_BF16, 5), output_checksum=float(y.double().mean().item()),
                note="3-D pixel shuffle is defined by this package (the reference's 3-D up-scaling branch raises): parity unpinned against BiaPy")


def breakdown(a, L, step, mode):
    prof = L.Profile()
    L.lib.prof = prof
    step()
    torch.cuda.synchronize()
    L.lib.prof = None
    rows = sorted(prof.summary().items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"# per-call event timing of one {mode} step (B={a.batch}, {a.patch}^3, {a.dtype}); sum = {tot:.3f} ms")
    for (name, key), (cnt, ms) in rows:
        fl = conv_flops(name, key) * cnt
        tf = f"{fl / (ms * 1e-3) / 1e12:8.1f} TF/s" if fl else " " * 13
        print(f"{ms:9.3f} ms {100 * ms / tot:5.1f}% x{cnt:<3d} {tf} {name} {key}")


if __name__ == "__main__":
    main()
