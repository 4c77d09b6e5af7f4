"""World-size>1 logic on CPU with gloo: slab partition + boundary hand-over of the sharded blend, and DDP over a
whole-network autograd.Function.  The compute inside is a NumPy stand-in built from the oracle (test infrastructure);
what is under test is biapy_amd.workflow's partition / exchange protocol, which is device-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from biapy_amd import workflow
from oracle import tiling_oracle as T


class NumpyBlend:
    """Same contract as workflow.DeviceBlend, evaluated with the oracle's arithmetic on CPU tensors."""

    def __init__(self, vol_zyx, patch_zyx, overlap):
        self.g = T.merge_grid(vol_zyx, patch_zyx, overlap, (0, 0, 0))
        self.P = tuple(patch_zyx)
        self.win = T.spline_window(self.P, tuple(g.ov_pixels for g in self.g))[..., 0]

    def blend(self, patches, z_lo, z_hi, rows, acc=None, wacc=None, seed=False, write_partial=False, out_dtype=None, out=None):
        gz, gy, gx = self.g
        p = patches.numpy()
        C = p.shape[-1]
        Y, X = gy.limit, gx.limit
        num = np.zeros((z_hi - z_lo, Y, X, C), np.float32)
        ws = np.zeros((z_hi - z_lo, Y, X, 1), np.float32)
        if seed:
            num[:] = acc.numpy()
            ws[:] = wacc.numpy()
        c = 0
        for iz in range(rows[0], rows[1]):
            z0 = gz.start(iz)
            for y0 in gy.starts():
                for x0 in gx.starts():
                    a, b = max(z0, z_lo), min(z0 + self.P[0], z_hi)
                    if b > a:
                        w = self.win[a - z0:b - z0, :, :, None]
                        num[a - z_lo:b - z_lo, y0:y0 + self.P[1], x0:x0 + self.P[2]] += p[c, a - z0:b - z0] * w
                        ws[a - z_lo:b - z_lo, y0:y0 + self.P[1], x0:x0 + self.P[2]] += w
                    c += 1
        if write_partial:
            acc.copy_(torch.from_numpy(num))
            wacc.copy_(torch.from_numpy(ws))
            return None
        res = torch.from_numpy(np.true_divide(num, ws + 1e-18).astype(np.float32))
        if out is not None:
            out.copy_(res)
            return out
        return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, vshape, pshape, ov, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(3)
    g = T.merge_grid(vshape[:3], pshape[:3], ov, (0, 0, 0))
    n = g[0].n * g[1].n * g[2].n
    pred = rs.rand(n, *pshape).astype(np.float32)
    ref = T.merge(pred, vshape, overlap=ov)
    plans = workflow.plan_slabs(g[0].starts(), pshape[0], vshape[0], world)
    lo, hi = plans[rank].rows
    per_row = g[1].n * g[2].n
    mine = torch.from_numpy(pred[lo * per_row:hi * per_row])
    out = workflow.sharded_blend(NumpyBlend(vshape[:3], pshape[:3], ov), mine, plans, rank, world, vshape[1], vshape[2], vshape[3], gather="all")
    ok = bool((out.numpy().view(np.uint32) == ref.view(np.uint32)).all())
    out0 = workflow.sharded_blend(NumpyBlend(vshape[:3], pshape[:3], ov), mine, plans, rank, world, vshape[1], vshape[2], vshape[3], gather="rank0")
    ok = ok and ((out0 is None) == (rank != 0))
    if rank == 0:
        ok = ok and bool((out0.numpy().view(np.uint32) == ref.view(np.uint32)).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,vshape,pshape,ov", [
    (2, (72, 40, 40, 1), (32, 32, 32, 1), (0.5, 0.5, 0.5)),
    (3, (80, 24, 28, 2), (32, 16, 16, 2), (0.6, 0.25, 0.0)),     # overlap > 50 %: three patch rows cover one slice
    (4, (40, 20, 20, 1), (32, 16, 16, 1), (0.5, 0.0, 0.5)),      # more ranks than patch rows: idle ranks must not dead-lock
    (8, (1024, 24, 24, 1), (128, 16, 16, 1), (0.5, 0.5, 0.5)),   # cfg 3's exact Z plan on 8 ranks: 2 patch rows per rank, 68-slice hand-over
])
def test_sharded_blend_bit_exact_gloo(world, vshape, pshape, ov):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, vshape, pshape, ov, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_input_slab_is_what_the_ranks_patches_read():
    """SlidingWindowPredictor.input_slab / tiling.crop_rows_needed (each GPU holds only its input slab + halo)
    against the oracle's crop coordinates: the slab is exactly the hull of the slices the rank's patches read, reflect-padding
    sources at the volume ends included; at cfg 3 a rank reads 188 of 1024 slices."""
    from biapy_amd import tiling
    from biapy_amd.workflow import SlidingWindowPredictor, split_rows

    for vol, patch, ov, pad in (((1024, 64, 64), (128, 32, 32), (0.5, 0.5, 0.5), (0, 0, 0)), ((100, 40, 40), (32, 16, 16), (0.25, 0, 0), (6, 2, 2)),
                                ((77, 20, 20), (24, 16, 16), (0.5, 0, 0), (4, 0, 0)), ((48, 20, 20), (48, 16, 16), (0, 0, 0), (10, 0, 0))):
        coords = T.crop_coords(vol, patch, ov, pad)                    # patch extents in PADDED coordinates
        g = T.crop_grid(vol, patch, ov, pad)
        per_row = g[1].n * g[2].n
        for world in (1, 2, 3, 8):
            sw = SlidingWindowPredictor(None, patch, ov, pad, forward=lambda x: x)
            for rank, (lo, hi) in enumerate(split_rows(g[0].n, world)):
                got = sw.input_slab(vol, rank, world)
                if hi <= lo:
                    assert got == (0, 0)
                    continue
                zs = set()
                for c in coords[lo * per_row:hi * per_row]:
                    for z in range(int(c[0]) - pad[0], int(c[1]) - pad[0]):
                        zs.add(-z if z < 0 else (2 * (vol[0] - 1) - z if z >= vol[0] else z))    # np.pad "reflect"
                assert got == (min(zs), max(zs) + 1), (vol, patch, world, rank, got, (min(zs), max(zs) + 1))
    sw = SlidingWindowPredictor(None, (128, 128, 128), (0.5, 0.5, 0.5), (0, 0, 0), forward=lambda x: x)
    assert [sw.input_slab((1024, 1024, 1024), r, 8) for r in (0, 3, 7)] == [(0, 188), (360, 548), (840, 1024)]
    assert tiling.crop_rows_needed((1024, 1024, 1024), (128, 128, 128), (0.5, 0.5, 0.5), (0, 0, 0), 0, 16) == (0, 1024)


def test_plan_slabs_rejects_backward_running_rows():
    """The reference's shift-back rule can make patch starts non-monotonic (overlap > 50 % on a short axis): not shardable."""
    g = T.merge_grid((14, 10, 40), (6, 9, 20), (0.6, 0.25, 0.6), (0, 2, 0))
    assert list(g[0].starts()) == [0, 2, 4, 6, 4, 6, 8]
    workflow.plan_slabs(g[0].starts(), 6, 14, 1)
    with pytest.raises(ValueError, match="not monotonic"):
        workflow.plan_slabs(g[0].starts(), 6, 14, 2)


def test_plan_slabs_cfg3():
    """cfg 3: 16 patch rows (starts 0,60,..,840,896), 8 GPUs -> two rows each, 68-slice hand-over, disjoint cover of [0,1024)."""
    g = T.merge_grid((1024, 1024, 1024), (128, 128, 128), (0.5, 0.5, 0.5), (0, 0, 0))
    plans = workflow.plan_slabs(g[0].starts(), 128, 1024, 8)
    assert [p.rows for p in plans] == [(2 * r, 2 * r + 2) for r in range(8)]
    assert plans[0].own == (0, 120) and plans[0].recv is None and plans[0].send == (120, 188)
    assert plans[3].own == (360, 480) and plans[3].recv == (360, 428) and plans[3].send == (480, 548)
    assert plans[7].own == (840, 1024) and plans[7].send is None
    cover = np.zeros(1024, int)
    for p in plans:
        cover[p.own[0]:p.own[1]] += 1
    assert (cover == 1).all()


class _WholeNetFn(torch.autograd.Function):
    """Stand-in with the same autograd shape as biapy_amd.resunet._ResUNetFn: one Function, all parameter grads at once."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.save_for_backward(x, *params)
        return x * params[0].sum() + params[1].sum()

    @staticmethod
    def backward(ctx, g):
        x, w, b = ctx.saved_tensors
        return None, torch.full_like(w, (g * x).sum().item()), torch.full_like(b, g.sum().item())


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(5))
        self.b = torch.nn.Parameter(torch.zeros(3))

    def forward(self, x):
        return _WholeNetFn.apply(x, self.w, self.b)


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = torch.nn.parallel.DistributedDataParallel(_Net())
    x = torch.full((4,), float(rank + 1))
    net(x).sum().backward()
    q.put((rank, net.module.w.grad[0].item(), net.module.b.grad[0].item()))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_averages_grads_of_a_whole_network_function():
    """DistributedDataParallel (base_workflow.py:952-958) must average gradients that arrive from ONE autograd.Function."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    # rank0: sum(g*x)=4, rank1: 8 -> mean 6 ; bias grad 4 on both -> 4
    assert all(abs(w - 6.0) < 1e-6 and abs(b - 4.0) < 1e-6 for _, w, b in res), res


def _small_net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ELU(), torch.nn.Conv3d(4, 1, 1))


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from biapy_amd.graphs import DataParallelTrainStep

    g = torch.Generator().manual_seed(10 + rank)                       # every rank its own shard of the data
    xs = [torch.randn(2, 1, 6, 6, 6, generator=g) for _ in range(3)]
    ts = [(torch.rand(2, 1, 6, 6, 6, generator=g) > 0.5).float() for _ in range(3)]
    loss_fn = torch.nn.BCEWithLogitsLoss()
    # reference behaviour: DDP wrap (base_workflow.py:952-958), both start from rank 0's weights
    ref = torch.nn.parallel.DistributedDataParallel(_small_net(0))
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    for x, t in zip(xs, ts):
        ropt.zero_grad(set_to_none=True)
        loss_fn(ref(x), t).backward()
        ropt.step()
    # flat-gradient step; rank 1 deliberately starts from different weights: the constructor must broadcast rank 0's
    net = _small_net(0 if rank == 0 else 7)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    step = DataParallelTrainStep(net, loss_fn, opt, xs[0], ts[0], graph=False)
    for x, t in zip(xs, ts):
        step(x, t)
    err = max((a - b).abs().max().item() for a, b in zip(net.parameters(), ref.module.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], o) for o in gathered)
    step._check_views()
    q.put((rank, err, same))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_data_parallel_step_matches_ddp():
    """biapy_amd.graphs.DataParallelTrainStep (eager form): same weights as DistributedDataParallel after 3 AdamW steps on
    disjoint shards, identical on every rank, parameters broadcast from rank 0 at construction."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(err < 1e-6 and same for _, err, same in res), res


class _PPOnCpu(torch.nn.Module):
    """The ResUNet++ drop-in's PARAMETERS (biapy_amd.resunetpp.ResUNetPlusPlus is an ordinary parameter holder: same names, shapes and order as
    the reference) driven by the CPU oracle graph: what the multi-process CPU tests can run of cfg 4 - the device engine needs the GPU."""

    def __init__(self, seed):
        super().__init__()
        from biapy_amd.resunetpp import ResUNetPlusPlus

        torch.manual_seed(seed)
        self.fm = [16, 32, 64]
        self.net = ResUNetPlusPlus(image_shape=(16, 16, 16, 1), activation="elu", feature_maps=self.fm, drop_values=[0.0] * 3, normalization="in", k_size=3,
                                   upsample_layer="convtranspose", yx_down=[2, 2], z_down=[2, 2], output_channels=[3], output_channel_info=["BCD"],
                                   head_activations=["ce_sigmoid", "ce_sigmoid", "tanh"], isotropy=[True] * 3, larger_io=False, conv_layers=[2] * 3)

    def forward(self, x):
        from oracle import resunetpp_oracle

        return resunetpp_oracle.resunetpp_forward(dict(self.net.named_parameters()), x, self.fm)


def _bcd_loss(logits, target):
    """instance_segmentation_loss for B, C (BCE on logits) and D (MSE through tanh): the CPU statement of biapy_amd.losses.InstanceChannelsLoss."""
    from oracle import loss_oracle

    pred = loss_oracle.apply_head_activations(logits, ["ce_sigmoid", "ce_sigmoid", "tanh"], training=True)
    return loss_oracle.instance_channels(pred, target, ["bce", "bce", "mse"], [1.0, 1.0, 1.0])


def _dp_pp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from biapy_amd.graphs import DataParallelTrainStep

    g = torch.Generator().manual_seed(40 + rank)
    xs = [torch.randn(1, 1, 16, 16, 16, generator=g) for _ in range(2)]
    ts = [torch.cat([(torch.rand(1, 2, 16, 16, 16, generator=g) > 0.5).float(), torch.rand(1, 1, 16, 16, 16, generator=g) * 2 - 1], 1) for _ in range(2)]
    ref = torch.nn.parallel.DistributedDataParallel(_PPOnCpu(0))                      # the reference's route: base_workflow.py:952-958
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    for x, t in zip(xs, ts):
        ropt.zero_grad(set_to_none=True)
        _bcd_loss(ref(x), t).backward()
        ropt.step()
    net = _PPOnCpu(0 if rank == 0 else 7)                                             # rank 1 starts elsewhere: rank 0's weights must be broadcast
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    step = DataParallelTrainStep(net, _bcd_loss, opt, xs[0], ts[0], graph=False)
    for x, t in zip(xs, ts):
        step(x, t)
    step._check_views()                                                               # every p.grad is still a view of the ONE flat all-reduce buffer
    nparam = sum(1 for _ in net.parameters())
    err = max((a - b).abs().max().item() for a, b in zip(net.parameters(), ref.module.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, err, all(torch.equal(gathered[0], o) for o in gathered), nparam, step.flat_grad.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_data_parallel_step_on_the_resunetpp_parameter_set():
    """cfg 4 is the "DDP training" configuration: DataParallelTrainStep over the ResUNet++ drop-in's own parameter set
    (80 tensors at three levels: residual blocks with normalised 3x3x3 shortcuts, bias-free squeeze-excite Linears, ASPP, attention gates, 3-channel
    head) and the B / C / D channel loss - one flat gradient slab, rank 0's weights broadcast, the same weights as DistributedDataParallel
    after two AdamW steps on disjoint shards, identical on both ranks.  The graph (CPU oracle) stands in for the device engine here."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_pp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(err < 2e-6 and same and npar >= 80 and nflat > 100000 for _, err, same, npar, nflat in res), res


def _epoch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from biapy_amd import train_engine as TE

    g = torch.Generator().manual_seed(30 + rank)
    n = 3 if rank == 0 else 5                                           # unequal shard lengths: the epoch average weights by count
    data = [(torch.randn(2, 4, 6, 6, 1, generator=g), (torch.rand(2, 4, 6, 6, 1, generator=g) > 0.5).float()) for _ in range(n)]
    net = _small_net(0)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    with torch.no_grad():
        local = [loss_fn(net(x.permute(0, 4, 1, 2, 3)), t.permute(0, 4, 1, 2, 3)).item() for x, t in data]
    ev = TE.evaluate(None, net, None, loss_fn, None, None, 0, data, loss_names=["loss"], device=torch.device("cpu"))
    q.put((rank, ev["loss"], sum(local), len(local)))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_averages_over_ranks_by_count():
    """train_engine.evaluate: the validation loss is the average over ALL batches of ALL ranks (MetricLogger
    .synchronize_between_processes, train_engine.py:318), also when the ranks hold different numbers of batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    want = (res[0][2] + res[1][2]) / (res[0][3] + res[1][3])
    assert abs(res[0][1] - want) < 1e-6 and abs(res[1][1] - want) < 1e-6, (res, want)


def test_gather_layout_of_the_stitched_volume():
    """the stitched output travels as ONE all-gather.  cfg 3 on 8 ranks gives slabs of 7 x 120 and 1 x 184 slices whose
    starts are multiples of 120: the common part is gathered in place (the collective's output IS the volume), the last rank's 64 extra slices
    follow as one broadcast; layouts that do not line up (idle ranks) take the padded form.  (The gloo cases of
    test_sharded_blend_bit_exact_gloo run both forms end to end, bit-exact.)"""
    g = T.merge_grid((1024, 16, 16), (128, 16, 16), (0.5, 0.5, 0.5), (0, 0, 0))
    plans = workflow.plan_slabs(g[0].starts(), 128, 1024, 8)
    assert [p.own for p in plans] == [(120 * r, 120 * (r + 1)) for r in range(7)] + [(840, 1024)]
    assert workflow.gather_layout(plans, 8) == ("inplace", 120, [(7, (960, 1024))])
    g = T.merge_grid((72, 40, 40), (32, 32, 32), (0.5, 0.5, 0.5), (0, 0, 0))
    kind, piece, tails = workflow.gather_layout(workflow.plan_slabs(g[0].starts(), 32, 72, 2), 2)
    assert kind == "inplace" and all(b > a for _, (a, b) in tails)
    g = T.merge_grid((40, 20, 20), (32, 16, 16), (0.5, 0.0, 0.5), (0, 0, 0))          # 2 patch rows on 4 ranks: two ranks hold nothing
    plans = workflow.plan_slabs(g[0].starts(), 32, 40, 4)
    kind, piece, tails = workflow.gather_layout(plans, 4)
    assert kind == "padded" and piece == max(p.own[1] - p.own[0] for p in plans) and tails is None
