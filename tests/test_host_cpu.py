"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares, host geometry, module schema."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from biapy_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "biapy_amd.h")).read()
    declared = set(re.findall(r"\b(bpx_[a-zA-Z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.startswith("bpx_PK") and d != "bpx_BF16"}
    declared = {d for d in declared if not d.startswith("bpx_stream")}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), f"{name} declared in include/biapy_amd.h but not exported by libbiapy_amd.so"
    assert set(_lib.EXPORTS) <= declared | {"bpx_debug_set_wgrad_tr"}
    assert _lib.lib.bpx_version() >= 100


def test_host_grid_matches_oracle(tiling_golden):
    from biapy_amd import tiling
    from oracle import tiling_oracle as T

    for name in ["c48", "docstring", "cfg3_1024", "template_pad10", "aniso", "odd", "single"]:
        a = tiling_golden[f"coords/{name}/args"]
        ov = tuple(float(v) for v in tiling_golden[f"coords/{name}/overlap"])
        g = tiling.crop_grid(a[0:3], a[4:7], ov, a[8:11])
        starts = [[tiling._start(g[ax], i) for i in range(g[ax].n)] for ax in range(3)]
        ref = tiling_golden[f"coords/{name}/coords"]
        assert sorted(set(ref[:, 0])) == sorted(set(starts[0]))
        assert sorted(set(ref[:, 2])) == sorted(set(starts[1]))
        assert sorted(set(ref[:, 4])) == sorted(set(starts[2]))
        assert len(ref) == g[0].n * g[1].n * g[2].n
    np.testing.assert_array_equal(tiling.taper_1d(128, 68), T.taper_1d(128, 68))


def test_dropin_argument_errors_match_reference():
    from biapy_amd import tiling

    v = np.zeros((16, 16, 16, 1), np.float32)
    with pytest.raises(ValueError, match="4 dimensional"):
        tiling.crop_3D_data_with_overlap(v[..., 0], (8, 8, 8, 1), verbose=False)
    with pytest.raises(ValueError, match="Padding"):
        tiling.crop_3D_data_with_overlap(v, (8, 8, 8, 1), padding=(4, 0, 0), verbose=False)
    with pytest.raises(ValueError, match="greater than"):
        tiling.crop_3D_data_with_overlap(v, (32, 8, 8, 1), verbose=False)
    with pytest.raises(ValueError, match="overlap"):
        tiling.crop_3D_data_with_overlap(v, (8, 8, 8, 1), overlap=(1.0, 0, 0), verbose=False)
    with pytest.raises(AssertionError):
        tiling.merge_3D_data_with_overlap(v, (16, 16, 16, 1), verbose=False)
    coords = tiling.crop_3D_data_with_overlap(v, (8, 8, 8, 1), overlap=(0.5, 0.5, 0.5), verbose=False, load_data=False)
    assert len(coords) == 64 and coords[1].x_start == 3 and coords[-1].z_end == 16


def test_module_state_dict_schema(resunet_golden):
    from biapy_amd.resunet import ResUNet

    m = ResUNet(image_shape=(128, 128, 128, 1), activation="elu", feature_maps=[16, 32, 64, 128, 256], drop_values=[0.0] * 5,
                normalization="in", yx_down=[2] * 4, z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5)
    keys = list(resunet_golden["cfg2/keys"])
    shapes = dict(zip(keys, resunet_golden["cfg2/shapes"]))
    sd = m.state_dict()
    assert list(sd.keys()) == keys
    assert all(str(tuple(v.shape)) == shapes[k] for k, v in sd.items())
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 1, 16, 16, 16))
    with pytest.raises(NotImplementedError):
        ResUNet(image_shape=(64, 64, 1), feature_maps=[16, 32], normalization="in", larger_io=False)


def test_losses_fail_loudly_without_gpu():
    """The device losses have no CPU fallback (product path rule): CPU tensors raise."""
    import pytest
    import torch

    from biapy_amd import losses

    with pytest.raises(RuntimeError, match="MI355X only"):
        losses.DiceCELoss()(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))
    with pytest.raises(NotImplementedError):
        losses.DiceLoss(batch_dice=False)


def test_prepost_fails_loudly_without_gpu():
    import pytest
    import torch

    from biapy_amd import prepost

    with pytest.raises(RuntimeError, match="MI355X only"):
        prepost.threshold_otsu(torch.zeros(4, 4, 4))


# ---- train_engine: step drivers (row T) -------------------------------------------------------------------------------
def _toy_loader(n, B=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 4, 6, 6, 1, generator=g), (torch.rand(B, 4, 6, 6, 1, generator=g) > 0.5).float()) for _ in range(n)]


def _toy_net(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ELU(), torch.nn.Conv3d(4, 1, 1))


def test_train_one_epoch_matches_a_plain_loop():
    """biapy_amd.train_engine.train_one_epoch (eager form, the one that runs without a GPU) == the reference loop's arithmetic
    (train_engine.py:127-180): zero_grad, forward on (B,C,Z,Y,X), loss, backward, step; returns ({loss, lr}, last step)."""
    from biapy_amd import train_engine as TE

    data = _toy_loader(7)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    a, b = _toy_net(), _toy_net()
    oa, ob = torch.optim.AdamW(a.parameters(), lr=1e-2), torch.optim.AdamW(b.parameters(), lr=1e-2)
    stats, last = TE.train_one_epoch(a, loss_fn, data, oa, torch.device("cpu"), epoch=0, patch_size=(4, 6, 6, 1), graph="off", sync_every=3)
    tot = 0.0
    for x, t in data:
        ob.zero_grad()
        loss = loss_fn(b(x.permute(0, 4, 1, 2, 3)), t.permute(0, 4, 1, 2, 3))
        loss.backward()
        ob.step()
        tot += loss.item()
    assert last == 6 and abs(stats["loss"] - tot / 7) < 1e-6 and stats["lr"] == 1e-2
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, atol=1e-7)
    ev = TE.evaluate(a, loss_fn, data[:3], torch.device("cpu"), epoch=0)
    with torch.no_grad():
        ref = sum(loss_fn(b(x.permute(0, 4, 1, 2, 3)), t.permute(0, 4, 1, 2, 3)).item() for x, t in data[:3]) / 3
    assert abs(ev["loss"] - ref) < 1e-6


def test_train_one_epoch_error_behaviour():
    """Same failures as the reference: wrong patch shape -> ValueError with its message (train_engine.py:139-143); a non-finite
    loss -> sys.exit(1) (:166-169)."""
    from biapy_amd import train_engine as TE

    net, loss_fn = _toy_net(), torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    with pytest.raises(ValueError, match="different shape than 'DATA.PATCH_SIZE'"):
        TE.train_one_epoch(net, loss_fn, _toy_loader(2), opt, "cpu", 0, patch_size=(8, 6, 6, 1), graph="off")
    bad = _toy_loader(4)
    bad[1] = (bad[1][0] * float("nan"), bad[1][1])
    with pytest.raises(SystemExit) as e:
        TE.train_one_epoch(net, loss_fn, bad, opt, "cpu", 0, patch_size=(4, 6, 6, 1), graph="off", sync_every=2)
    assert e.value.code == 1
    with pytest.raises(ValueError, match="graph='on'"):
        TE.train_one_epoch(net, loss_fn, _toy_loader(1), opt, "cpu", 0, graph="on")


def test_chunk_grid_rank_partition_covers_every_chunk_once():
    """ChunkedPredictor's ownership rule (a sampler pad-repeat is left to the rank holding the chunk's first occurrence): over all
    ranks every chunk is written exactly once and the written regions tile the volume, for ragged grids and world sizes that do
    not divide the chunk count."""
    from biapy_amd.chunked import ChunkGrid

    for dim, crop, pad in (((33, 47, 129), (16, 32, 64), (3, 5, 10)), ((18, 33, 65), (16, 32, 64), (6, 12, 24)), ((40, 64, 64), (32, 32, 32), (0, 0, 0))):
        grid = ChunkGrid(dim, crop, pad)
        for world in (1, 2, 3, 7, 8, 16):
            owned = []
            for rank in range(world):
                owned += [v for k, v in enumerate(grid.rank_order(world, rank)) if rank + k * world < grid.total]
            assert sorted(owned) == list(range(grid.total)), (dim, world)
        cover = np.zeros(dim, np.int32)
        for v in range(grid.total):
            r = grid.region(v)
            cover[r[3]:r[3] + r[6], r[4]:r[4] + r[7], r[5]:r[5] + r[8]] += 1
            t = grid.index_tables(v)
            assert len(t) == sum(crop) and t.min() >= 0 and (t[:crop[0]] < dim[0]).all() and (t[crop[0]:crop[0] + crop[1]] < dim[1]).all()
        assert (cover == 1).all()


def test_lift_params_is_the_same_convolution():
    """engine.lift_params / unlift_grads (2D and (1,3,3) weights -> zero-padded 3x3x3): the lifted weights compute the same
    convolution on a one-slice / any volume, and gradients map back to the centre z-tap - checked with PyTorch's CPU convs."""
    import torch.nn.functional as F

    from biapy_amd.engine import lift_params, needs_lift, unlift_grads

    g = torch.Generator().manual_seed(0)
    P = {"c2d": torch.randn(5, 3, 3, 3, generator=g), "aniso": torch.randn(4, 3, 1, 3, 3, generator=g), "iso": torch.randn(4, 3, 3, 3, 3, generator=g),
         "ct2d": torch.randn(3, 5, 2, 2, generator=g), "k1": torch.randn(6, 3, 1, 1, generator=g), "bias": torch.randn(5, generator=g)}
    assert [needs_lift(v) for v in P.values()] == [True, True, False, True, True, False]
    Q = lift_params(P)
    assert Q["iso"] is P["iso"] and Q["bias"] is P["bias"] and Q["c2d"].shape == (5, 3, 3, 3, 3) and Q["ct2d"].shape == (3, 5, 1, 2, 2)
    x2 = torch.randn(2, 3, 9, 11, generator=g)
    assert torch.allclose(F.conv3d(x2.unsqueeze(2), Q["c2d"], padding=1)[:, :, 0], F.conv2d(x2, P["c2d"], padding=1), atol=1e-6)
    x3 = torch.randn(2, 3, 5, 9, 11, generator=g)
    assert torch.allclose(F.conv3d(x3, Q["aniso"], padding=1), F.conv3d(x3, P["aniso"], padding=(0, 1, 1)), atol=1e-6)
    assert torch.allclose(F.conv_transpose3d(x2.unsqueeze(2), Q["ct2d"], stride=(1, 2, 2))[:, :, 0], F.conv_transpose2d(x2, P["ct2d"], stride=2), atol=1e-6)
    G = {k: torch.randn(v.shape, generator=g) for k, v in Q.items()}
    U = unlift_grads(G, P)
    assert all(U[k].shape == P[k].shape for k in P)
    assert torch.equal(U["c2d"], G["c2d"][:, :, 1]) and torch.equal(U["aniso"][:, :, 0], G["aniso"][:, :, 1]) and torch.equal(U["k1"], G["k1"][:, :, 0])


@pytest.mark.parametrize("dim,d", [((10, 13, 16), 3), ((8, 8, 8), 6), ((5, 20, 7), 2)])
def test_dilated_conv_is_conv_on_sublattices(dim, d):
    """biapy_amd.dilation.lattice_tables: gathering the d^3 sub-lattices into the batch (zeros where a table says -1), an ordinary
    'same' 3x3x3 convolution and the inverse scatter reproduce torch's dilated convolution (padding = dilation) - the property the
    device path of ASPP (heads.py:77-104) rests on.  NumPy stands in for the two HIP kernels here."""
    import torch.nn.functional as F

    from biapy_amd.dilation import lattice_shape, lattice_tables

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, *dim, generator=g)
    w = torch.randn(4, 3, 3, 3, 3, generator=g)
    ref = F.conv3d(x, w, padding=d, dilation=d)
    T = lattice_tables(dim, d)
    nz, ny, nx = lattice_shape(dim, d)
    assert T.shape == (d ** 3, nz + ny + nx)
    xs = torch.zeros(2 * d ** 3, 3, nz, ny, nx)
    for n in range(2):
        for q in range(d ** 3):
            tz, ty, tx = T[q, :nz], T[q, nz:nz + ny], T[q, nz + ny:]
            sub = x[n][:, np.clip(tz, 0, None)][:, :, np.clip(ty, 0, None)][:, :, :, np.clip(tx, 0, None)].clone()
            sub[:, tz < 0] = 0
            sub[:, :, ty < 0] = 0
            sub[:, :, :, tx < 0] = 0
            xs[n * d ** 3 + q] = sub
    ys = F.conv3d(xs, w, padding=1)
    out = torch.zeros_like(ref)
    for n in range(2):
        for q in range(d ** 3):
            tz, ty, tx = T[q, :nz], T[q, nz:nz + ny], T[q, nz + ny:]
            kz, ky, kx = np.where(tz >= 0)[0], np.where(ty >= 0)[0], np.where(tx >= 0)[0]
            out[n][:, tz[kz][:, None, None], ty[ky][None, :, None], tx[kx][None, None, :]] = ys[n * d ** 3 + q][:, kz][:, :, ky][:, :, :, kx]
    assert torch.allclose(out, ref, atol=1e-5)
