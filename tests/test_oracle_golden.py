"""The oracle (oracle/*.py) against the golden vectors captured from the reference (CPU only)."""
import numpy as np
import pytest
import torch

from make_golden import synth_pred, synth_volume  # seeded input generators (pure numpy, no reference import)
from oracle import net_oracle, tiling_oracle as T

COORD_CASES = ["c48", "docstring", "docstring_noov", "cfg3_1024", "template_pad10", "aniso", "odd", "exactfit", "single"]
DATA_CASES = ["m48", "m_odd", "m_pad", "m_zeros", "m_median"]


@pytest.mark.parametrize("name", COORD_CASES)
def test_crop_coords_match_reference(tiling_golden, name):
    a = tiling_golden[f"coords/{name}/args"]
    ov = tuple(tiling_golden[f"coords/{name}/overlap"])
    got = T.crop_coords(a[0:3], a[4:7], ov, tuple(a[8:11]))
    np.testing.assert_array_equal(got, tiling_golden[f"coords/{name}/coords"])


def test_cfg3_grid_is_4096_patches(tiling_golden):
    c = tiling_golden["coords/cfg3_1024/coords"]
    assert c.shape == (4096, 6)
    assert sorted(set(c[:, 0]))[-2:] == [840, 896]


@pytest.mark.parametrize("name", ["w0", "w4", "w68", "wmix"])
def test_window_bit_exact(tiling_golden, name):
    a = tiling_golden[f"window/{name}/args"]
    got = T.spline_window(tuple(a[:3]), tuple(a[3:]))
    ref = tiling_golden[f"window/{name}/win"]
    assert got.dtype == np.float32 and got.shape == ref.shape
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name", DATA_CASES)
def test_crop_and_merge_bit_exact(tiling_golden, name):
    g = tiling_golden
    a = g[f"data/{name}/args"]
    vshape, pshape, pad, med = tuple(a[0:4]), tuple(a[4:8]), tuple(a[8:11]), bool(a[11])
    ov = tuple(g[f"data/{name}/overlap"])
    seed = int(g[f"data/{name}/seed"])
    pad_type = str(g[f"data/{name}/pad_type"])
    vol, mask = synth_volume(seed, vshape)
    p, coords = T.crop(vol, pshape, ov, pad, pad_type, med)
    pm, _ = T.crop(mask, pshape[:3] + (1,), ov, pad, pad_type, False)
    np.testing.assert_array_equal(coords, g[f"data/{name}/coords"])
    np.testing.assert_array_equal(p[0], g[f"data/{name}/patch0"])
    np.testing.assert_array_equal(p[-1], g[f"data/{name}/patch_last"])
    np.testing.assert_array_equal(pm[-1], g[f"data/{name}/mask_patch_last"])
    crc = int(np.frombuffer(p.tobytes(), dtype=np.uint8).astype(np.uint64).sum())
    assert crc == int(g[f"data/{name}/patches_crc"][0])
    pred = synth_pred(seed, p.shape)
    merged, merged_mask = T.merge(pred, vshape, data_mask=pm, overlap=ov, padding=pad)
    np.testing.assert_array_equal(merged.view(np.uint32), g[f"data/{name}/merged"].view(np.uint32))
    np.testing.assert_array_equal(merged_mask, g[f"data/{name}/merged_mask"])


@pytest.mark.parametrize("lab", [1, 2, 3, 5, 255])
def test_label_truncation_quirk(tiling_golden, lab):
    """Labels >=3 come back partly as label-1 through the float blend; the oracle must reproduce that."""
    pm = np.full((27, 32, 32, 32, 1), lab, dtype=np.uint8)
    pd = np.ones((27, 32, 32, 32, 1), dtype=np.float32)
    _, mm = T.merge(pd, (48, 48, 48, 1), data_mask=pm, overlap=(0.5, 0.5, 0.5))
    ref = tiling_golden[f"label/{lab}/merged_mask"]
    np.testing.assert_array_equal(mm, ref)
    if lab >= 3:
        assert (ref != lab).any()


def test_crop_errors():
    v = np.zeros((16, 16, 16, 1), np.float32)
    with pytest.raises(ValueError):
        T.crop(v[..., 0], (8, 8, 8, 1))
    with pytest.raises(ValueError):
        T.crop_coords((16, 16, 16), (8, 8, 8), overlap=(1.0, 0, 0))


def _small(resunet_golden):
    g = resunet_golden
    sd = {k[len("small/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("small/sd/")}
    x = torch.from_numpy(g["small/x"]).permute(0, 4, 1, 2, 3)
    tgt = torch.from_numpy(g["small/target"]).to(torch.float32)
    return sd, x, tgt, list(g["small/feature_maps"])


def test_net_oracle_forward_matches_reference(resunet_golden):
    sd, x, tgt, fm = _small(resunet_golden)
    logits = net_oracle.resunet_forward(sd, x, fm)
    ref = torch.from_numpy(resunet_golden["small/logits"])
    assert (logits - ref).abs().max().item() < 2e-5
    loss = net_oracle.bce_with_logits(logits, tgt)
    assert abs(loss.item() - float(resunet_golden["small/loss"])) < 1e-6


def test_net_oracle_grads_match_reference(resunet_golden):
    sd, x, tgt, fm = _small(resunet_golden)
    loss, logits, grads = net_oracle.train_step_grads(sd, x, tgt, feature_maps=fm)
    for k in resunet_golden.files:
        if k.startswith("small/gradnorm/"):
            name = k[len("small/gradnorm/"):]
            ref = float(resunet_golden[k])
            assert abs(grads[name].norm().item() - ref) <= 1e-4 * max(ref, 1e-6) + 1e-9, name
        if k.startswith("small/grad/"):
            name = k[len("small/grad/"):]
            ref = torch.from_numpy(resunet_golden[k])
            assert (grads[name] - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1e-6) + 1e-9, name


def test_state_dict_schema_cfg2(resunet_golden):
    """init_state_dict must produce exactly the reference's 98 keys / shapes for the cfg-2 architecture."""
    sd = net_oracle.init_state_dict(1, [16, 32, 64, 128, 256])
    keys = list(resunet_golden["cfg2/keys"])
    shapes = dict(zip(keys, resunet_golden["cfg2/shapes"]))
    assert set(sd.keys()) == set(keys)
    for k, v in sd.items():
        assert str(tuple(v.shape)) == shapes[k], k
    assert sum(v.numel() for v in sd.values()) == int(resunet_golden["cfg2/n_params"]) == 6693777


def test_flop_count_matches_baseline():
    assert net_oracle.count_flops_forward(1, [16, 32, 64, 128, 256], (128, 128, 128)) == 303734718464


def _case2d(g, name):
    a = g[f"{name}/args"]
    dshape, cshape, pad = tuple(int(v) for v in a[:4]), tuple(int(v) for v in a[4:7]), tuple(int(v) for v in a[7:9])
    seed = int(g[f"{name}/seed"])
    rs = np.random.RandomState(3000 + seed)
    data = rs.rand(*dshape).astype(np.float32)
    mask = np.array([0, 1, 2, 3, 7, 255], dtype=np.uint8)[rs.randint(0, 6, size=tuple(dshape[:3]) + (1,))]
    return data, mask, dshape, cshape, tuple(g[f"{name}/overlap"]), pad, str(g[f"{name}/pad_type"]), seed


@pytest.mark.parametrize("name", ["t256", "ov", "zeros", "fit"])
def test_tiling2d_oracle_matches_reference(tiling2d_golden, name):
    """2D crop / merge (data_2D_manipulation.py:54-533) restated through the z = image-index embedding: bit-exact."""
    from oracle import tiling_oracle as T

    g = tiling2d_golden
    data, mask, dshape, cshape, ov, pad, pad_type, seed = _case2d(g, name)
    p, cc = T.crop2d(data, cshape, ov, pad, pad_type)
    pm, _ = T.crop2d(mask, cshape[:2] + (1,), ov, pad, pad_type)
    np.testing.assert_array_equal(cc, g[f"{name}/coords"])
    assert int(np.frombuffer(p.tobytes(), dtype=np.uint8).astype(np.uint64).sum()) == int(g[f"{name}/patches_crc"][0])
    np.testing.assert_array_equal(p[-1], g[f"{name}/patch_last"])
    np.testing.assert_array_equal(pm[-1], g[f"{name}/mask_patch_last"])
    pred = np.random.RandomState(4000 + seed).rand(*p.shape).astype(np.float32)
    merged, merged_mask = T.merge2d(pred, dshape, pm, ov, pad)
    np.testing.assert_array_equal(merged.view(np.uint32), g[f"{name}/merged"].view(np.uint32))
    np.testing.assert_array_equal(merged_mask, g[f"{name}/merged_mask"])


def test_net_oracle_anisotropic_matches_reference(resunet_aniso_golden):
    """Z_DOWN = [1, 2] (anisotropic pooling / transposed conv): oracle logits, loss and gradients vs the reference fixture."""
    import torch

    from oracle import net_oracle

    g = resunet_aniso_golden
    fm, zd = [int(v) for v in g["feature_maps"]], [int(v) for v in g["z_down"]]
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32)) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).permute(0, 4, 1, 2, 3)
    tgt = torch.from_numpy(g["target"]).float()
    loss, logits, grads = net_oracle.train_step_grads(sd, x, tgt, feature_maps=fm, z_down=zd)
    assert (logits - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for k in g.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(g[k])
            assert (grads[k[5:]] - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k


def _prepost_volume(seed, shape):
    rs = np.random.RandomState(5000 + seed)
    v = rs.gamma(2.0, 30.0, size=shape).astype(np.float32) + rs.rand(*shape).astype(np.float32)
    v[rs.rand(*shape) < 0.001] = 4000.0
    return v


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_prepost_oracle_matches_reference(prepost_golden, name):
    """percentile_clip / zero_mean_unit_variance_normalization (biapy/data/norm.py) restated: same bounds, bits, moments."""
    from oracle import prepost_oracle as PO

    g = prepost_golden
    shape = tuple(int(v) for v in g[f"{name}/args"])
    lo, hi = (float(v) for v in g[f"{name}/pct"])
    v = _prepost_volume(int(g[f"{name}/seed"]), shape)
    clipped, x_lwr, x_upr = PO.percentile_clip(v.copy(), lo, hi)
    assert (x_lwr, x_upr) == tuple(float(b) for b in g[f"{name}/bounds"])
    assert int(np.frombuffer(clipped.tobytes(), dtype=np.uint8).astype(np.uint64).sum()) == int(g[f"{name}/clipped_crc"][0])
    normed, mean, std = PO.zero_mean_unit_variance_normalization(clipped.copy())
    assert (mean, std) == tuple(float(b) for b in g[f"{name}/mean_std"])
    np.testing.assert_array_equal(normed[shape[0] // 2, ::3, ::5], g[f"{name}/normed_slice"])


def test_tta_group_and_transforms_match_reference(tta_golden):
    """Orientation group (order included), apply and inverse of biapy/data/post_processing/tta.py: oracle and product."""
    from biapy_amd import tta as T
    from oracle import tta_oracle as TO

    g = tta_golden
    for ndim in (2, 3):
        for level in ("full", "flips", "none"):
            ref = g[f"group/{ndim}/{level}"]
            for impl in (TO.group(ndim, level), T.build_axis_transform_group(ndim, level)):
                got = np.array([list(p) + list(s) for p, s in impl], dtype=np.int64)
                np.testing.assert_array_equal(got, ref)
    for name, ndim in (("a3", 3), ("a2", 2)):
        arr = g[name]
        for n, (p, s) in enumerate(TO.group(ndim, "full")):
            np.testing.assert_array_equal(TO.apply(arr, p, s), g[f"apply/{name}/{n}"])
            ip, is_ = TO.inverse(p, s)
            np.testing.assert_array_equal(np.array(list(ip) + list(is_)), g[f"inverse/{name}/{n}"])
            np.testing.assert_array_equal(TO.apply(TO.apply(arr, p, s), ip, is_), g[f"roundtrip/{name}/{n}"])


def test_head_activations_oracle_matches_the_reference_method(head_acts_golden):
    """Row A: oracle.loss_oracle.apply_head_activations against ``Base_Workflow.apply_model_activations`` itself (inference and
    training, sigmoid / tanh / linear channels, one and two softmax groups, semantic "class" blocks): bit-exact."""
    from oracle import loss_oracle as LO

    g = head_acts_golden
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == 6
    for name in names:
        acts = [str(a) for a in g[f"{name}/acts"]]
        x = torch.from_numpy(g[f"{name}/logits"])
        for training, key in ((False, "infer"), (True, "train")):
            np.testing.assert_array_equal(LO.apply_head_activations(x, acts, training=training).numpy(), g[f"{name}/{key}"], err_msg=f"{name} {key}")


def test_oracle_pipeline_matches_the_reference_harness(harness_golden, resunet_golden):
    """the oracle's crop -> forward -> sigmoid -> merge (and its TTA) against the output of the reference's
    own ``Base_Workflow.process_test_sample`` (tests/golden/harness_golden.npz).  fp32 on both sides; the reference forwards the patches
    in mini-batches of TRAIN.BATCH_SIZE, the oracle all at once, so oneDNN may sum in a different order: 5e-6 absolute."""
    from oracle import net_oracle, tiling_oracle, tta_oracle

    h, g = harness_golden, resunet_golden
    sd = {k[len("small/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("small/sd/")}
    fm = [int(v) for v in g["small/feature_maps"]]
    patch = (32, 32, 32)

    def fwd(batch):                                          # (n, z, y, x, c) numpy -> probabilities, same layout
        with torch.no_grad():
            return torch.sigmoid(net_oracle.resunet_forward(sd, torch.from_numpy(np.ascontiguousarray(batch)).permute(0, 4, 1, 2, 3), fm)).permute(0, 2, 3, 4, 1).contiguous().numpy()

    q = h["plain/params"]
    ov, pad = tuple(q[:3]), tuple(int(v) for v in q[3:6])
    p, _ = tiling_oracle.crop(h["vol"], patch + (1,), ov, pad)
    assert int(h["plain/batches"]) == -(-p.shape[0] // int(q[6]))
    got = tiling_oracle.merge(fwd(p), h["vol"].shape, overlap=ov, padding=pad)
    assert np.abs(got - h["plain/pred"]).max() < 5e-6
    q = h["tta/params"]
    ov, pad, bs = tuple(q[:3]), tuple(int(v) for v in q[3:6]), int(q[6])
    p, _ = tiling_oracle.crop(h["tta/vol"], patch + (1,), ov, pad)
    for key, mode, level in (("tta/flips_mean", "mean", "flips"), ("tta/full_max", "max", "full")):
        pr = np.stack([tta_oracle.ensemble(p[k], fwd, 3, mode, level, bs) for k in range(p.shape[0])], 0)
        got = tiling_oracle.merge(pr, h["tta/vol"].shape, overlap=ov, padding=pad)
        assert np.abs(got - h[key]).max() < 5e-6, key


def test_oracle_harness_tail_matches_the_reference(harness_tail_golden, resunet_golden):
    """process_test_sample past the blend: pad_to_shape, the crop back to reflected_orig_shape and the class arg-max of the
    oracle against what the reference produced (harness_tail_golden.npz)."""
    from oracle import loss_oracle, net_oracle, tiling_oracle

    h, g = harness_tail_golden, resunet_golden
    sd = {k[len("small/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("small/sd/")}
    fm = [int(v) for v in g["small/feature_maps"]]
    patch = (32, 32, 32)
    # reflect
    padded = tiling_oracle.pad_to_shape(h["reflect/vol"], patch + (1,))
    np.testing.assert_array_equal(padded, h["reflect/padded"])
    q = h["reflect/params"]
    ov, pad = tuple(q[:3]), tuple(int(v) for v in q[3:6])
    p, _ = tiling_oracle.crop(padded, patch + (1,), ov, pad)
    with torch.no_grad():
        pr = torch.sigmoid(net_oracle.resunet_forward(sd, torch.from_numpy(np.ascontiguousarray(p)).permute(0, 4, 1, 2, 3), fm)).permute(0, 2, 3, 4, 1).contiguous().numpy()
    got = tiling_oracle.crop_to_reflected_orig_shape(tiling_oracle.merge(pr, padded.shape, overlap=ov, padding=pad), h["reflect/vol"].shape)
    assert got.shape == h["reflect/pred"].shape and np.abs(got - h["reflect/pred"]).max() < 5e-6
    # class block
    sd2 = dict(sd)
    sd2["heads.1.weight"], sd2["heads.1.bias"] = torch.from_numpy(h["class/heads.1.weight"]), torch.from_numpy(h["class/heads.1.bias"])
    q = h["class/params"]
    ov, pad = tuple(q[:3]), tuple(int(v) for v in q[3:6])
    p, _ = tiling_oracle.crop(h["class/vol"], patch + (1,), ov, pad)
    with torch.no_grad():
        lo = net_oracle.resunet_forward(sd2, torch.from_numpy(np.ascontiguousarray(p)).permute(0, 4, 1, 2, 3), fm, n_heads=2)
        pr = loss_oracle.apply_head_activations(lo, ["ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"], training=False).permute(0, 2, 3, 4, 1).contiguous().numpy()
    got = tiling_oracle.class_argmax(tiling_oracle.merge(pr, h["class/vol"].shape[:3] + (4,), overlap=ov, padding=pad), 3)
    ref = h["class/pred"]
    assert np.abs(got[..., 0] - ref[..., 0]).max() < 5e-6
    assert (got[..., 1] != ref[..., 1]).mean() < 1e-4          # arg-max of blended softmax values: ties within rounding are the only differences


def test_pixel_shuffle3d_extends_pixel_shuffle():
    """The 3-D pixel shuffle of the cfg-5 up-scaling stage is DEFINED by this package (the reference applies nn.PixelShuffle, a 2-D operator,
    and raises on 5-D tensors: rcan.py:317-319): check that the definition is nn.PixelShuffle's rule with one more axis - for every z
    sub-position a, the (b, e) sub-positions of a z-slice are F.pixel_shuffle of the channels that carry that a - and that it is a bijection."""
    import torch.nn.functional as F

    from oracle.rcan_oracle import pixel_shuffle3d

    s, C, Z, Y, X = 3, 2, 2, 4, 5
    x = torch.arange(C * s ** 3 * Z * Y * X, dtype=torch.float32).reshape(1, C * s ** 3, Z, Y, X)
    y = pixel_shuffle3d(x, s)
    assert y.shape == (1, C, Z * s, Y * s, X * s)
    assert torch.equal(torch.sort(y.flatten())[0], x.flatten())                       # a permutation of the elements
    xr = x.reshape(1, C, s, s * s, Z, Y, X)
    for a in range(s):
        for z in range(Z):
            plane = xr[:, :, a, :, z].reshape(1, C * s * s, Y, X)                    # channels c s^2 + b s + e of z sub-position a
            assert torch.equal(y[:, :, z * s + a], F.pixel_shuffle(plane, s))
    # the element the formula names
    n, c, z, yy, xx, a, b_, e = 0, 1, 1, 3, 2, 2, 0, 1
    assert y[n, c, s * z + a, s * yy + b_, s * xx + e] == x[n, c * s ** 3 + (a * s + b_) * s + e, z, yy, xx]


def test_tta_ensemble_oracle_matches_reference(tta_ensemble_golden):
    """The whole scalar-field TTA routine - pad to square (reflect / edge), predict every orientation, undo, mean / min / max, crop -
    against ``ensemble_predictions`` of the reference (post_processing.py:1386-1540) on five shapes x five settings: bit-exact."""
    from oracle import tta_oracle as TO

    g = tta_ensemble_golden
    for name, shape, ndim in TO.ENSEMBLE_CASES:
        img = g[f"{name}/img"]
        assert img.shape == shape
        for mode, level, bs in TO.ENSEMBLE_SETTINGS:
            np.testing.assert_array_equal(TO.ensemble(img, TO.standin_pred, ndim, mode, level, bs), g[f"{name}/{mode}/{level}/{bs}"],
                                          err_msg=f"{name} {mode} {level} {bs}")


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_unet_oracle_matches_reference(unet_golden, tag):
    """oracle/unet_oracle.py (plain U-Net, row U) vs the reference's U_Net outputs: logits, BCE loss and gradients."""
    import torch
    import torch.nn.functional as F

    from oracle import unet_oracle

    g = unet_golden
    fm, zd = [int(v) for v in g[f"{tag}/feature_maps"]], [int(v) for v in g[f"{tag}/z_down"]]
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(pre)}
    xl = torch.from_numpy(g[f"{tag}/x"])
    nd = xl.dim() - 2
    x = xl.permute(0, nd + 1, *range(1, nd + 1))
    logits = unet_oracle.unet_forward(sd, x, fm, z_down=zd)
    loss = F.binary_cross_entropy_with_logits(logits, torch.from_numpy(g[f"{tag}/target"]).float())
    loss.backward()
    assert (logits.detach() - torch.from_numpy(g[f"{tag}/logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g[f"{tag}/loss"])) < 1e-6
    pre = f"{tag}/grad/"
    for k in g.files:
        if k.startswith(pre):
            ref = torch.from_numpy(g[k])
            assert (sd[k[len(pre):]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k


@pytest.mark.parametrize("tag,shape", [("2d", (64, 64, 1)), ("3d", (16, 32, 32, 1))])
def test_unet_module_keeps_the_reference_state_dict(unet_golden, tag, shape):
    """biapy_amd.unet.U_Net owns its parameters under the reference's names and shapes (strict load of the reference's
    state_dict) and refuses to run without the GPU (no CPU fallback)."""
    import torch

    from biapy_amd.unet import U_Net

    g = unet_golden
    fm, zd = [int(v) for v in g[f"{tag}/feature_maps"]], [int(v) for v in g[f"{tag}/z_down"]]
    m = U_Net(image_shape=shape, activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", yx_down=[2] * (len(fm) - 1),
              z_down=zd, isotropy=[True] * len(fm), larger_io=False, conv_layers=[2] * len(fm))
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros((1, 1) + tuple(shape[:-1])))
    with pytest.raises(NotImplementedError):
        U_Net(image_shape=shape, feature_maps=fm, drop_values=[0.0] * len(fm), normalization="bn", larger_io=False)


@pytest.mark.parametrize("tag,shape", [("2d", (64, 64, 1)), ("anisok", (8, 32, 32, 1)), ("wide48", (8, 32, 32, 1))])
def test_resunet_variants_oracle_and_module(resunet_variants_golden, tag, shape):
    """2D ResUNet and anisotropic (1,3,3)-kernel levels: the oracle reproduces the reference's logits / loss / gradients, and
    biapy_amd.resunet.ResUNet owns parameters of exactly the reference's names and shapes (strict load)."""
    import torch
    import torch.nn.functional as F

    from biapy_amd.resunet import ResUNet
    from oracle import net_oracle

    g = resunet_variants_golden
    fm, zd = [int(v) for v in g[f"{tag}/feature_maps"]], [int(v) for v in g[f"{tag}/z_down"]]
    iso = [bool(v) for v in g[f"{tag}/isotropy"]]
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(pre)}
    xl = torch.from_numpy(g[f"{tag}/x"])
    nd = xl.dim() - 2
    x = xl.permute(0, nd + 1, *range(1, nd + 1))
    logits = net_oracle.resunet_forward(sd, x, fm, z_down=zd)
    loss = F.binary_cross_entropy_with_logits(logits, torch.from_numpy(g[f"{tag}/target"]).float())
    loss.backward()
    assert (logits.detach() - torch.from_numpy(g[f"{tag}/logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g[f"{tag}/loss"])) < 1e-6
    pre_g = f"{tag}/grad/"
    for k in g.files:
        if k.startswith(pre_g):
            ref = torch.from_numpy(g[k])
            assert (sd[k[len(pre_g):]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k
    m = ResUNet(image_shape=shape, activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", yx_down=[2] * (len(fm) - 1),
                z_down=zd, isotropy=iso, larger_io=False, conv_layers=[2] * len(fm))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)


ACTS = ["relu", "silu", "leaky_relu", "gelu", "tanh", "sigmoid", "softplus"]


@pytest.mark.parametrize("act", ACTS)
def test_resunet_block_activations_oracle_matches_reference(resunet_activations_golden, act):
    """Every block activation of the reference's get_activation (blocks.py:1973-1998) that the engine offers: the oracle graph reproduces the
    reference ResUNet's logits, loss and gradients built with that activation (fixture from tests/golden/make_golden.py resunet_activations),
    and the drop-in module accepts the name."""
    import torch
    import torch.nn.functional as F

    from biapy_amd.resunet import ResUNet
    from oracle import net_oracle

    g = resunet_activations_golden
    fm = [int(v) for v in g["feature_maps"]]
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).permute(0, 4, 1, 2, 3)
    logits = net_oracle.resunet_forward(sd, x, fm, activation=act)
    loss = F.binary_cross_entropy_with_logits(logits, torch.from_numpy(g["target"]).float())
    loss.backward()
    assert (logits.detach() - torch.from_numpy(g[f"{act}/logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g[f"{act}/loss"])) < 1e-6
    for k in g.files:
        if k.startswith(f"{act}/gradnorm/"):
            name = k[len(f"{act}/gradnorm/"):]
            ref = float(g[k])
            assert abs(sd[name].grad.norm().item() - ref) <= 1e-4 * ref + 1e-6, name
    m = ResUNet(image_shape=(16, 16, 16, 1), activation=act, feature_maps=fm, drop_values=[0.0] * 2, normalization="in", yx_down=[2], z_down=[2],
                isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2)
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)


def dropout_masks(g):
    """{block prefix: (p, keep mask (B, C, Z, Y, X) bool)} of resunet_dropout_golden.npz."""
    import torch
    out = {}
    for k in g.files:
        if k.startswith("mask/"):
            name = k[5:]
            shape = tuple(int(v) for v in g[f"mask_shape/{name}"])
            n = int(np.prod(shape))
            out[name] = (float(g[f"p/{name}"]), torch.from_numpy(np.unpackbits(g[k])[:n].reshape(shape).astype(bool)))
    return out


def test_resunet_dropout_oracle_matches_reference(resunet_dropout_golden):
    """MODEL.DROPOUT_VALUES > 0 (resunet.py:250, :270, :299 -> blocks.py:163): with the masks torch drew for the reference made explicit, the oracle
    reproduces the reference's training-mode logits, loss and gradients; without masks it is the evaluation-mode forward; and the drop-in
    module takes the option."""
    import torch
    import torch.nn.functional as F

    from biapy_amd.resunet import ResUNet
    from oracle import net_oracle

    g = resunet_dropout_golden
    fm = [int(v) for v in g["feature_maps"]]
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).permute(0, 4, 1, 2, 3)
    masks = dropout_masks(g)
    assert sorted(masks) == ["bottleneck", "down_path.0", "up_paths.0.0.conv_block"]
    logits = net_oracle.resunet_forward(sd, x, fm, dropout=masks)
    loss = F.binary_cross_entropy_with_logits(logits, torch.from_numpy(g["target"]).float())
    loss.backward()
    assert (logits.detach() - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for k in g.files:
        if k.startswith("gradnorm/"):
            ref = float(g[k])
            assert abs(sd[k[9:]].grad.norm().item() - ref) <= 1e-4 * ref + 1e-6, k
        if k.startswith("grad/"):
            ref = torch.from_numpy(g[k])
            assert (sd[k[5:]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k
    with torch.no_grad():
        ev = net_oracle.resunet_forward(sd, x, fm)
    assert (ev - torch.from_numpy(g["logits_eval"])).abs().max().item() < 2e-5
    assert (ev - logits.detach()).abs().max().item() > 1e-2                      # the masks matter
    m = ResUNet(image_shape=(16, 16, 16, 1), activation="elu", feature_maps=fm, drop_values=[float(v) for v in g["drop_values"]], normalization="in",
                yx_down=[2], z_down=[2], isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2)
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
    assert m.cfg.dropout == (0.1, 0.3)


def _chunked_case(g, tag):
    dim, crop, pad = tuple(int(v) for v in g[f"{tag}/dim"]), tuple(int(v) for v in g[f"{tag}/crop"]), tuple(int(v) for v in g[f"{tag}/padding"])
    vol = np.random.RandomState(int(g[f"{tag}/seed"])).randint(0, 256, size=dim + (1,)).astype(np.uint8)
    return dim, crop, pad, vol


def _wsum(p):
    return int((p.astype(np.int64).ravel() * (np.arange(p.size) % 977 + 1)).sum())


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_chunked_oracle_and_host_grid_match_reference(chunked_golden, tag):
    """By-chunks tiler: oracle/chunked_oracle.py AND the product's host arithmetic (biapy_amd.chunked.ChunkGrid: regions, reflect
    index tables, strip offsets) against the reference generator's own per-chunk outputs - read region, write-back region,
    padding added, and the padded patches (position-weighted checksums for all chunks, full arrays for case c)."""
    from biapy_amd.chunked import ChunkGrid
    from oracle import chunked_oracle as CO

    g = chunked_golden
    dim, crop, pad, vol = _chunked_case(g, tag)
    if f"{tag}/vol" in g.files:
        np.testing.assert_array_equal(vol, g[f"{tag}/vol"])
    grid = ChunkGrid(dim, crop, pad)
    assert grid.total == g[f"{tag}/extract"].shape[0]
    Pz, Py, Px = crop
    for vid in range(grid.total):
        q, ext, real = CO.patch_coords(vid, dim, crop, pad)
        assert [v for ab in ext for v in ab] == list(g[f"{tag}/extract"][vid]) and [v for ab in real for v in ab] == list(g[f"{tag}/real"][vid])
        patch, _, strip = CO.extract(vol, vid, crop, pad)
        assert strip == g[f"{tag}/added_pad"][vid].tolist()[:3]      # the reference appends [0, 0] for the channel axis
        assert int(patch.astype(np.int64).sum()) == int(g[f"{tag}/patch_sums"][vid]) and _wsum(patch) == int(g[f"{tag}/patch_wsums"][vid])
        # product host logic
        _, _, _, pe, pr = grid.patch_coords(vid)
        assert list(pe) == list(g[f"{tag}/extract"][vid]) and list(pr) == list(g[f"{tag}/real"][vid])
        t = grid.index_tables(vid)
        mine = vol[t[:Pz]][:, t[Pz:Pz + Py]][:, :, t[Pz + Py:]]
        np.testing.assert_array_equal(mine, patch)
        r = grid.region(vid)
        assert list(r[:3]) == [s[0] for s in strip] and list(r[3:6]) == [pr.z_start, pr.y_start, pr.x_start]
        if f"{tag}/patches" in g.files:
            np.testing.assert_array_equal(patch, g[f"{tag}/patches"][vid])


def test_chunked_rank_order_is_the_distributed_sampler():
    """ChunkGrid.rank_order == torch's DistributedSampler(shuffle=False) over the chunk ids (generator __iter__, :603-612)."""
    from torch.utils.data import DistributedSampler

    from biapy_amd.chunked import ChunkGrid

    grid = ChunkGrid((33, 47, 129), (16, 32, 64), (3, 5, 10))
    ids = list(range(grid.total))
    for world in (1, 2, 3, 5, 8, 40):
        for rank in range(world):
            ref = list(DistributedSampler(ids, num_replicas=world, rank=rank, shuffle=False))
            assert grid.rank_order(world, rank) == ref
    with pytest.raises(ValueError, match="Axis problem"):
        ChunkGrid((10, 47, 129), (16, 32, 64), (3, 5, 10))
    with pytest.raises(ValueError, match="can not be greater than half"):
        ChunkGrid((33, 47, 129), (16, 32, 64), (8, 5, 10))


def test_chunked_oracle_identity_round_trip(chunked_golden):
    """read -> identity prediction -> strip -> insert reproduces the volume (the chunks tile it exactly)."""
    from oracle import chunked_oracle as CO

    for tag in ("c", "d"):
        dim, crop, pad, vol = _chunked_case(chunked_golden, tag)
        out = CO.predict_by_chunks(vol, lambda p: p.astype(np.float32), crop, pad)
        np.testing.assert_array_equal(out, vol.astype(np.float32))


def test_rcan_oracle_and_module_match_reference(rcan_golden):
    """RCAN trunk (row S): oracle output / L1 loss / gradients vs the reference's ``rcan(ndim=3, upscaling_layer=False)``; the
    drop-in module owns the reference's parameter names and refuses what the path does not cover."""
    import torch

    from biapy_amd.rcan import rcan
    from oracle import rcan_oracle

    g = rcan_golden
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).permute(0, 4, 1, 2, 3)
    y = rcan_oracle.rcan_forward(sd, x, int(g["num_rg"]), int(g["num_rcab"]))
    loss = torch.nn.L1Loss()(y, torch.from_numpy(g["target"]))
    loss.backward()
    assert (y.detach() - torch.from_numpy(g["y"])).abs().max().item() < 2e-5 and abs(loss.item() - float(g["loss"])) < 1e-6
    for k in g.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(g[k])
            assert (sd[k[5:]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k
    m = rcan(ndim=3, num_channels=1, filters=16, scale=2, num_rg=2, num_rcab=2, reduction=16, upscaling_layer=False, out_channels=1, head_activations=["linear"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 1, 8, 8, 8))
    for bad in (dict(ndim=2), dict(upscaling_layer=True, filters=32), dict(upscaling_layer=True, scale=5), dict(filters=64), dict(num_channels=3)):
        kw = dict(ndim=3, num_channels=1, filters=16, num_rg=1, num_rcab=1, upscaling_layer=False)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            rcan(**kw)


def test_resunetpp_oracle_matches_reference(resunetpp_golden):
    """oracle/resunetpp_oracle.py (row X: SE, ASPP with dilated convs, attention, 3x3x3 shortcuts with norm) vs the reference's
    ResUNetPlusPlus: logits, MSE loss and gradients.  The device path for this model is the open row; the oracle is pinned ahead."""
    import torch

    from oracle import resunetpp_oracle as RO

    g = resunetpp_golden
    fm = [int(v) for v in g["feature_maps"]]
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32)).requires_grad_(True) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).permute(0, 4, 1, 2, 3)
    logits = RO.resunetpp_forward(sd, x, fm)
    loss = torch.nn.MSELoss()(logits, torch.from_numpy(g["target"].astype(np.float32)))
    loss.backward()
    assert (logits.detach() - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5 and abs(loss.item() - float(g["loss"])) < 1e-6
    for k in g.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(g[k])
            assert (sd[k[5:]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k
        if k.startswith("gradnorm/") and float(g[k]) > 1e-6:
            assert abs(sd[k[9:]].grad.norm().item() - float(g[k])) <= 1e-3 * float(g[k]), k


def test_loss_oracle_matches_the_reference_loss_classes():
    """oracle/loss_oracle.py against the reference's own CrossEntropyLoss_wrapper / DiceLoss / DiceCELoss /
    instance_segmentation_loss (biapy/engine/metrics.py:493-586, :726-762, :764-973, :1418-1810) on seeded inputs: value and the
    gradient w.r.t. the logits (tests/golden/losses_golden.npz, generated by importing the reference)."""
    import os

    from make_golden import loss_inputs
    from oracle import loss_oracle as LO

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses_golden.npz"))
    z1, t1, z3, t3 = loss_inputs()

    def check(name, fn, z):
        zz = z.clone().requires_grad_(True)
        val = fn(zz)
        val.backward()
        assert abs(val.item() - float(gold[f"{name}/value"])) < 1e-6, name
        assert np.abs(zz.grad.numpy() - gold[f"{name}/grad"]).max() < 1e-8 + 1e-5 * np.abs(gold[f"{name}/grad"]).max(), name

    check("bce", lambda z: LO.bce(z, t1), z1)
    check("dice", lambda z: LO.dice(z, t1), z1)
    check("dice_per_sample", lambda z: LO.dice(z, t1, batch_dice=False), z1)
    check("dice_ce_1_1", lambda z: LO.dice_ce(z, t1), z1)
    check("dice_ce_03_17", lambda z: LO.dice_ce(z, t1, 0.3, 1.7), z1)
    acts = ["ce_sigmoid", "ce_sigmoid", "tanh"]
    check("instance_bcd_mse", lambda z: LO.instance_channels(LO.apply_head_activations(z, acts), t3, ["bce", "bce", "mse"], (1, 1, 1)), z3)
    check("instance_bcd_l1_w", lambda z: LO.instance_channels(LO.apply_head_activations(z, acts), t3, ["bce", "bce", "l1"], (0.5, 0.25, 2.0)), z3)


def test_loss_oracle_matches_the_reference_multiclass_cross_entropy():
    """oracle/loss_oracle.softmax_ce against the reference's CrossEntropyLoss_wrapper with num_classes > 2 (metrics.py:493-586): plain, "manual" class
    weights, an ignore value, both, and a list of two predictions (tests/golden/losses_multiclass_golden.npz, generated by importing the reference)."""
    import os

    from make_golden import loss_inputs_multiclass
    from oracle import loss_oracle as LO

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses_multiclass_golden.npz"))
    z3, y3, z5, y5 = loss_inputs_multiclass()

    def check(name, fn, *zs):
        zz = [z.clone().requires_grad_(True) for z in zs]
        val = fn(*zz)
        val.backward()
        assert abs(val.item() - float(gold[f"{name}/value"])) < 2e-6, name
        for k, z in enumerate(zz):
            ref = gold[f"{name}/grad{k}"]
            assert np.abs(z.grad.numpy() - ref).max() < 1e-9 + 1e-5 * np.abs(ref).max(), (name, k)

    check("ce3", lambda z: LO.softmax_ce(z, y3), z3)
    check("ce3_w", lambda z: LO.softmax_ce(z, y3, [0.2, 0.5, 0.3]), z3)
    check("ce5_ignore", lambda z: LO.softmax_ce(z, y5, None, 255), z5)
    check("ce5_w_ignore", lambda z: LO.softmax_ce(z, y5, [1.0, 2.0, 0.5, 0.25, 4.0], 255), z5)
    check("ce3_deep", lambda a, b: LO.softmax_ce_deep([a, b], y3), z3, torch.from_numpy(gold["ce3_deep/zh"]))
    # the confusion counts: every kept voxel is counted once as a label and once as a prediction
    c = LO.confusion_counts(z5, y5, 255)
    kept = int((y5 != 255).sum())
    assert c.shape == (3, 5) and int(c[1].sum()) == kept and int(c[2].sum()) == kept and bool((c[0] <= torch.minimum(c[1], c[2])).all())


def test_resunetpp_module_keeps_the_reference_state_dict(resunetpp_golden):
    """biapy_amd.resunetpp.ResUNetPlusPlus (parameter holder of row X): same state_dict keys, order and shapes as the reference's
    ResUNetPlusPlus (the fixture's), strict loading works, and the cfg-4 architecture has the reference's 11,148,710 parameters
    (SURVEY.md 8a row X)."""
    import torch

    from biapy_amd.resunetpp import ResUNetPlusPlus

    g = resunetpp_golden
    fm = [int(v) for v in g["feature_maps"]]
    kw = dict(activation="elu", drop_values=[0.0] * 5, normalization="in", k_size=3, upsample_layer="convtranspose", yx_down=[2] * 4, z_down=[2] * 4,
              output_channels=[3], output_channel_info=["BCD"], head_activations=["ce_sigmoid", "ce_sigmoid", "tanh"], isotropy=[True] * 5,
              larger_io=False, conv_layers=[2] * 5)
    m = ResUNetPlusPlus(image_shape=(16, 32, 32, 1), feature_maps=fm, **kw)
    ref = [(k[3:], g[k].shape) for k in g.files if k.startswith("sd/")]
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in ref]
    m.load_state_dict({k[3:]: torch.from_numpy(g[k].astype(np.float32)) for k in g.files if k.startswith("sd/")}, strict=True)
    big = ResUNetPlusPlus(image_shape=(80, 80, 80, 1), feature_maps=[16, 32, 64, 128, 256], **kw)
    assert sum(p.numel() for p in big.parameters()) == 11148710
    with pytest.raises(NotImplementedError):
        ResUNetPlusPlus(image_shape=(64, 64, 1), feature_maps=fm, **kw)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 1, 16, 32, 32))


@pytest.mark.parametrize("tag,shape", [("pre122", (8, 16, 16, 1)), ("pre232", (4, 8, 16, 1)), ("post222", (8, 16, 16, 1)), ("post122", (8, 16, 16, 1))])
def test_resunet_sr_oracle_and_module(resunet_sr_golden, tag, shape):
    """Row S (3-D super-resolution through ResUNet.pre/post_upsampling, resunet.py:206-213, :326-333): the oracle reproduces the
    reference's output, L1 loss and gradients, and the drop-in owns the reference's parameters (names, order, shapes)."""
    import torch

    from biapy_amd.resunet import ResUNet
    from oracle import net_oracle

    g = resunet_sr_golden
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(pre)}
    x = torch.from_numpy(g[f"{tag}/x"]).permute(0, 4, 1, 2, 3)
    y = net_oracle.resunet_forward(sd, x, [16, 32])
    loss = torch.nn.L1Loss()(y, torch.from_numpy(g[f"{tag}/target"]))
    loss.backward()
    assert (y.detach() - torch.from_numpy(g[f"{tag}/out"])).abs().max().item() < 2e-5 and abs(loss.item() - float(g[f"{tag}/loss"])) < 1e-6
    for k in g.files:
        if k.startswith(f"{tag}/grad/"):
            ref = torch.from_numpy(g[k])
            assert (sd[k[len(tag) + 6:]].grad - ref).norm().item() <= 1e-4 * ref.norm().item() + 1e-7, k
    m = ResUNet(image_shape=shape, activation="elu", feature_maps=[16, 32], drop_values=[0.0] * 2, normalization="in", yx_down=[2], z_down=[2],
                isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2, head_activations=["linear"],
                upsampling_factor=tuple(int(v) for v in g[f"{tag}/factor"]), upsampling_position=str(g[f"{tag}/pos"]))
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)


def test_tta_spec_oracle_matches_the_reference_classes(tta_spec_golden):
    """Direction-carrying channels (flows, per-axis magnitudes, anisotropic offsets, rays, affinities): the oracle's restatement of the
    channel groups' supports / remap and of the spec-aware ensemble against the reference's own classes driven through its
    ensemble_predictions (make_golden.py tta_spec) - kept orientations and every ensembled prediction, bit-exact."""
    from oracle import tta_oracle as TO

    g = tta_spec_golden
    for name, shape, ndim, cout, groups in TO.spec_cases():
        img = g[f"{name}/img"]
        assert img.shape == shape
        for mode, level, bs in TO.SPEC_SETTINGS:
            kept = TO.filter_orientations(groups, TO.group(ndim, level))
            np.testing.assert_array_equal(np.array([list(p) + list(s) for p, s in kept]), g[f"{name}/kept/{level}"], err_msg=f"{name} {level}")
            got = TO.ensemble_spec(img, lambda b: TO.standin_pred_multi(b, cout), ndim, groups, mode, level, bs)
            np.testing.assert_array_equal(got, g[f"{name}/{mode}/{level}/{bs}"], err_msg=f"{name} {mode} {level} {bs}")
    # the spec the workflows get from build_tta_spec(["Gz", "Gv", "Gh", "B", "E_sigma_0..2"], 3): flows on (z, y, x) = channels (0, 1, 2), sigmas in
    # Cartesian order -> (z, y, x) = channels (6, 5, 4), unsigned; its groups are named "flow" / "E_sigma"
    assert list(g["from_names/kinds"]) == ["VectorChannels:flow", "VectorChannels:E_sigma", "ScalarChannels:scalar"]
    groups = [{"kind": "vector", "axis_channels": [0, 1, 2], "signed": True, "axis_scale": None},
              {"kind": "vector", "axis_channels": [6, 5, 4], "signed": False, "axis_scale": None}, {"kind": "scalar", "channels": [3]}]
    got = TO.ensemble_spec(g["from_names/img"], lambda b: TO.standin_pred_multi(b, 7), 3, groups, "max", "full", 4)
    np.testing.assert_array_equal(got, g["from_names/max/full/4"])
