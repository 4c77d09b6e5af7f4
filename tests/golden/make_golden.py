"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference, imported through _ref_shim).  The
fixtures are data: seeded inputs and the reference's outputs.  Re-run with
    python tests/golden/make_golden.py                 # every fixture that imports single reference modules through the namespace shim
    python tests/golden/make_golden.py harness         # harness_golden.npz: Base_Workflow.process_test_sample (imports the WHOLE package
    python tests/golden/make_golden.py head_acts       # head_acts_golden.npz  behind stand-ins for missing third-party modules: run alone)
    python tests/golden/make_golden.py build_model     # build_model_kwargs.json (same)
The GPU box never runs this file; tests read the committed .npz files.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim as shim  # noqa: E402

quiet = lambda: contextlib.redirect_stdout(io.StringIO())  # noqa: E731


def coords_array(coords):
    return np.array([[c.z_start, c.z_end, c.y_start, c.y_end, c.x_start, c.x_end] for c in coords], dtype=np.int64)


def zero_stride_volume(shape, dtype=np.uint8):
    return np.lib.stride_tricks.as_strided(np.zeros(1, dtype=dtype), shape=shape, strides=(0,) * len(shape))


def synth_volume(seed, vshape):
    """Seeded test volume + label mask; tests regenerate these instead of loading them."""
    rs = np.random.RandomState(1000 + seed)
    vol = rs.rand(*vshape).astype(np.float32)
    mask = np.array([0, 1, 2, 3, 7, 255], dtype=np.uint8)[rs.randint(0, 6, size=tuple(vshape[:3]) + (1,))]
    return vol, mask


def synth_pred(seed, pshape):
    """Seeded stand-in for the network output on every patch (not the identity, so the blend matters)."""
    return np.random.RandomState(2000 + seed).rand(*pshape).astype(np.float32)


def tiling_fixtures():
    d3 = shim.load("biapy.data.data_3D_manipulation")
    out = {}
    # ---- (1) coordinate-only cases (load_data=False) -------------------------------------------
    coord_cases = {
        "c48": ((48, 48, 48, 1), (32, 32, 32, 1), (0.5, 0.5, 0.5), (0, 0, 0)),
        "docstring": ((165, 768, 1024, 1), (80, 80, 80, 1), (0.5, 0.5, 0.5), (0, 0, 0)),
        "docstring_noov": ((165, 768, 1024, 1), (80, 80, 80, 1), (0, 0, 0), (0, 0, 0)),
        "cfg3_1024": ((1024, 1024, 1024, 1), (128, 128, 128, 1), (0.5, 0.5, 0.5), (0, 0, 0)),
        "template_pad10": ((100, 200, 180, 1), (80, 80, 80, 1), (0, 0, 0), (10, 10, 10)),
        "aniso": ((20, 128, 128, 1), (20, 64, 64, 1), (0, 0.25, 0.25), (4, 16, 16)),
        "odd": ((100, 90, 70, 2), (32, 40, 24, 2), (0.3, 0.1, 0.6), (2, 4, 3)),
        "exactfit": ((64, 64, 64, 1), (32, 32, 32, 1), (0, 0, 0), (0, 0, 0)),
        "single": ((32, 40, 48, 1), (32, 40, 48, 1), (0.5, 0.5, 0.5), (0, 0, 0)),
    }
    for name, (vshape, pshape, ov, pad) in coord_cases.items():
        with quiet():
            cc = d3.crop_3D_data_with_overlap(zero_stride_volume(vshape), pshape, overlap=ov, padding=pad, load_data=False)
        out[f"coords/{name}/args"] = np.array(list(vshape) + list(pshape) + list(pad), dtype=np.int64)
        out[f"coords/{name}/overlap"] = np.array(ov, dtype=np.float64)
        out[f"coords/{name}/coords"] = coords_array(cc)

    # ---- (2) spline windows ---------------------------------------------------------------------
    for name, (pshape, ovpx) in {
        "w0": ((8, 8, 8), (0, 0, 0)),
        "w4": ((16, 16, 16), (4, 4, 4)),
        "w68": ((128, 128, 128), (68, 68, 68)),
        "wmix": ((12, 20, 9), (3, 10, 7)),
    }.items():
        out[f"window/{name}/args"] = np.array(list(pshape) + list(ovpx), dtype=np.int64)
        out[f"window/{name}/win"] = d3._get_spline_window_3D(pshape, ovpx)

    # ---- (3) crop + merge data cases -------------------------------------------------------------
    # inputs are regenerated from legacy-RandomState seeds (bit-stable across NumPy versions) so the
    # fixture only has to carry the reference's OUTPUTS
    data_cases = {
        "m48": ((48, 48, 48, 1), (32, 32, 32, 1), (0.5, 0.5, 0.5), (0, 0, 0), "reflect", False),
        "m_odd": ((50, 45, 37, 2), (16, 20, 12, 2), (0.3, 0.1, 0.6), (2, 4, 3), "reflect", False),
        "m_pad": ((40, 60, 50, 1), (32, 32, 32, 1), (0, 0, 0), (6, 6, 6), "reflect", False),
        "m_zeros": ((40, 33, 50, 3), (24, 16, 32, 3), (0.25, 0.5, 0.0), (4, 2, 8), "zeros", False),
        "m_median": ((36, 36, 36, 1), (16, 16, 16, 1), (0.5, 0.5, 0.5), (2, 2, 2), "reflect", True),
    }
    for seed, (name, (vshape, pshape, ov, pad, pad_type, med)) in enumerate(data_cases.items()):
        vol, mask = synth_volume(seed, vshape)
        with quiet():
            p, pm, cc = d3.crop_3D_data_with_overlap(
                vol, pshape, data_mask=mask, overlap=ov, padding=pad, median_padding=med, pad_type=pad_type
            )
            # the merge consumes (a perturbed copy of) the patches so it is not just the identity
            pred = synth_pred(seed, p.shape)
            merged, merged_mask = d3.merge_3D_data_with_overlap(pred, vshape, data_mask=pm, overlap=ov, padding=pad)
        out[f"data/{name}/args"] = np.array(list(vshape) + list(pshape) + list(pad) + [int(med)], dtype=np.int64)
        out[f"data/{name}/overlap"] = np.array(ov, dtype=np.float64)
        out[f"data/{name}/pad_type"] = np.array(pad_type)
        out[f"data/{name}/seed"] = np.array(seed)
        out[f"data/{name}/coords"] = coords_array(cc)
        out[f"data/{name}/patches_crc"] = np.array([int(np.frombuffer(p.tobytes(), dtype=np.uint8).astype(np.uint64).sum())])
        out[f"data/{name}/patch0"] = p[0]
        out[f"data/{name}/patch_last"] = p[-1]
        out[f"data/{name}/mask_patch_last"] = pm[-1]
        out[f"data/{name}/merged"] = merged
        out[f"data/{name}/merged_mask"] = merged_mask

    # ---- (4) constant-label masks: pins the truncating uint8 cast ----------
    for lab in (1, 2, 3, 5, 255):
        pm = np.full((27, 32, 32, 32, 1), lab, dtype=np.uint8)
        pd = np.ones((27, 32, 32, 32, 1), dtype=np.float32)
        with quiet():
            _, mm = d3.merge_3D_data_with_overlap(pd, (48, 48, 48, 1), data_mask=pm, overlap=(0.5, 0.5, 0.5))
        out[f"label/{lab}/merged_mask"] = mm
    np.savez_compressed(os.path.join(HERE, "tiling_golden.npz"), **out)
    print("tiling_golden.npz:", len(out), "arrays")


def tiling2d_fixtures():
    """2D crop / merge (data_2D_manipulation.py:54-533): the reference's outputs on seeded inputs."""
    d2 = shim.load("biapy.data.data_2D_manipulation")
    out = {}
    cases = {
        "t256": ((2, 300, 280, 1), (256, 256, 1), (0.0, 0.0), (0, 0), "reflect"),
        "ov": ((3, 70, 90, 2), (32, 32, 2), (0.5, 0.25), (4, 2), "reflect"),
        "zeros": ((2, 64, 50, 1), (24, 40, 1), (0.3, 0.0), (3, 6), "zeros"),
        "fit": ((1, 48, 48, 1), (48, 48, 1), (0.5, 0.5), (0, 0), "reflect"),
    }
    for seed, (name, (dshape, cshape, ov, pad, pad_type)) in enumerate(cases.items()):
        rs = np.random.RandomState(3000 + seed)
        data = rs.rand(*dshape).astype(np.float32)
        mask = np.array([0, 1, 2, 3, 7, 255], dtype=np.uint8)[rs.randint(0, 6, size=tuple(dshape[:3]) + (1,))]
        with quiet():
            p, pm, cc = d2.crop_data_with_overlap(data, cshape, data_mask=mask, overlap=ov, padding=pad, pad_type=pad_type)
            pred = np.random.RandomState(4000 + seed).rand(*p.shape).astype(np.float32)
            merged, merged_mask = d2.merge_data_with_overlap(pred, dshape, data_mask=pm, overlap=ov, padding=pad)
        out[f"{name}/args"] = np.array(list(dshape) + list(cshape) + list(pad), dtype=np.int64)
        out[f"{name}/overlap"] = np.array(ov, dtype=np.float64)
        out[f"{name}/pad_type"] = np.array(pad_type)
        out[f"{name}/seed"] = np.array(seed)
        out[f"{name}/coords"] = np.array([[c.y_start, c.y_end, c.x_start, c.x_end] for c in cc], dtype=np.int64)
        out[f"{name}/patches_crc"] = np.array([int(np.frombuffer(p.tobytes(), dtype=np.uint8).astype(np.uint64).sum())])
        out[f"{name}/patch_last"] = p[-1]
        out[f"{name}/mask_patch_last"] = pm[-1]
        out[f"{name}/merged"] = merged
        out[f"{name}/merged_mask"] = merged_mask
    np.savez_compressed(os.path.join(HERE, "tiling2d_golden.npz"), **out)
    print("tiling2d_golden.npz:", len(out), "arrays")


def resunet_fixtures():
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    def build(fm, patch, in_ch=1, seed=0):
        torch.manual_seed(seed)
        depth = len(fm) - 1
        with quiet():
            net = rmod.ResUNet(
                image_shape=tuple(patch) + (in_ch,), activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm),
                normalization="in", k_size=3, upsample_layer="convtranspose", yx_down=[2] * depth, z_down=[2] * depth,
                output_channels=[1], output_channel_info=["F"], head_activations=["ce_sigmoid"], isotropy=[True] * len(fm),
                larger_io=False, conv_layers=[2] * len(fm),
            )
        # perturb norm affine + biases so the fixture exercises them (reference init is gamma=1, beta=0, bias=0)
        g = torch.Generator().manual_seed(seed + 100)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if v.ndim == 1:
                    v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
        return net

    out = {}
    # small architecture: full weights committed
    fm, patch, B = [16, 32, 64], (32, 32, 32), 2
    net = build(fm, patch)
    g = torch.Generator().manual_seed(0)
    x5 = torch.randn(B, *patch, 1, generator=g)                 # (B,Z,Y,X,C) as the data loader hands it over
    x = x5.permute(0, 4, 1, 2, 3)                               # to_pytorch_format (biapy/utils/misc.py:689-713)
    tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
    net.train()
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)            # metrics.py:543-544
    loss.backward()
    out["small/feature_maps"] = np.array(fm)
    out["small/x"] = x5.numpy()
    out["small/target"] = tgt.numpy().astype(np.uint8)
    out["small/logits"] = logits.detach().numpy()
    out["small/loss"] = np.array(loss.item(), dtype=np.float64)
    for k, v in net.state_dict().items():
        out[f"small/sd/{k}"] = v.numpy()
    for k, p in net.named_parameters():
        out[f"small/gradnorm/{k}"] = np.array(p.grad.norm().item(), dtype=np.float64)
    # a few full gradients (first conv, a decoder conv, a norm affine, the up-conv, the head)
    for k in [
        "down_path.0.block.0.block.0.weight", "down_path.1.block.0.weight", "down_path.1.block.0.bias",
        "up_paths.0.0.up.weight", "up_paths.0.1.conv_block.block.2.block.0.weight", "up_paths.0.1.conv_block.shortcut.0.weight",
        "heads.0.weight", "heads.0.bias", "bottleneck.block.3.block.0.bias",
    ]:
        out[f"small/grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
    # oracle restatement must agree with the reference right here
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    lo = net_oracle.resunet_forward(sd, x, fm)
    err = (lo - logits.detach()).abs().max().item()
    print("oracle vs reference (small) max abs err:", err)
    assert err < 2e-5

    # full cfg-2 architecture at 32^3: weights NOT committed (26.8 MB); pinned here and by hash
    fm2 = [16, 32, 64, 128, 256]
    net2 = build(fm2, (32, 32, 32), seed=1)
    net2.eval()
    with torch.no_grad():
        l2 = net2(x[:1])
        sd2 = {k: v.detach() for k, v in net2.state_dict().items()}
        lo2 = net_oracle.resunet_forward(sd2, x[:1], fm2)
    err2 = (lo2 - l2).abs().max().item()
    print("oracle vs reference (cfg2 arch) max abs err:", err2, "keys:", len(sd2), "params:", sum(v.numel() for v in sd2.values()))
    assert err2 < 2e-5
    out["cfg2/keys"] = np.array(list(sd2.keys()))
    out["cfg2/shapes"] = np.array([str(tuple(v.shape)) for v in sd2.values()])
    out["cfg2/n_params"] = np.array(sum(v.numel() for v in sd2.values()))
    out["cfg2/oracle_vs_ref_err"] = np.array(err2)
    np.savez_compressed(os.path.join(HERE, "resunet_golden.npz"), **out)
    print("resunet_golden.npz:", len(out), "arrays")


def tta_fixtures():
    """Orientation groups and transforms of biapy/data/post_processing/tta.py (the part of TTA that imports here)."""
    # the package __init__ pulls in post_processing.py (cv2, h5py, zarr, scikit-image: not installed); tta.py itself only needs
    # NumPy, so it is loaded from its file
    import importlib.util

    spec = importlib.util.spec_from_file_location("_ref_tta", os.path.join(shim.REF, "biapy", "data", "post_processing", "tta.py"))
    tt = importlib.util.module_from_spec(spec)
    sys.modules["_ref_tta"] = tt
    spec.loader.exec_module(tt)
    out = {}
    for ndim in (2, 3):
        for level in ("full", "flips", "none"):
            grp = tt.build_axis_transform_group(ndim, level=level)
            out[f"group/{ndim}/{level}"] = np.array([list(t.perm) + list(t.sign) for t in grp], dtype=np.int64)
    rs = np.random.RandomState(6000)
    a3 = rs.rand(3, 4, 5, 2).astype(np.float32)
    a2 = rs.rand(4, 6, 3).astype(np.float32)
    out["a3"], out["a2"] = a3, a2
    for name, arr, ndim in (("a3", a3, 3), ("a2", a2, 2)):
        for n, t in enumerate(tt.build_axis_transform_group(ndim, level="full")):
            out[f"apply/{name}/{n}"] = t.apply(arr)
            out[f"roundtrip/{name}/{n}"] = t.inverse.apply(t.apply(arr))
            out[f"inverse/{name}/{n}"] = np.array(list(t.inverse.perm) + list(t.inverse.sign), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "tta_golden.npz"), **out)
    print("tta_golden.npz:", len(out), "arrays")


def tta_ensemble_fixtures():
    """ensemble_predictions of biapy/data/post_processing/post_processing.py:1386-1540 (tta_spec=None) with the stand-in predictor of
    oracle/tta_oracle.py: padding to square (reflect / edge), every orientation predicted, undone, reduced (mean / min / max), cropped."""
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import tta_oracle as TO

    pp = shim.load_post_processing()
    out = {}
    rs = np.random.RandomState(6100)
    for name, shape, ndim in TO.ENSEMBLE_CASES:
        img = rs.rand(*shape).astype(np.float32)
        out[f"{name}/img"] = img
        back = (0, 2, 3, 1) if ndim == 2 else (0, 2, 3, 4, 1)          # tensor (B,C,spatial) -> numpy (B,spatial,C)
        fwd = (0, 3, 1, 2) if ndim == 2 else (0, 4, 1, 2, 3)

        def pred_func(batch):                                             # what model_call_func does: numpy batch in, channel-first tensor out
            return torch.from_numpy(TO.standin_pred(np.asarray(batch))).permute(*fwd)

        for mode, level, bs in TO.ENSEMBLE_SETTINGS:
            r = pp.ensemble_predictions(img, pred_func, back, fwd, torch.device("cpu"), ndim, batch_size_value=bs, mode=mode, group=level)
            out[f"{name}/{mode}/{level}/{bs}"] = r.permute(*back)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "tta_ensemble_golden.npz"), **out)
    print("tta_ensemble_golden.npz:", len(out), "arrays")


def tta_spec_fixtures():
    """ensemble_predictions with a TTASpec (direction-carrying channels: flows / per-axis magnitudes, rays, affinities): the reference's own channel
    groups (tta.py:318-540) built from the plain descriptions of oracle/tta_oracle.spec_cases(), driven through post_processing.ensemble_predictions
    with the exact-arithmetic stand-in predictor."""
    import importlib

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import tta_oracle as TO

    pp = shim.load_post_processing()
    tt = importlib.import_module("biapy.data.post_processing.tta")      # the module post_processing itself imported (isinstance checks)
    out = {}
    rs = np.random.RandomState(6200)
    for name, shape, ndim, cout, groups in TO.spec_cases():
        ref_groups = []
        for g in groups:
            if g["kind"] == "scalar":
                ref_groups.append(tt.ScalarChannels(channels=tuple(g["channels"])))
            elif g["kind"] == "vector":
                ref_groups.append(tt.VectorChannels(axis_channels=tuple(g["axis_channels"]), signed=g["signed"],
                                                    axis_scale=None if g["axis_scale"] is None else tuple(g["axis_scale"])))
            elif g["kind"] == "rays":
                ref_groups.append(tt.RayChannels(start=g["start"], dirs=np.asarray(g["dirs"])))
            else:
                ref_groups.append(tt.AffinityChannels(layout=dict(g["layout"])))
        spec = tt.TTASpec(ndim=ndim, n_channels=cout, groups=ref_groups)
        img = rs.rand(*shape).astype(np.float32)
        out[f"{name}/img"] = img
        back = (0, 2, 3, 1) if ndim == 2 else (0, 2, 3, 4, 1)
        fwd = (0, 3, 1, 2) if ndim == 2 else (0, 4, 1, 2, 3)

        def pred_func(batch, cout=cout, fwd=fwd):
            return torch.from_numpy(TO.standin_pred_multi(np.asarray(batch), cout)).permute(*fwd)

        for mode, level, bs in TO.SPEC_SETTINGS:
            r = pp.ensemble_predictions(img, pred_func, back, fwd, torch.device("cpu"), ndim, batch_size_value=bs, mode=mode, tta_spec=spec, group=level)
            out[f"{name}/{mode}/{level}/{bs}"] = r.permute(*back)[0].numpy()
            kept, _ = spec.filter_orientations(tt.build_axis_transform_group(ndim, level=level))
            out[f"{name}/kept/{level}"] = np.array([list(t.perm) + list(t.sign) for t in kept], dtype=np.int64)
    # the spec as the workflows build it: build_tta_spec from the channel names (its vector groups are NAMED after their family, "flow" / "E_sigma")
    names = ["Gz", "Gv", "Gh", "B", "E_sigma_0", "E_sigma_1", "E_sigma_2"]
    spec = tt.build_tta_spec(names, 3)
    out["from_names/kinds"] = np.array([type(g).__name__ + ":" + g.name for g in spec.groups])
    img = rs.rand(4, 6, 6, 1).astype(np.float32)
    out["from_names/img"] = img
    back, fwd = (0, 2, 3, 4, 1), (0, 4, 1, 2, 3)
    r = pp.ensemble_predictions(img, lambda b: torch.from_numpy(TO.standin_pred_multi(np.asarray(b), len(names))).permute(*fwd), back, fwd, torch.device("cpu"), 3,
                                batch_size_value=4, mode="max", tta_spec=spec, group="full")
    out["from_names/max/full/4"] = r.permute(*back)[0].numpy()
    # build_tta_spec itself: the structure of the spec for a set of channel-name lists (what biapy_amd.tta.build_tta_spec has to reproduce)
    import json

    def dump(spec):
        gs = []
        for g in spec.groups:
            d = {"cls": type(g).__name__, "name": g.name}
            if isinstance(g, tt.VectorChannels):
                d.update(axis_channels=list(g.axis_channels), signed=bool(g.signed), axis_scale=None if g.axis_scale is None else [float(v) for v in g.axis_scale])
            elif isinstance(g, tt.RayChannels):
                d.update(start=int(g.start), dirs=np.asarray(g.dirs, dtype=np.float64).tolist())
            elif isinstance(g, tt.AffinityChannels):
                d.update(layout=[[int(a), int(o), int(c)] for (a, o), c in sorted(g.layout.items())])
            else:
                d.update(channels=[int(c) for c in g.channels])
            gs.append(d)
        return {"ndim": spec.ndim, "n_channels": spec.n_channels, "groups": gs}

    TO_NAMES = TO.BUILD_SPEC_CASES
    for k, (cnames, nd, extra, aniso) in enumerate(TO_NAMES):
        out[f"build/{k}"] = np.array(json.dumps(dump(tt.build_tta_spec(cnames, nd, extra, aniso))))
    out["parse/0"] = np.array(json.dumps(tt.parse_model_output_channel_names(["Gv+Gh+B", "class"])))
    np.savez_compressed(os.path.join(HERE, "tta_spec_golden.npz"), **out)
    print("tta_spec_golden.npz:", len(out), "arrays")


def harness_fixtures():
    """Rows P / B / F / A of SURVEY 8a pinned to the reference harness ITSELF: ``Base_Workflow.process_test_sample`` (per-patch branch,
    base_workflow.py:1874-2131) is run unbound on a stand-in ``self`` that carries only the attributes the branch reads (a namespace cfg
    with the TEST / DATA / TRAIN fields, the reference ResUNet with the weights of resunet_golden.npz on the CPU, axes orders); it calls the
    reference's own crop_3D_data_with_overlap -> predict_batches_in_test -> model_call_func -> apply_model_activations ->
    merge_3D_data_with_overlap (and ensemble_predictions with TEST.AUGMENTATION).  The run is stopped after the raw prediction has been
    recorded (``return_prediction``), before the post-processing that is out of scope."""
    import contextlib
    import io
    import types
    from types import SimpleNamespace as NS

    import torch

    bw = shim.load_full_reference()
    from biapy.models.resunet import ResUNet

    g = np.load(os.path.join(HERE, "resunet_golden.npz"))
    sd = {k[len("small/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("small/sd/")}
    fm = [int(v) for v in g["small/feature_maps"]]
    with contextlib.redirect_stdout(io.StringIO()):
        model = ResUNet(image_shape=(32, 32, 32, 1), activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", k_size=3,
                        upsample_layer="convtranspose", yx_down=[2] * (len(fm) - 1), z_down=[2] * (len(fm) - 1), output_channels=[1],
                        output_channel_info=["F"], head_activations=["ce_sigmoid"], isotropy=[True] * len(fm), larger_io=False, conv_layers=[2] * len(fm))
    model.load_state_dict(sd, strict=True)
    model.eval()

    def run(vol, ov, pad, bs, aug=False, group="auto", mode="mean"):
        cfg = NS(TEST=NS(FULL_IMG=False, REUSE_PREDICTIONS=False, VERBOSE=False, REDUCE_MEMORY=False, AUGMENTATION=aug, AUGMENTATION_MODE=mode,
                         AUGMENTATION_GROUP=group, SAVE_MODEL_RAW_OUTPUT=False),
                 PROBLEM=NS(NDIM="3D", TYPE="SEMANTIC_SEG", SELF_SUPERVISED=NS(PRETEXT_TASK="")),
                 DATA=NS(PATCH_SIZE=(32, 32, 32, 1), TEST=NS(OVERLAP=ov, PADDING=pad, MEDIAN_PADDING=False), PREPROCESS=NS(TEST=False),
                         REFLECT_TO_COMPLETE_SHAPE=False),
                 TRAIN=NS(BATCH_SIZE=bs), MODEL=NS(SOURCE="biapy"), LOSS=NS(CONTRAST=NS(ENABLE=False)), PATHS=NS(RESULT_DIR=NS(PER_IMAGE="")))
        s = NS(cfg=cfg, model=model, device=torch.device("cpu"), test_device=torch.device("cpu"), axes_order=(0, 4, 1, 2, 3), axes_order_back=(0, 2, 3, 4, 1),
               dtype=np.float32, stats={"per_crop": {}, "merge_patches": {}, "patch_by_batch_counter": 0}, apply_activations=True,
               head_activations=["ce_sigmoid"], model_output_channels=[1], model_output_channel_info=["F"], padding_type="reflect",
               separated_class_channel=False, return_prediction=True, _predictions=[], dims=3, save_to_disk=False)
        B = bw.Base_Workflow
        for name in ("model_call_func", "apply_model_activations", "predict_batches_in_test"):
            setattr(s, name, types.MethodType(getattr(B, name), s))
        s.apply_roi_mask = lambda p: p
        s._log_tta_once = lambda: False
        s.current_sample = {"X": vol[None].copy(), "Y": None, "X_filename": "v.tif"}
        try:
            with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()):
                B.process_test_sample(s)
        except AttributeError as e:              # the first attribute of the post-processing part that the stand-in does not carry
            assert s._predictions, e
        return s._predictions[0]["data"][0], s.stats["patch_by_batch_counter"]

    out = {}
    rs = np.random.RandomState(2)
    vol = rs.randn(48, 40, 56, 1).astype(np.float32)
    out["vol"] = vol
    out["plain/params"] = np.array([0.5, 0.25, 0.5, 0, 4, 0, 5], dtype=np.float64)        # overlap zyx, padding zyx, batch
    out["plain/pred"], n = run(vol, (0.5, 0.25, 0.5), (0, 4, 0), 5)
    out["plain/batches"] = np.array(n)
    small = vol[:40, :32, :48].copy()                                                         # TTA: 16 forwards per patch on the CPU
    out["tta/vol"] = small
    out["tta/params"] = np.array([0.25, 0.0, 0.5, 0, 0, 0, 3], dtype=np.float64)
    out["tta/flips_mean"], _ = run(small, (0.25, 0.0, 0.5), (0, 0, 0), 3, aug=True, group="flips", mode="mean")
    out["tta/full_max"], _ = run(small, (0.25, 0.0, 0.5), (0, 0, 0), 3, aug=True, group="auto", mode="max")
    np.savez_compressed(os.path.join(HERE, "harness_golden.npz"), **out)
    print("harness_golden.npz:", {k: v.shape for k, v in out.items()})


def harness_tail_fixtures():
    """The steps of ``Base_Workflow.process_test_sample`` PAST the blended prediction, from the reference itself:
      * reflect: DATA.REFLECT_TO_COMPLETE_SHAPE - the test generator pads axes shorter than the patch in front with ``pad_to_shape``
        (data_manipulation.py:3218-3300), the workflow crops the prediction back to ``reflected_orig_shape`` (base_workflow.py:2089-2131);
      * class: ``separated_class_channel`` - the model returns {"pred", "class"}, the harness concatenates them per batch (:1677) and turns
        the trailing class block of the BLENDED volume into one arg-max channel (:2135-2141)."""
    import contextlib
    import io
    import types
    from types import SimpleNamespace as NS

    import torch

    bw = shim.load_full_reference()
    from biapy.data.data_manipulation import pad_to_shape
    from biapy.models.resunet import ResUNet

    g = np.load(os.path.join(HERE, "resunet_golden.npz"))
    sd = {k[len("small/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("small/sd/")}
    fm = [int(v) for v in g["small/feature_maps"]]

    def build(out_channels):
        with contextlib.redirect_stdout(io.StringIO()):
            m = ResUNet(image_shape=(32, 32, 32, 1), activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", k_size=3,
                        upsample_layer="convtranspose", yx_down=[2] * (len(fm) - 1), z_down=[2] * (len(fm) - 1), output_channels=out_channels,
                        output_channel_info=["F", "Db"][: len(out_channels)], head_activations=["ce_sigmoid"] * sum(out_channels), isotropy=[True] * len(fm),
                        larger_io=False, conv_layers=[2] * len(fm))
        return m

    def run(model, vol, ov, pad, bs, reflect_shape=None, class_block=0, acts=("ce_sigmoid",), ch=(1,), info=("F",)):
        cfg = NS(TEST=NS(FULL_IMG=False, REUSE_PREDICTIONS=False, VERBOSE=False, REDUCE_MEMORY=False, AUGMENTATION=False, AUGMENTATION_MODE="mean",
                         AUGMENTATION_GROUP="auto", SAVE_MODEL_RAW_OUTPUT=False),
                 PROBLEM=NS(NDIM="3D", TYPE="SEMANTIC_SEG" if not class_block else "INSTANCE_SEG", SELF_SUPERVISED=NS(PRETEXT_TASK="")),
                 DATA=NS(PATCH_SIZE=(32, 32, 32, 1), TEST=NS(OVERLAP=ov, PADDING=pad, MEDIAN_PADDING=False), PREPROCESS=NS(TEST=False),
                         REFLECT_TO_COMPLETE_SHAPE=reflect_shape is not None),
                 TRAIN=NS(BATCH_SIZE=bs), MODEL=NS(SOURCE="biapy"), LOSS=NS(CONTRAST=NS(ENABLE=False)), PATHS=NS(RESULT_DIR=NS(PER_IMAGE="")))
        s = NS(cfg=cfg, model=model, device=torch.device("cpu"), test_device=torch.device("cpu"), axes_order=(0, 4, 1, 2, 3), axes_order_back=(0, 2, 3, 4, 1),
               dtype=np.float32, stats={"per_crop": {}, "merge_patches": {}, "patch_by_batch_counter": 0}, apply_activations=True,
               head_activations=list(acts), model_output_channels=list(ch), model_output_channel_info=list(info), padding_type="reflect",
               separated_class_channel=bool(class_block), return_prediction=True, _predictions=[], dims=3, save_to_disk=False)
        B = bw.Base_Workflow
        for name in ("model_call_func", "apply_model_activations", "predict_batches_in_test"):
            setattr(s, name, types.MethodType(getattr(B, name), s))
        s.apply_roi_mask = lambda p: p
        s._log_tta_once = lambda: False
        s.current_sample = {"X": vol[None].copy(), "Y": None, "X_filename": "v.tif"}
        if reflect_shape is not None:
            s.current_sample["reflected_orig_shape"] = tuple(reflect_shape)
        try:
            with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
                B.process_test_sample(s)
        except AttributeError as e:
            assert s._predictions, e
        return s._predictions[0]["data"][0]

    out = {}
    rs = np.random.RandomState(12)
    # ---- reflect: z and x shorter than the 32^3 patch, y longer -----------------------------------------------------------------
    m1 = build([1]); m1.load_state_dict(sd, strict=True); m1.eval()
    vol = rs.randn(20, 44, 25, 1).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        padded = pad_to_shape(vol, (32, 32, 32, 1))
    out["reflect/vol"], out["reflect/padded"] = vol, padded
    out["reflect/params"] = np.array([0.0, 0.5, 0.0, 0, 0, 0, 3], dtype=np.float64)
    out["reflect/pred"] = run(m1, padded, (0.0, 0.5, 0.0), (0, 0, 0), 3, reflect_shape=vol.shape)
    assert out["reflect/pred"].shape == vol.shape, out["reflect/pred"].shape
    # ---- class: a second head of 3 channels is the class block ---------------------------------------------------------------------
    m2 = build([1, 3])
    sd2 = dict(sd)
    gen = torch.Generator().manual_seed(5)
    sd2["heads.1.weight"] = 0.5 * torch.randn(m2.state_dict()["heads.1.weight"].shape, generator=gen)
    sd2["heads.1.bias"] = 0.1 * torch.randn(3, generator=gen)
    m2.load_state_dict(sd2, strict=True); m2.eval()

    class Split(torch.nn.Module):          # what a reference model with a separated class head returns
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, x):
            y = self.net(x)
            y = y["pred"] if isinstance(y, dict) else y
            return {"pred": y[:, :1], "class": y[:, 1:]}

    vol2 = rs.randn(40, 32, 48, 1).astype(np.float32)
    out["class/vol"] = vol2
    out["class/heads.1.weight"], out["class/heads.1.bias"] = sd2["heads.1.weight"].numpy(), sd2["heads.1.bias"].numpy()
    out["class/params"] = np.array([0.25, 0.0, 0.5, 0, 0, 0, 4], dtype=np.float64)
    out["class/pred"] = run(Split(m2), vol2, (0.25, 0.0, 0.5), (0, 0, 0), 4, class_block=3, acts=("ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"),
                            ch=(1, 3), info=("F", "class"))
    assert out["class/pred"].shape == vol2.shape[:3] + (2,), out["class/pred"].shape
    np.savez_compressed(os.path.join(HERE, "harness_tail_golden.npz"), **out)
    print("harness_tail_golden.npz:", {k: v.shape for k, v in out.items()})


HEAD_ACT_CASES = [  # (name, PROBLEM.TYPE, model_output_channels, model_output_channel_info, head_activations - one per channel)
    ("sem1", "SEMANTIC_SEG", [1], ["F"], ["ce_sigmoid"]),
    ("sem3", "SEMANTIC_SEG", [3], ["F"], ["ce_softmax", "ce_softmax", "ce_softmax"]),
    ("bcd", "INSTANCE_SEG", [1, 1, 1], ["B", "C", "D"], ["ce_sigmoid", "ce_sigmoid", "tanh"]),
    ("f_db", "INSTANCE_SEG", [1, 3], ["F", "Db"], ["ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"]),
    ("lin", "INSTANCE_SEG", [1, 1], ["F", "D"], ["linear", "ce_sigmoid"]),
    ("two_groups", "INSTANCE_SEG", [2, 1, 2], ["Db", "D", "Dc"], ["ce_softmax", "ce_softmax", "tanh", "ce_softmax", "ce_softmax"]),
]


def build_model_kwargs_fixture():
    """SURVEY 8b, model registry: the keyword arguments ``biapy.models.build_model`` (models/__init__.py:84-179) passes to the class it finds
    under the fixed name in ``biapy.models.<architecture>`` - recorded by putting a recorder class under that name, exactly where the
    maintainer patch of INTEGRATION.md puts the MI355X class - for cfg 2 (resunet), cfg 4 (resunet++) and the super-resolution form."""
    import importlib
    import json
    from types import SimpleNamespace as NS

    import torch

    shim.load_full_reference()
    import biapy.models as M

    rec = {}

    def recorder(tag):
        class Rec(torch.nn.Module):
            def __init__(self, **kw):
                super().__init__()
                rec[tag[0]] = kw

        return Rec

    tag = [""]
    rmod = importlib.import_module("biapy.models.resunet")
    ref_resunet = rmod.ResUNet                       # the reference class itself: its parameter count goes into the record
    rmod.ResUNet = recorder(tag)
    importlib.import_module("biapy.models.resunet++").ResUNetPlusPlus = recorder(tag)

    def cfg(arch, ptype, patch, fm=(16, 32, 64, 128, 256), zd=None, drop=None):
        n = len(fm)
        return NS(MODEL=NS(ARCHITECTURE=arch, ACTIVATION="ELU", FEATURE_MAPS=list(fm), DROPOUT_VALUES=list(drop) if drop is not None else [0.0] * n, NORMALIZATION="in",
                           KERNEL_SIZE=3, UPSAMPLE_LAYER="convtranspose", YX_DOWN=[2] * (n - 1), Z_DOWN=list(zd) if zd is not None else [2] * (n - 1),
                           ISOTROPY=[True] * n, LARGER_IO=False, CONV_LAYERS=[2] * n,
                           CONV_BLOCK_ORDER="conv_norm_act", UNET_SR_UPSAMPLE_POSITION="pre", SOURCE="biapy"),
                  PROBLEM=NS(NDIM="3D", TYPE=ptype, IMAGE_TO_IMAGE=NS(SEPARATED_DECODERS_PER_HEAD=False), INSTANCE_SEG=NS(SEPARATED_DECODERS_PER_HEAD=False),
                             DETECTION=NS(SEPARATED_DECODERS_PER_HEAD=False), SUPER_RESOLUTION=NS(UPSCALING=(2, 2, 2))),
                  DATA=NS(PATCH_SIZE=patch), LOSS=NS(CONTRAST=NS(ENABLE=False, PROJ_DIM=256)))

    bcd = ["ce_sigmoid", "ce_sigmoid", "tanh"]
    for name, c, oc, oi, ha in (("cfg2_resunet", cfg("resunet", "SEMANTIC_SEG", (128, 128, 128, 1)), [1], ["F"], ["ce_sigmoid"]),
                                ("cfg4_resunet++", cfg("resunet++", "INSTANCE_SEG", (80, 80, 80, 1)), [3], ["BCD"], bcd),
                                ("sr_resunet", cfg("resunet", "SUPER_RESOLUTION", (64, 64, 64, 1)), [1], ["F"], ["linear"]),
                                # the MODEL / DATA.PATCH_SIZE lines of two of the reference's own 3-D ResUNet templates (round 4: widths beyond powers of two)
                                # templates/instance_segmentation/Ovarian_Reserve_paper/ovarian_reserve_training.yaml (DATA_CHANNELS BCD)
                                ("ovarian_reserve_resunet", cfg("resunet", "INSTANCE_SEG", (40, 128, 128, 1), fm=(48, 64, 80, 96), zd=(1, 1, 1)), [3], ["BCD"], bcd),
                                # templates/instance_segmentation/CartoCell_paper/cartocell_training_latest.yaml (DATA_CHANNELS BCM)
                                ("cartocell_resunet", cfg("resunet", "INSTANCE_SEG", (80, 80, 80, 1), fm=(52, 68, 84), zd=(1, 1), drop=(0.1, 0.1, 0.1)), [3], ["BCM"],
                                 ["ce_sigmoid"] * 3)):
        tag[0] = name
        try:
            M.build_model(c, oc, oi, ha, torch.device("cpu"))
        except AttributeError:
            pass                                     # the stand-in cfg ends where build_model turns to its model summary
        assert name in rec, name
    rmod.ResUNet = ref_resunet                       # back in place: the class refers to itself by its module-level name
    with quiet():
        rec["_reference_parameter_counts"] = {name: sum(p.numel() for p in ref_resunet(**rec[name]).parameters())
                                              for name in ("cfg2_resunet", "ovarian_reserve_resunet", "cartocell_resunet")}
    with open(os.path.join(HERE, "build_model_kwargs.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("build_model_kwargs.json:", {k: len(v) for k, v in rec.items()}, rec["_reference_parameter_counts"])


def head_acts_fixtures():
    """Row A: ``Base_Workflow.apply_model_activations`` (base_workflow.py:1367-1470) run unbound on a stand-in self, inference and training."""
    import types
    from types import SimpleNamespace as NS

    import torch

    bw = shim.load_full_reference()
    out = {}
    g = torch.Generator().manual_seed(6200)
    for name, ptype, chans, infos, acts in HEAD_ACT_CASES:
        C = sum(chans)
        x = torch.randn(2, C, 4, 5, 6, generator=g) * 3
        out[f"{name}/logits"] = x.numpy()
        out[f"{name}/acts"] = np.array(acts)
        s = NS(apply_activations=True, cfg=NS(PROBLEM=NS(TYPE=ptype, SELF_SUPERVISED=NS(PRETEXT_TASK=""))), model_output_channel_info=infos,
               model_output_channels=chans, head_activations=acts)
        f = types.MethodType(bw.Base_Workflow.apply_model_activations, s)
        out[f"{name}/infer"] = f(x.clone(), training=False).numpy()
        out[f"{name}/train"] = f(x.clone(), training=True).numpy()
    np.savez_compressed(os.path.join(HERE, "head_acts_golden.npz"), **out)
    print("head_acts_golden.npz:", len(out), "arrays")


def synth_prepost(seed, shape):
    """Seeded intensity volume with a heavy tail (so that percentile clipping matters) - tests regenerate it."""
    rs = np.random.RandomState(5000 + seed)
    v = rs.gamma(2.0, 30.0, size=shape).astype(np.float32) + rs.rand(*shape).astype(np.float32)
    v[rs.rand(*shape) < 0.001] = 4000.0
    return v


def prepost_fixtures():
    """biapy/data/norm.py outputs (percentile_clip, zero_mean_unit_variance_normalization) on seeded volumes."""
    nm = shim.load("biapy.data.norm")
    out = {}
    for seed, (name, shape, lo, hi) in enumerate([("a", (20, 33, 47), 0.2, 99.8), ("b", (64, 64, 64), 1.0, 99.0), ("c", (7, 130, 131), 5.0, 50.0)]):
        v = synth_prepost(seed, shape)
        clipped, x_lwr, x_upr = nm.percentile_clip(v.copy(), lo, hi)
        normed, mean, std = nm.zero_mean_unit_variance_normalization(clipped.copy())
        out[f"{name}/args"] = np.array(list(shape), dtype=np.int64)
        out[f"{name}/pct"] = np.array([lo, hi], dtype=np.float64)
        out[f"{name}/seed"] = np.array(seed)
        out[f"{name}/bounds"] = np.array([x_lwr, x_upr], dtype=np.float64)
        out[f"{name}/mean_std"] = np.array([mean, std], dtype=np.float64)
        out[f"{name}/clipped_crc"] = np.array([int(np.frombuffer(clipped.tobytes(), dtype=np.uint8).astype(np.uint64).sum())])
        out[f"{name}/normed_slice"] = normed[shape[0] // 2, ::3, ::5].copy()
    np.savez_compressed(os.path.join(HERE, "prepost_golden.npz"), **out)
    print("prepost_golden.npz:", len(out), "arrays")


def resunet_aniso_fixtures():
    """Anisotropic ResUNet (MODEL.Z_DOWN = [1, 2]: pooling / transposed conv (1,2,2) at the first level): reference logits,
    loss and every gradient of a small net (fm 16-32-64, 8x32x32 patches)."""
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    fm, patch, B, zd = [16, 32, 64], (8, 32, 32), 2, [1, 2]
    torch.manual_seed(5)
    with quiet():
        net = rmod.ResUNet(
            image_shape=tuple(patch) + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * 3, normalization="in", k_size=3,
            upsample_layer="convtranspose", yx_down=[2, 2], z_down=zd, output_channels=[1], output_channel_info=["F"],
            head_activations=["ce_sigmoid"], isotropy=[True] * 3, larger_io=False, conv_layers=[2] * 3,
        )
    g = torch.Generator().manual_seed(105)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if v.ndim == 1:
                v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
    g = torch.Generator().manual_seed(6)
    x5 = torch.randn(B, *patch, 1, generator=g)
    x = x5.permute(0, 4, 1, 2, 3)
    tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
    net.train()
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    out = {"feature_maps": np.array(fm), "z_down": np.array(zd), "x": x5.numpy(), "target": tgt.numpy().astype(np.uint8),
           "logits": logits.detach().numpy(), "loss": np.array(loss.item(), dtype=np.float64)}
    for k, v in net.state_dict().items():
        out[f"sd/{k}"] = v.numpy().astype(np.float16) if v.ndim == 5 and v.numel() > 20000 else v.numpy()
    # the big conv weights are stored as fp16 to keep the fixture small: re-run the reference ON THE ROUNDED weights so that
    # the stored outputs belong to exactly the stored parameters
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(torch.from_numpy(out[f"sd/{k}"].astype(np.float32)))
    net.zero_grad()
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    for k, p in net.named_parameters():
        out[f"gradnorm/{k}"] = np.array(p.grad.norm().item(), dtype=np.float64)
    for k in ["down_path.0.block.0.block.0.weight", "up_paths.0.0.up.weight", "up_paths.0.1.up.weight", "up_paths.0.1.up.bias",
              "down_path.1.block.0.weight", "heads.0.weight"]:
        out[f"grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    lo = net_oracle.resunet_forward(sd, x, fm, z_down=zd)
    err = (lo - logits.detach()).abs().max().item()
    print("oracle vs reference (anisotropic) max abs err:", err)
    assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "resunet_aniso_golden.npz"), **out)
    print("resunet_aniso_golden.npz:", len(out), "arrays")


def unet_fixtures():
    """Plain U-Net (row U / cfg 1 family): reference ``U_Net`` logits, BCE loss and every gradient norm of a 2D net
    (fm 16-32-64, 64x64 patches, B=2) and a 3D net (fm 16-32, 16x32x32, z_down [1])."""
    umod = shim.load("biapy.models.unet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import unet_oracle

    out = {}
    for tag, fm, patch, zd, seed in (("2d", [16, 32, 64], (64, 64), [2, 2], 11), ("3d", [16, 32], (16, 32, 32), [1], 12)):
        depth = len(fm) - 1
        torch.manual_seed(seed)
        with quiet():
            net = umod.U_Net(
                image_shape=tuple(patch) + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", k_size=3,
                upsample_layer="convtranspose", yx_down=[2] * depth, z_down=zd, output_channels=[1], output_channel_info=["F"],
                head_activations=["ce_sigmoid"], isotropy=[True] * len(fm), larger_io=False, conv_layers=[2] * len(fm),
            )
        g = torch.Generator().manual_seed(100 + seed)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if v.ndim == 1:                       # move the IN affine / biases off their init so their gradients are exercised
                    v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
        B = 2
        xl = torch.randn(B, *patch, 1, generator=g)                              # channels-last, as the data pipeline holds it
        x = xl.permute(0, len(patch) + 1, *range(1, len(patch) + 1))
        tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
        net.train()
        logits = net(x)
        loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
        loss.backward()
        out[f"{tag}/feature_maps"] = np.array(fm)
        out[f"{tag}/z_down"] = np.array(zd)
        out[f"{tag}/x"] = xl.numpy()
        out[f"{tag}/target"] = tgt.numpy().astype(np.uint8)
        out[f"{tag}/logits"] = logits.detach().numpy()
        out[f"{tag}/loss"] = np.array(loss.item(), dtype=np.float64)
        for k, v in net.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()
        for k, p_ in net.named_parameters():
            out[f"{tag}/gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        names = dict(net.named_parameters())
        for k in ["down_path.0.block.0.block.0.weight", "down_path.0.block.1.block.1.weight", "up_paths.0.0.up.0.weight",
                  "up_paths.0.0.up.1.bias", "heads.0.weight"]:
            out[f"{tag}/grad/{k}"] = names[k].grad.numpy()
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        lo = unet_oracle.unet_forward(sd, x, fm, z_down=zd)
        err = (lo - logits.detach()).abs().max().item()
        print(f"unet {tag}: params {sum(p_.numel() for p_ in net.parameters())}, oracle vs reference max abs err {err:.3e}")
        assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "unet_golden.npz"), **out)
    print("unet_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "unet_golden.npz")) // 1024, "KiB")


def resunet_variants_fixtures():
    """ResUNet configurations that run through zero-padded 3x3x3 kernels: a 2D network (fm 16-32-64, 64x64) and a 3D one with
    anisotropic levels (MODEL.ISOTROPY = [False, False, True] -> (1,3,3) kernels, Z_DOWN = [1, 2]); and (round 4) one with
    feature maps 48-64 (the head reads 48 features).  Reference logits, loss, gradient norms and a few full gradients."""
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    out = {}
    for tag, fm, patch, zd, iso, seed in (("2d", [16, 32, 64], (64, 64), [2, 2], [True] * 3, 31),
                                          ("anisok", [16, 32, 64], (8, 32, 32), [1, 2], [False, False, True], 32),
                                          # round 4: the first widths of the reference's Ovarian-Reserve template (FEATURE_MAPS [48, 64, 80, 96], Z_DOWN 1):
                                          # a head fed by 48 features, a 112-channel concatenation (two levels keep the fixture small)
                                          ("wide48", [48, 64], (8, 32, 32), [1], [True] * 2, 33)):
        depth = len(fm) - 1
        torch.manual_seed(seed)
        with quiet():
            net = rmod.ResUNet(
                image_shape=tuple(patch) + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * len(fm), normalization="in", k_size=3,
                upsample_layer="convtranspose", yx_down=[2] * depth, z_down=zd, output_channels=[1], output_channel_info=["F"],
                head_activations=["ce_sigmoid"], isotropy=iso, larger_io=False, conv_layers=[2] * len(fm),
            )
        g = torch.Generator().manual_seed(100 + seed)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if v.ndim == 1:
                    v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
        B = 2
        xl = torch.randn(B, *patch, 1, generator=g)
        x = xl.permute(0, len(patch) + 1, *range(1, len(patch) + 1))
        tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
        net.train()
        logits = net(x)
        loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
        loss.backward()
        out[f"{tag}/feature_maps"], out[f"{tag}/z_down"], out[f"{tag}/isotropy"] = np.array(fm), np.array(zd), np.array(iso)
        out[f"{tag}/x"], out[f"{tag}/target"] = xl.numpy(), tgt.numpy().astype(np.uint8)
        out[f"{tag}/logits"], out[f"{tag}/loss"] = logits.detach().numpy(), np.array(loss.item(), dtype=np.float64)
        for k, v in net.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()
        names = dict(net.named_parameters())
        for k, p_ in names.items():
            out[f"{tag}/gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        for k in ["down_path.0.block.0.block.0.weight", "down_path.1.block.2.block.0.weight", "up_paths.0.0.up.weight",
                  "up_paths.0.1.conv_block.shortcut.0.weight", "heads.0.weight", "heads.0.bias", "bottleneck.block.2.block.0.weight",
                  "up_paths.0.0.conv_block.shortcut.0.weight"]:
            if k in names and (tag == "wide48" or k in ("down_path.0.block.0.block.0.weight", "down_path.1.block.2.block.0.weight", "up_paths.0.0.up.weight",
                                                        "up_paths.0.1.conv_block.shortcut.0.weight", "heads.0.weight")):
                out[f"{tag}/grad/{k}"] = names[k].grad.numpy()
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        lo = net_oracle.resunet_forward(sd, x, fm, z_down=zd)
        err = (lo - logits.detach()).abs().max().item()
        print(f"resunet {tag}: kernel shapes {sorted(set(tuple(v.shape[2:]) for v in sd.values() if v.ndim >= 4))}, oracle vs reference {err:.3e}")
        assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "resunet_variants_golden.npz"), **out)
    print("resunet_variants_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_variants_golden.npz")) // 1024, "KiB")


def resunet_activations_fixtures():
    """The reference ResUNet with each block activation of get_activation (blocks.py:1973-1998) that the MI355X engine offers beyond ELU - relu,
    silu, leaky_relu, gelu, tanh, sigmoid, softplus - on a 2-level 3D network: reference logits, loss and all gradient norms.  Pins
    oracle/net_oracle.py's activation switch (and through it the kernels' run-time-activation instances) to the reference classes."""
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    out = {}
    fm, patch = [16, 32], (16, 16, 16)
    for i, act in enumerate(["relu", "silu", "leaky_relu", "gelu", "tanh", "sigmoid", "softplus"]):
        torch.manual_seed(60)                         # the same weights and the same batch for every activation: stored once
        with quiet():
            net = rmod.ResUNet(image_shape=patch + (1,), activation=act, feature_maps=fm, drop_values=[0.0] * 2, normalization="in", k_size=3,
                               upsample_layer="convtranspose", yx_down=[2], z_down=[2], output_channels=[1], output_channel_info=["F"],
                               head_activations=["ce_sigmoid"], isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2)
        g = torch.Generator().manual_seed(160)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if v.ndim == 1:
                    v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
        B = 2
        xl = torch.randn(B, *patch, 1, generator=g)
        x = xl.permute(0, 4, 1, 2, 3)
        tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
        net.train()
        logits = net(x)
        loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
        loss.backward()
        if i == 0:
            out["x"], out["target"] = xl.numpy(), tgt.numpy().astype(np.uint8)
            for k, v in net.state_dict().items():
                out[f"sd/{k}"] = v.numpy()
        else:
            assert all(np.array_equal(out[f"sd/{k}"], v.numpy()) for k, v in net.state_dict().items()) and np.array_equal(out["x"], xl.numpy())
        out[f"{act}/logits"], out[f"{act}/loss"] = logits.detach().numpy(), np.array(loss.item(), dtype=np.float64)
        for k, p_ in net.named_parameters():
            out[f"{act}/gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        for k in ["down_path.0.block.0.block.0.weight", "down_path.0.block.1.block.0.weight", "up_paths.0.0.conv_block.shortcut.0.weight", "heads.0.weight"]:
            out[f"{act}/grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        lo = net_oracle.resunet_forward(sd, x, fm, activation=act)
        err = (lo - logits.detach()).abs().max().item()
        print(f"resunet activation {act}: oracle vs reference {err:.3e}")
        assert err < 2e-5
    out["feature_maps"] = np.array(fm)
    np.savez_compressed(os.path.join(HERE, "resunet_activations_golden.npz"), **out)
    print("resunet_activations_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_activations_golden.npz")) // 1024, "KiB")


def resunet_explicit_tail_fixtures():
    """how the reference builds and applies its explicit activation lists (prepare_activation_layers, blocks.py:2001-2051; forward tail
    resunet.py:413-425).  Three channel layouts, each as (raw head outputs with explicit_activations=False, outputs with it True) of the SAME weights:
    a multi-class head (output_channels [3], ce_softmax x3: ONE collected activation -> joint softmax over the 3 channels), a mask + distance-like
    pair (["F", "Db"], sigmoid + 3x softmax: the list stops after the first softmax, so channel 1 is a softmax over one channel and channels 2, 3
    stay raw) and a class head whose first channel activation is a softmax (the pred list ends up with one entry)."""
    rmod = shim.load("biapy.models.resunet")
    out = {}
    fm, patch = [16, 32], (8, 8, 8)
    cases = {
        "multiclass": dict(output_channels=[3], output_channel_info=["F"], head_activations=["ce_softmax"] * 3),
        "mask_then_softmax": dict(output_channels=[1, 3], output_channel_info=["F", "Db"], head_activations=["ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"]),
        "two_then_class": dict(output_channels=[2, 2], output_channel_info=["F", "class"], head_activations=["tanh", "ce_sigmoid", "ce_softmax", "ce_softmax"]),
    }
    g = torch.Generator().manual_seed(172)
    xl = torch.randn(2, *patch, 1, generator=g)
    out["x"] = xl.numpy()
    for tag, kw in cases.items():
        nets = []
        for explicit in (False, True):
            torch.manual_seed(72)
            with quiet():
                nets.append(rmod.ResUNet(image_shape=patch + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * 2, normalization="in", k_size=3,
                                         upsample_layer="convtranspose", yx_down=[2], z_down=[2], explicit_activations=explicit, isotropy=[True] * 2,
                                         larger_io=False, conv_layers=[2] * 2, **kw).eval())
        nets[1].load_state_dict(nets[0].state_dict())
        with torch.no_grad():
            raw, act = nets[0](xl.permute(0, 4, 1, 2, 3)), nets[1](xl.permute(0, 4, 1, 2, 3))
        for name, o in (("raw", raw), ("act", act)):
            if isinstance(o, dict):
                for k, v in o.items():
                    out[f"{tag}/{name}/{k}"] = v.numpy()
            else:
                out[f"{tag}/{name}/pred"] = o.numpy()
        out[f"{tag}/kwargs"] = np.array(json.dumps(kw))
    np.savez_compressed(os.path.join(HERE, "resunet_explicit_tail_golden.npz"), **out)
    print("resunet_explicit_tail_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_explicit_tail_golden.npz")) // 1024, "KiB")


def resunet_class_head_fixtures():
    """The reference ResUNet with a classification head (output_channels [1, 3], output_channel_info ["F", "class"], resunet.py:180, :408-443): the
    out dict without and with explicit_activations (sigmoid on the mask channel, softmax over the class block), a loss over both heads
    (BCE on pred + cross-entropy on class) and all gradient norms - what the drop-in's output tail and head kernel must reproduce."""
    rmod = shim.load("biapy.models.resunet")
    out = {}
    fm, patch = [16, 32], (16, 16, 16)
    for tag, explicit in (("logits", False), ("explicit", True)):
        torch.manual_seed(71)
        with quiet():
            net = rmod.ResUNet(image_shape=patch + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * 2, normalization="in", k_size=3,
                               upsample_layer="convtranspose", yx_down=[2], z_down=[2], output_channels=[1, 3], output_channel_info=["F", "class"],
                               explicit_activations=explicit, head_activations=["ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"],
                               isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2)
        g = torch.Generator().manual_seed(171)
        B = 2
        xl = torch.randn(B, *patch, 1, generator=g)
        tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
        cls_t = torch.randint(0, 3, (B,) + patch, generator=g)
        net.train()
        o = net(xl.permute(0, 4, 1, 2, 3))
        assert isinstance(o, dict) and set(o) == {"pred", "class"}
        if explicit:
            loss = torch.nn.functional.binary_cross_entropy(o["pred"], tgt) + torch.nn.functional.nll_loss(torch.log(o["class"] + 1e-12), cls_t)
        else:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(o["pred"], tgt) + torch.nn.functional.cross_entropy(o["class"], cls_t)
        loss.backward()
        if tag == "logits":
            out["x"], out["target"], out["class_target"] = xl.numpy(), tgt.numpy().astype(np.uint8), cls_t.numpy().astype(np.uint8)
            for k, v in net.state_dict().items():
                out[f"sd/{k}"] = v.detach().numpy()
        out[f"{tag}/pred"], out[f"{tag}/class"], out[f"{tag}/loss"] = o["pred"].detach().numpy(), o["class"].detach().numpy(), np.array(loss.item(), dtype=np.float64)
        for k, p_ in net.named_parameters():
            out[f"{tag}/gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        for k in ["heads.0.weight", "heads.1.weight", "heads.1.bias", "down_path.0.block.1.block.0.weight"]:
            out[f"{tag}/grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
        net.eval()
        with quiet():
            net1 = rmod.ResUNet(image_shape=patch + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * 2, normalization="in", k_size=3,
                                upsample_layer="convtranspose", yx_down=[2], z_down=[2], output_channels=[1, 3], output_channel_info=["F", "class"],
                                explicit_activations=explicit, head_activations=["ce_sigmoid", "ce_softmax", "ce_softmax", "ce_softmax"],
                                isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2, return_one_tensor=True)
        net1.load_state_dict(net.state_dict())
        with torch.no_grad():
            out[f"{tag}/one_tensor"] = net1(xl.permute(0, 4, 1, 2, 3)).numpy()
    out["feature_maps"] = np.array(fm)
    np.savez_compressed(os.path.join(HERE, "resunet_class_head_golden.npz"), **out)
    print("resunet_class_head_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_class_head_golden.npz")) // 1024, "KiB")


def resunet_dropout_fixtures():
    """The reference ResUNet in TRAINING mode with drop_values > 0 (resunet.py:250, :270, :299 -> blocks.py:163 nn.Dropout after Conv -> Norm -> Act of
    every block's first ConvBlock): the masks torch drew are captured by forward hooks on the nn.Dropout modules (keep = output != 0 where the
    input != 0), so that the oracle - and the device, through its explicit-mask mode - can be held to the reference's logits, loss and gradients
    with the random draw taken out of the comparison."""
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    out = {}
    fm, patch, drops = [16, 32], (16, 16, 16), [0.1, 0.3]
    torch.manual_seed(83)
    with quiet():
        net = rmod.ResUNet(image_shape=patch + (1,), activation="elu", feature_maps=fm, drop_values=drops, normalization="in", k_size=3,
                           upsample_layer="convtranspose", yx_down=[2], z_down=[2], output_channels=[1], output_channel_info=["F"],
                           head_activations=["ce_sigmoid"], isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2)
    g = torch.Generator().manual_seed(183)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if v.ndim == 1:
                v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
    B = 2
    xl = torch.randn(B, *patch, 1, generator=g)
    x = xl.permute(0, 4, 1, 2, 3)
    tgt = (torch.rand(B, 1, *patch, generator=g) > 0.5).to(torch.float32)
    masks, ps = {}, {}
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Dropout):
            prefix = name.split(".block.")[0]           # "down_path.0" / "bottleneck" / "up_paths.0.1.conv_block"

            def hook(mod, inp, outp, prefix=prefix):
                if not mod.training:
                    return
                assert prefix not in masks, (prefix, sorted(masks))
                masks[prefix] = (outp != 0) | (inp[0] == 0)
                ps[prefix] = float(mod.p)
            m.register_forward_hook(hook)
    net.train()
    torch.manual_seed(283)
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    assert sorted(masks) == sorted(["down_path.0", "bottleneck", "up_paths.0.0.conv_block"]), sorted(masks)
    out["x"], out["target"] = xl.numpy(), tgt.numpy().astype(np.uint8)
    for k, v in net.state_dict().items():
        out[f"sd/{k}"] = v.detach().numpy()
    for k, mk in masks.items():
        out[f"mask/{k}"] = np.packbits(mk.numpy().astype(np.uint8).ravel())
        out[f"mask_shape/{k}"] = np.array(mk.shape)
        out[f"p/{k}"] = np.array(ps[k], dtype=np.float64)
        print(f"dropout site {k}: p = {ps[k]}, kept {mk.float().mean().item():.4f}")
    out["logits"], out["loss"] = logits.detach().numpy(), np.array(loss.item(), dtype=np.float64)
    for k, p_ in net.named_parameters():
        out[f"gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
    for k in ["down_path.0.block.0.block.0.weight", "down_path.0.block.1.block.0.weight", "bottleneck.block.3.block.0.weight", "up_paths.0.0.conv_block.block.2.block.0.weight",
              "bottleneck.block.2.block.1.weight", "heads.0.weight"]:
        out[f"grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    lo = net_oracle.resunet_forward(sd, x, fm, dropout={k: (ps[k], masks[k]) for k in masks})
    err = (lo - logits.detach()).abs().max().item()
    print(f"resunet dropout: oracle (explicit masks) vs reference {err:.3e}")
    assert err < 2e-5
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = net(x).numpy()
    out["feature_maps"], out["drop_values"] = np.array(fm), np.array(drops)
    np.savez_compressed(os.path.join(HERE, "resunet_dropout_golden.npz"), **out)
    print("resunet_dropout_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_dropout_golden.npz")) // 1024, "KiB")


def chunked_fixtures():
    """By-chunks tiler: the reference generator's own ``_patch_coords`` / ``extract_and_prepare_sample`` on seeded uint8
    volumes (the class is loaded with the third-party modules it never calls on this path stubbed; a bare object carrying the
    attributes its ``__init__`` derives (:272-289) stands in for a Zarr-backed instance)."""
    import importlib
    import math
    import types

    shim.install()

    class _Any(types.ModuleType):
        __path__ = []

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            sub = _Any(self.__name__ + "." + k)
            sys.modules[sub.__name__] = sub
            return sub

        def __call__(self, *a, **k):
            return self

    g = types.ModuleType("biapy.data.generators")
    g.__path__ = [os.path.join(shim.REF, "biapy/data/generators")]
    sys.modules["biapy.data.generators"] = g
    mod = None
    for _ in range(40):
        try:
            mod = importlib.import_module("biapy.data.generators.chunked_test_pair_data_generator")
            break
        except ModuleNotFoundError as e:
            if e.name.startswith("biapy"):
                raise
            sys.modules[e.name] = _Any(e.name)
    cls = mod.chunked_test_pair_data_generator
    out = {}
    # d: padding wider than the last, clipped chunk -> np.pad reflects more than once about the edges of the clipped region
    cases = [("a", (70, 90, 100), (32, 32, 32), (4, 8, 2), False, 11), ("b", (40, 64, 64), (32, 32, 32), (0, 0, 0), False, 12),
             ("c", (33, 47, 129), (16, 32, 64), (3, 5, 10), True, 13), ("d", (18, 33, 65), (16, 32, 64), (6, 12, 24), False, 14)]
    for tag, dim, crop, pad, keep, seed in cases:
        rs = np.random.RandomState(seed)
        vol = rs.randint(0, 256, size=dim + (1,)).astype(np.uint8)
        me = types.SimpleNamespace()
        me.X_parallel_data, me.Y_parallel_data = vol, None
        me.input_axes = me.mask_input_axes = "ZYXC"
        me.crop_shape, me.padding = crop + (1,), pad
        me.z_dim, me.y_dim, me.x_dim = dim
        me.step_z, me.step_y, me.step_x = (c - 2 * p for c, p in zip(crop, pad))
        me.vols_per_z, me.vols_per_y, me.vols_per_x = (math.ceil(d / s) for d, s in zip(dim, (me.step_z, me.step_y, me.step_x)))
        me.z_vol_start, me.vols_per_z_effective = 0, me.vols_per_z
        me.convert_to_rgb, me.norm_module = False, {}
        total = me.vols_per_z * me.vols_per_y * me.vols_per_x
        ext, real, pads, patches = [], [], [], []
        for vid in range(total):
            z, y, x, pe, pr = cls._patch_coords(me, vid)
            data, added = cls.extract_and_prepare_sample(me, z, y, x, pe)
            ext.append([pe.z_start, pe.z_end, pe.y_start, pe.y_end, pe.x_start, pe.x_end])
            real.append([pr.z_start, pr.z_end, pr.y_start, pr.y_end, pr.x_start, pr.x_end])
            pads.append(added)
            patches.append(data)
        out[f"{tag}/dim"], out[f"{tag}/crop"], out[f"{tag}/padding"] = np.array(dim), np.array(crop), np.array(pad)
        out[f"{tag}/extract"], out[f"{tag}/real"], out[f"{tag}/added_pad"] = np.array(ext), np.array(real), np.array(pads)
        out[f"{tag}/patch_sums"] = np.array([int(p_.astype(np.int64).sum()) for p_ in patches])
        out[f"{tag}/patch_wsums"] = np.array([int((p_.astype(np.int64).ravel() * (np.arange(p_.size) % 977 + 1)).sum()) for p_ in patches])
        if keep:
            out[f"{tag}/vol"] = vol
            out[f"{tag}/patches"] = np.stack(patches)
        out[f"{tag}/seed"] = np.array(seed)       # vol = RandomState(seed).randint(0, 256, dim + (1,)).astype(uint8)
        print(f"chunked {tag}: dim {dim} crop {crop} pad {pad}: {total} chunks")
    np.savez_compressed(os.path.join(HERE, "chunked_golden.npz"), **out)
    print("chunked_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "chunked_golden.npz")) // 1024, "KiB")


def rcan_fixtures():
    """RCAN trunk (row S; the 3-D up-scaling branch of the reference raises): reference ``rcan(ndim=3, upscaling_layer=False)``
    with 16 filters, 2 groups x 2 RCABs on a 16x24x32 patch, B = 2: output, L1 loss and all gradients."""
    rmod = shim.load("biapy.models.rcan")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import rcan_oracle

    torch.manual_seed(41)
    net = rmod.rcan(ndim=3, num_channels=1, filters=16, scale=2, num_rg=2, num_rcab=2, reduction=16, upscaling_layer=False, out_channels=1,
                    head_activations=["linear"])
    g = torch.Generator().manual_seed(141)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if v.ndim == 1:
                v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
    xl = torch.randn(2, 16, 24, 32, 1, generator=g)
    x = xl.permute(0, 4, 1, 2, 3)
    tgt = torch.randn(2, 1, 16, 24, 32, generator=g)
    net.train()
    y = net(x)
    loss = torch.nn.L1Loss()(y, tgt)          # SR workflows train with MAE (LOSS.TYPE = "MAE")
    loss.backward()
    out = {"x": xl.numpy(), "target": tgt.numpy(), "y": y.detach().numpy(), "loss": np.array(loss.item(), dtype=np.float64),
           "num_rg": np.array(2), "num_rcab": np.array(2)}
    for k, v in net.state_dict().items():
        out[f"sd/{k}"] = v.numpy()
    for k, p_ in net.named_parameters():
        out[f"gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        if p_.numel() <= 7000:
            out[f"grad/{k}"] = p_.grad.numpy()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    err = (rcan_oracle.rcan_forward(sd, x, 2, 2) - y.detach()).abs().max().item()
    print("rcan: params", sum(p_.numel() for p_ in net.parameters()), "oracle vs reference", err)
    assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "rcan_golden.npz"), **out)
    print("rcan_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "rcan_golden.npz")) // 1024, "KiB")


def resunetpp_fixtures():
    """ResUNet++ (row X, cfg 4 family; not yet on the device - the fixture pins the oracle the device path will be checked with):
    reference ``ResUNetPlusPlus`` with fm 16-32-64 on a 16x32x32 patch, B = 2, three output channels (B, C: ce_sigmoid; D: tanh
    are head activations outside the model): logits, an MSE loss and all gradient norms + a few full gradients."""
    rmod = shim.load("biapy.models.resunet++")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import resunetpp_oracle

    fm = [16, 32, 64]          # depth 1: every block type once or twice, 0.7 M parameters (a 4-level net is a 5.7 MB fixture)
    torch.manual_seed(51)
    with quiet():
        net = rmod.ResUNetPlusPlus(
            image_shape=(16, 32, 32, 1), activation="elu", feature_maps=fm, drop_values=[0.0] * 3, normalization="in", k_size=3,
            upsample_layer="convtranspose", yx_down=[2, 2], z_down=[2, 2], output_channels=[3], output_channel_info=["BCD"],
            head_activations=["ce_sigmoid", "ce_sigmoid", "linear"], isotropy=[True] * 3, larger_io=False, conv_layers=[2] * 3,
        )
    g = torch.Generator().manual_seed(151)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if v.ndim == 1:
                v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
    xl = torch.randn(2, 16, 32, 32, 1, generator=g)
    x = xl.permute(0, 4, 1, 2, 3)
    tgt = torch.randn(2, 3, 16, 32, 32, generator=g)
    net.train()
    logits = net(x)
    loss = torch.nn.MSELoss()(logits, tgt)
    loss.backward()
    out = {"feature_maps": np.array(fm), "x": xl.numpy(), "target": tgt.numpy().astype(np.float16), "logits": logits.detach().numpy(),
           "loss": np.array(loss.item(), dtype=np.float64)}
    for k, v in net.state_dict().items():
        out[f"sd/{k}"] = v.numpy().astype(np.float16) if v.numel() > 20000 else v.numpy()
    # big weights are stored as fp16: re-run the reference on the rounded weights / target so that outputs belong to the stored data
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(torch.from_numpy(out[f"sd/{k}"].astype(np.float32)))
    tgt = torch.from_numpy(out["target"].astype(np.float32))
    net.zero_grad()
    logits = net(x)
    loss = torch.nn.MSELoss()(logits, tgt)
    loss.backward()
    out["logits"], out["loss"] = logits.detach().numpy(), np.array(loss.item(), dtype=np.float64)
    names = dict(net.named_parameters())
    for k, p_ in names.items():
        out[f"gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
    for k in ["down_path.0.shortcut.0.weight", "sqex_blocks.0.excitation.0.weight", "attentions.0.0.conv_attn.2.weight",
              "aspp_out.0.aspp_block2.0.weight", "aspp_out.0.output.weight", "up_paths.0.0.conv_block.shortcut.1.weight", "heads.0.weight"]:
        out[f"grad/{k}"] = names[k].grad.numpy()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    err = (resunetpp_oracle.resunetpp_forward(sd, x, fm) - logits.detach()).abs().max().item()
    print("resunet++: params", sum(p_.numel() for p_ in net.parameters()), "oracle vs reference", err)
    assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "resunetpp_golden.npz"), **out)
    print("resunetpp_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunetpp_golden.npz")) // 1024, "KiB")


def resunet_sr_fixtures():
    """Row S, the 3-D super-resolution route that works in the reference: ``ResUNet(upsampling_factor=...,
    upsampling_position="pre" | "post")`` (resunet.py:206-213, :326-333, :368-369, :399-400).  Reference outputs, L1 loss (the SR
    workflows' MAE), gradient norms and a few full gradients of a small 3-D net: pre x(1,2,2), pre x(2,3,2), post x(2,2,2), post x(1,2,2)."""
    rmod = shim.load("biapy.models.resunet")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import net_oracle

    out = {}
    fm = [16, 32]
    for tag, patch, factor, pos, seed in (("pre122", (8, 16, 16), (1, 2, 2), "pre", 71), ("pre232", (4, 8, 16), (2, 3, 2), "pre", 72),
                                          ("post222", (8, 16, 16), (2, 2, 2), "post", 73), ("post122", (8, 16, 16), (1, 2, 2), "post", 74)):
        torch.manual_seed(seed)
        with quiet():
            net = rmod.ResUNet(
                image_shape=tuple(patch) + (1,), activation="elu", feature_maps=fm, drop_values=[0.0] * 2, normalization="in", k_size=3,
                upsample_layer="convtranspose", yx_down=[2], z_down=[2], output_channels=[1], output_channel_info=["F"], head_activations=["linear"],
                upsampling_factor=factor, upsampling_position=pos, isotropy=[True] * 2, larger_io=False, conv_layers=[2] * 2,
            )
        g = torch.Generator().manual_seed(100 + seed)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if v.ndim == 1:
                    v.add_(0.1 * (torch.rand(v.shape, generator=g) * 2 - 1))
        xl = torch.rand(2, *patch, 1, generator=g)
        x = xl.permute(0, 4, 1, 2, 3)
        hi = tuple(p * f for p, f in zip(patch, factor))
        tgt = torch.rand(2, 1, *hi, generator=g)
        net.train()
        y = net(x)
        assert tuple(y.shape[2:]) == hi
        loss = torch.nn.L1Loss()(y, tgt)
        loss.backward()
        out[f"{tag}/factor"], out[f"{tag}/pos"] = np.array(factor), np.array(pos)
        out[f"{tag}/x"], out[f"{tag}/target"] = xl.numpy(), tgt.numpy()
        out[f"{tag}/out"], out[f"{tag}/loss"] = y.detach().numpy(), np.array(loss.item(), dtype=np.float64)
        for k, v in net.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()
        names = dict(net.named_parameters())
        for k, p_ in names.items():
            out[f"{tag}/gradnorm/{k}"] = np.array(p_.grad.norm().item(), dtype=np.float64)
        full = ["down_path.0.block.0.block.0.weight", "down_path.0.shortcut.0.weight", "heads.0.weight", f"{pos}_upsampling.weight", f"{pos}_upsampling.bias"]
        for k in full:
            out[f"{tag}/grad/{k}"] = names[k].grad.numpy()
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        err = (net_oracle.resunet_forward(sd, x, fm) - y.detach()).abs().max().item()
        print(f"resunet SR {tag}: out {tuple(y.shape)}, oracle vs reference {err:.3e}")
        assert err < 2e-5
    np.savez_compressed(os.path.join(HERE, "resunet_sr_golden.npz"), **out)
    print("resunet_sr_golden.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "resunet_sr_golden.npz")) // 1024, "KiB")


def loss_inputs(seed=61, shape=(2, 6, 10, 12)):
    """Seeded logits / targets of the loss fixture (tests rebuild exactly these)."""
    g = torch.Generator().manual_seed(seed)
    B = shape[0]
    z1 = torch.randn(B, 1, *shape[1:], generator=g) * 2.0
    t1 = (torch.rand(B, 1, *shape[1:], generator=g) > 0.6).float()
    z3 = torch.randn(B, 3, *shape[1:], generator=g) * 1.5
    t3 = torch.cat([(torch.rand(B, 2, *shape[1:], generator=g) > 0.5).float(), torch.rand(B, 1, *shape[1:], generator=g) * 2 - 1], 1)
    return z1, t1, z3, t3


def loss_inputs_multiclass(seed=67, shape=(2, 6, 10, 12)):
    """Seeded class logits / label maps of the multi-class loss fixture (tests rebuild exactly these): 3 and 5 classes, the 5-class label map
    with ~10 % of the voxels set to the ignore value 255."""
    g = torch.Generator().manual_seed(seed)
    B = shape[0]
    z3 = torch.randn(B, 3, *shape[1:], generator=g) * 2.0
    y3 = torch.randint(0, 3, (B, 1, *shape[1:]), generator=g).float()
    z5 = torch.randn(B, 5, *shape[1:], generator=g) * 1.5
    y5 = torch.randint(0, 5, (B, 1, *shape[1:]), generator=g).float()
    y5[torch.rand(B, 1, *shape[1:], generator=g) < 0.1] = 255.0
    return z3, y3, z5, y5


def losses_multiclass_fixtures():
    """Row L, multi-class case (round 6): the reference's CrossEntropyLoss_wrapper with num_classes > 2 (metrics.py:493-586 ->
    torch.nn.CrossEntropyLoss) - plain, with "manual" class weights, with an ignore value, and on a list of two predictions (deep
    supervision weights 0.5^i / sum, target rescaled by scale_target) - value and gradient w.r.t. the logits."""
    met = shim.load("biapy.engine.metrics")
    z3, y3, z5, y5 = loss_inputs_multiclass()
    out = {}

    def rec(name, fn, *zs):
        zz = [z.clone().requires_grad_(True) for z in zs]
        val = fn(*zz)
        val.backward()
        out[f"{name}/value"] = np.float64(val.item())
        for k, z in enumerate(zz):
            out[f"{name}/grad{k}"] = z.grad.numpy().copy()
        print(name, val.item())

    rec("ce3", lambda z: met.CrossEntropyLoss_wrapper(num_classes=3, ndim=3)(z, y3), z3)
    rec("ce3_w", lambda z: met.CrossEntropyLoss_wrapper(num_classes=3, ndim=3, class_rebalance="manual", class_weights=[0.2, 0.5, 0.3])(z, y3), z3)
    rec("ce5_ignore", lambda z: met.CrossEntropyLoss_wrapper(num_classes=5, ndim=3, ignore_index=255)(z, y5), z5)
    rec("ce5_w_ignore", lambda z: met.CrossEntropyLoss_wrapper(num_classes=5, ndim=3, class_rebalance="manual", class_weights=[1.0, 2.0, 0.5, 0.25, 4.0],
                                                               ignore_index=255)(z, y5), z5)
    g = torch.Generator().manual_seed(71)
    zh = torch.randn(2, 3, 3, 5, 6, generator=g)          # a half-resolution second prediction
    rec("ce3_deep", lambda a, b: met.CrossEntropyLoss_wrapper(num_classes=3, ndim=3)([a, b], y3), z3, zh)
    out["ce3_deep/zh"] = zh.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "losses_multiclass_golden.npz"), **out)
    print("losses_multiclass_golden.npz:", len(out), "arrays")


def losses_fixtures():
    """Rows L and X: the reference's own loss classes (biapy/engine/metrics.py) on seeded inputs - value and gradient w.r.t. the
    logits of CrossEntropyLoss_wrapper (binary), DiceLoss, DiceCELoss (two weightings) and instance_segmentation_loss for the
    B, C, D channels (bce, bce, mse|l1; the D channel through tanh as the workflow applies it in training)."""
    met = shim.load("biapy.engine.metrics")
    z1, t1, z3, t3 = loss_inputs()
    out = {}

    def rec(name, fn, z):
        zz = z.clone().requires_grad_(True)
        val = fn(zz)
        val.backward()
        out[f"{name}/value"] = np.float64(val.item())
        out[f"{name}/grad"] = zz.grad.numpy().copy()
        print(name, val.item())

    rec("bce", lambda z: met.CrossEntropyLoss_wrapper(num_classes=2, ndim=3)(z, t1), z1)
    rec("dice", lambda z: met.DiceLoss(batch_dice=True)(z, t1), z1)
    rec("dice_per_sample", lambda z: met.DiceLoss(batch_dice=False)(z, t1), z1)     # round 6: Dice per sample, then the mean (metrics.py:749-751)
    rec("dice_ce_1_1", lambda z: met.DiceCELoss(num_classes=2, ndim=3)(z, t1), z1)
    rec("dice_ce_03_17", lambda z: met.DiceCELoss(num_classes=2, ndim=3, w_ce=0.3, w_dice=1.7)(z, t1), z1)
    acts = ["ce_sigmoid", "ce_sigmoid", "tanh"]

    def train_act(z):           # what model_call_func(is_train=True) does to the logits (base_workflow.py:1403-1457)
        return torch.cat([z[:, 0:1], z[:, 1:2], torch.tanh(z[:, 2:3])], 1)

    for tag, losses, w in (("bcd_mse", ["bce", "bce", "mse"], (1, 1, 1)), ("bcd_l1_w", ["bce", "bce", "l1"], (0.5, 0.25, 2.0))):
        crit = met.instance_segmentation_loss(channel_weights=w, ndim=3, out_channels=["B", "C", "D"], losses_to_use=losses,
                                              channel_extra_opts={}, gt_channels_expected=3)
        rec(f"instance_{tag}", lambda z, crit=crit: crit(train_act(z), t3), z3)
    del acts
    np.savez_compressed(os.path.join(HERE, "losses_golden.npz"), **out)
    print("losses_golden.npz:", len(out), "arrays")


def train_loop_case(name):
    """Seeded toy problem of the train-loop fixture: (net, data, val data, cfg values).  Tests rebuild exactly this."""
    import types

    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ELU(), torch.nn.Conv3d(4, 1, 1))
    g = torch.Generator().manual_seed(12)
    data = [(torch.randn(2, 4, 6, 6, 1, generator=g), (torch.rand(2, 4, 6, 6, 1, generator=g) > 0.5).float()) for _ in range(13)]
    val = [(torch.randn(2, 4, 6, 6, 1, generator=g), (torch.rand(2, 4, 6, 6, 1, generator=g) > 0.5).float()) for _ in range(4)]
    clip, sched = {"plain": (0.0, ""), "clip_onecycle": (0.05, "onecycle"), "plateau": (0.0, "reduceonplateau")}[name]
    cfg = types.SimpleNamespace(DATA=types.SimpleNamespace(PATCH_SIZE=(4, 6, 6, 1)),
                                TRAIN=types.SimpleNamespace(GRADIENT_CLIP_NORM=clip, LR_SCHEDULER=types.SimpleNamespace(NAME=sched), VERBOSE=False))
    return net, data, val, cfg


def train_loop_fixtures():
    """Row T: the reference's own train_one_epoch / evaluate (biapy/engine/train_engine.py) run on a toy CPU problem, called
    exactly as Base_Workflow.train does (base_workflow.py:1070-1088, :1114-1126).  Stored: the returned stats, the step index
    and the trained weights of three cases (plain; gradient clipping + one-cycle schedule; reduce-on-plateau over 3 epochs)."""
    import typing

    shim.install()
    sys.modules["biapy.engine"].Scheduler = typing.Any
    te = shim.load("biapy.engine.train_engine")
    dev = torch.device("cpu")
    out = {}
    for name in ("plain", "clip_onecycle", "plateau"):
        net, data, val, cfg = train_loop_case(name)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
        sched = None
        if name == "clip_onecycle":
            sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-2, total_steps=len(data) * 2)
        if name == "plateau":
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=0, threshold=10.0)   # reduces after every epoch
        loss_fn = torch.nn.BCEWithLogitsLoss()

        def call(batch, is_train=False):
            return net(batch.permute(0, 4, 1, 2, 3))

        def prep(targets, batch):
            return targets.permute(0, 4, 1, 2, 3)

        def metric(outputs, targets, metric_logger=None):
            p_ = (torch.sigmoid(outputs) > 0.5).float()
            iou = ((p_ * targets).sum() / torch.clamp(((p_ + targets) > 0).float().sum(), min=1.0)).item()
            if metric_logger:
                metric_logger.meters["IoU"].update(iou)

        epochs = 3 if name == "plateau" else 2
        for epoch in range(epochs):
            with quiet():
                stats, step = te.train_one_epoch(cfg, model=net, model_call_func=call, loss_function=loss_fn, metric_function=metric,
                                                 prepare_targets=prep, data_loader=data, optimizer=[opt], device=dev, epoch=epoch, log_writer=None,
                                                 lr_scheduler=[sched], verbose=False, memory_bank=None, total_iters=0, contrast_warmup_iters=0,
                                                 loss_names=["loss"])
                ev = te.evaluate(cfg, model=net, model_call_func=call, loss_function=loss_fn, metric_function=metric, prepare_targets=prep,
                                 epoch=epoch, data_loader=val, lr_scheduler=[sched], memory_bank=None, loss_names=["loss"])
            for k, v in stats.items():
                out[f"{name}/e{epoch}/train/{k}"] = np.float64(v)
            for k, v in ev.items():
                out[f"{name}/e{epoch}/val/{k}"] = np.float64(v)
            out[f"{name}/e{epoch}/step"] = np.int64(step)
        for k, v in net.state_dict().items():
            out[f"{name}/w/{k}"] = v.numpy().copy()
        print(name, {k: float(v) for k, v in stats.items()}, {k: float(v) for k, v in ev.items()}, "lr", opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(HERE, "train_loop_golden.npz"), **out)
    print("train_loop_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiling", "tiling2d", "resunet", "resunet_aniso", "prepost", "tta", "tta_ensemble", "tta_spec", "unet", "resunet_variants", "resunet_activations", "resunet_class_head", "resunet_explicit_tail", "resunet_dropout", "chunked", "rcan", "resunetpp", "train_loop", "losses", "losses_multiclass", "resunet_sr"]
    if "prepost" in which:
        prepost_fixtures()
    if "tta" in which:
        tta_fixtures()
    if "tta_ensemble" in which:
        tta_ensemble_fixtures()
    if "tta_spec" in which:
        tta_spec_fixtures()
    if "build_model" in which:                  # full import: on its own
        build_model_kwargs_fixture()
    if "head_acts" in which:                    # full import as well: on its own
        head_acts_fixtures()
    if "harness" in which:                      # imports the whole reference package: run it on its own (python make_golden.py harness)
        harness_fixtures()
    if "harness_tail" in which:                 # likewise on its own
        harness_tail_fixtures()
    if "tiling" in which:
        tiling_fixtures()
    if "tiling2d" in which:
        tiling2d_fixtures()
    if "resunet" in which:
        resunet_fixtures()
    if "resunet_aniso" in which:
        resunet_aniso_fixtures()
    if "unet" in which:
        unet_fixtures()
    if "resunet_variants" in which:
        resunet_variants_fixtures()
    if "resunet_activations" in which:
        resunet_activations_fixtures()
    if "resunet_dropout" in which:
        resunet_dropout_fixtures()
    if "resunet_class_head" in which:
        resunet_class_head_fixtures()
    if "resunet_explicit_tail" in which:
        resunet_explicit_tail_fixtures()
    if "chunked" in which:
        chunked_fixtures()
    if "rcan" in which:
        rcan_fixtures()
    if "resunetpp" in which:
        resunetpp_fixtures()
    if "train_loop" in which:
        train_loop_fixtures()
    if "losses" in which:
        losses_fixtures()
    if "losses_multiclass" in which:
        losses_multiclass_fixtures()
    if "resunet_sr" in which:
        resunet_sr_fixtures()
