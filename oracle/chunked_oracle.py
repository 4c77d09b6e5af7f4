"""CPU oracle of the by-chunks tiler (NumPy).  TEST INFRASTRUCTURE ONLY: imported by ``tests/`` only.

Parity status: PINNED.  ``tests/golden/make_golden.py chunked`` imports the reference's
``chunked_test_pair_data_generator`` class in the build container (third-party modules it never calls on this path are
stubbed), calls its own ``_patch_coords`` / ``extract_and_prepare_sample`` on seeded volumes and stores, per chunk, the read
region, the write-back region, the padding added and the padded patches in ``tests/golden/chunked_golden.npz``;
``tests/test_oracle_golden.py`` checks this restatement against them.

Restates (paths relative to /root/reference):
  * grid ................ biapy/data/generators/chunked_test_pair_data_generator.py:272-289 (step, chunks per axis)
  * chunk regions ....... :440-487 (``_patch_coords``)
  * read + reflect pad .. :524-565 (``extract_and_prepare_sample``; ``np.pad(data, pad_to_add, "reflect")``)
  * write-back .......... biapy/engine/base_workflow.py:2603-2610 (strip ``max(pad added, padding)``, insert at the chunk)
"""
from __future__ import annotations

import math

import numpy as np


def grid(dim, crop, padding):
    step = [c - 2 * p for c, p in zip(crop[:3], padding)]
    vols = [math.ceil(d / s) for d, s in zip(dim, step)]
    return step, vols


def patch_coords(vol_id, dim, crop, padding):
    step, vols = grid(dim, crop, padding)
    q = [int(v) for v in np.unravel_index(vol_id, vols)]
    ext = [(max(0, q[a] * step[a] - padding[a]), min((q[a] + 1) * step[a] + padding[a], dim[a])) for a in range(3)]
    real = [(q[a] * step[a], min((q[a] + 1) * step[a], dim[a])) for a in range(3)]
    return q, ext, real


def extract(vol, vol_id, crop, padding):
    """(padded patch, pad_to_add after the max(., padding) update) of chunk ``vol_id`` of ``vol`` (Z,Y,X,C)."""
    dim = vol.shape[:3]
    step, _ = grid(dim, crop, padding)
    q, ext, _ = patch_coords(vol_id, dim, crop, padding)
    data = vol[ext[0][0]:ext[0][1], ext[1][0]:ext[1][1], ext[2][0]:ext[2][1]]
    pads = []
    for a in range(3):
        left = abs(q[a] * step[a] - padding[a]) if q[a] * step[a] - padding[a] < 0 else 0
        pads.append([left, crop[a] - (ext[a][1] - ext[a][0]) - left])
    data = np.pad(data, pads + [[0, 0]], "reflect")
    strip = [[max(p[0], padding[a]), max(p[1], padding[a])] for a, p in enumerate(pads)]
    return data, pads, strip


def predict_by_chunks(vol, pred_func, crop, padding):
    """Whole pipeline on the CPU: every chunk read, predicted by ``pred_func((1,Pz,Py,Px,C)) -> (1,Pz,Py,Px,Cout)`` and inserted."""
    dim = vol.shape[:3]
    _, vols = grid(dim, crop, padding)
    out = None
    for vid in range(vols[0] * vols[1] * vols[2]):
        patch, _, strip = extract(vol, vid, crop, padding)
        pred = np.asarray(pred_func(patch[None]))[0]
        _, _, real = patch_coords(vid, dim, crop, padding)
        raw = pred[strip[0][0]:pred.shape[0] - strip[0][1], strip[1][0]:pred.shape[1] - strip[1][1], strip[2][0]:pred.shape[2] - strip[2][1]]
        if out is None:
            out = np.zeros(tuple(dim) + (pred.shape[-1],), dtype=np.float32)
        out[real[0][0]:real[0][1], real[1][0]:real[1][1], real[2][0]:real[2][1]] = raw
    return out
