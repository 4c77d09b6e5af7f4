"""Properties of the COMPILED kernels that the round-5 speed-ups rest on (no GPU: hipcc cross-compiles gfx950 here).

A wave's VMEM operations retire in order, so an ``s_waitcnt vmcnt(0)`` inside a persistent loop is a wait for every store issued so far; hipcc emits
one wherever its wait-count pass loses track (joins of predicated loads / stores, inner loops with run-time trip counts, exit tests between unrolled
blocks).  The kernels below were rewritten so that their steady-state loops wait with COUNTED waits only; a refactor
that re-introduces a predicate or an inner test would silently bring the ``vmcnt(0)`` back - this test sees it.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc to cross-compile the kernels to ISA")


def _isa(src: str, tmp_path_factory) -> str:
    out = str(tmp_path_factory.mktemp("isa") / (src + ".s"))
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only",
           os.path.join(ROOT, "biapy_amd", "csrc", src + ".hip"), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.fixture(scope="module")
def pointwise_isa(tmp_path_factory):
    return _isa("pointwise", tmp_path_factory)


@pytest.fixture(scope="module")
def elementwise_isa(tmp_path_factory):
    return _isa("elementwise", tmp_path_factory)


def _loops(text: str, name_pattern: str):
    """[(kernel, [lines of a loop = header block + the blocks annotated 'in Loop: Header=<it>'])] of the kernels whose mangled name matches."""
    found = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        if not re.search(name_pattern, name):
            continue
        loops, cur = {}, None
        for line in body:
            mm = re.match(r"^\.L(BB\d+_\d+):(.*)", line)
            if mm:
                lab, rest = mm.group(1), mm.group(2)
                hm = re.search(r"Header=(BB\d+_\d+)", rest)
                cur = lab if "Loop Header" in rest else (hm.group(1) if hm else None)
                if "Loop Header" in rest:
                    loops.setdefault(lab, [])
            if cur is not None and cur in loops:
                loops[cur].append(line)
        found.append((name, list(loops.values())))
    return found


def _main_loop(loops):
    """The loop with the most memory instructions (the persistent block / tile loop; the others are epilogue reductions)."""
    return max(loops, key=lambda reg: sum(1 for l in reg if re.search(r"(buffer|global)_(load|store)", l)))


def test_transposed_conv_one_k_step_kernel_waits_with_counts_only(pointwise_isa):
    kernels = _loops(pointwise_isa, r"convt_k1_kernel")
    assert len(kernels) == 2, [k for k, _ in kernels]               # fp16 and bf16 storage
    for name, loops in kernels:
        reg = _main_loop(loops)
        stores = [l for l in reg if "buffer_store_dwordx4" in l]
        loads = [l for l in reg if "buffer_load_dwordx4" in l]
        assert len(stores) >= 8 and len(loads) >= 4, (name, len(stores), len(loads))
        assert not any("s_waitcnt vmcnt(0)" in l for l in reg), f"{name}: the block loop waits for all its stores again"
        assert not any(re.search(r"buffer_store_dwordx4 .*, s\d+ offen", l) for l in stores), \
            f"{name}: a 128-bit buffer store with an SGPR soffset (the second of two back-to-back ones loses its data: scripts/probes/soffset_store_hazard.hip)"
        waits = [int(x) for l in reg for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)]
        assert waits and min(waits) >= 8, (name, waits)              # the stores of at least one whole block stay in flight at every wait


def test_first_layer_buffer_instances_wait_with_counts_only(elementwise_isa):
    fwd = _loops(elementwise_isa, r"conv_c1_fwd_kernelI\w+Lb1E")
    assert len(fwd) == 3, [k for k, _ in fwd]                       # bf16, fp16, fp32 storage
    for name, loops in fwd:
        reg = _main_loop(loops)
        assert any("buffer_store" in l for l in reg) and any("buffer_load_dword " in l for l in reg), name
        assert not any("s_waitcnt vmcnt(0)" in l for l in reg), f"{name}: the tile loop waits for the previous tile's stores again"
    wg = _loops(elementwise_isa, r"conv_c1_wgrad_mfma_kernelI\w+Lb1ELb1E")
    assert len(wg) == 2, [k for k, _ in wg]                         # t as fp16 (mixed mode) and bf16, IN-backward folded in, buffer-addressed
    for name, loops in wg:
        reg = _main_loop(loops)
        assert not any("s_waitcnt vmcnt(0)" in l for l in reg), f"{name}: a staged tile's wait covers the younger tile's requests again"
        assert any("s_load_dwordx16" in l for l in reg), f"{name}: the sample's coefficients no longer come through the scalar cache"
        waits = [int(x) for l in reg for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)]
        assert waits and min(waits) >= 7, (name, waits)              # the other stage's seven requests stay in flight
