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

# The thresholds below (kernel counts by mangled name, minimum counted waits, block shapes) were read off the code hipcc 7.2 generates; another
# compiler release may legitimately schedule differently, so the module is skipped rather than failed there.
HIPCC_DERIVED_WITH = "7.2"


def _hipcc_version() -> str:
    if not os.path.exists(HIPCC):
        return ""
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version:\s*(\d+\.\d+)", out)
    return m.group(1) if m else ""


pytestmark = [pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc to cross-compile the kernels to ISA"),
              pytest.mark.skipif(os.path.exists(HIPCC) and _hipcc_version() != HIPCC_DERIVED_WITH,
                                 reason=f"ISA thresholds were derived with hipcc {HIPCC_DERIVED_WITH}, this is {_hipcc_version()}")]


def _isa(src: str, tmp_path_factory) -> str:
    out = str(tmp_path_factory.mktemp("isa") / (src + ".s"))
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result", "-S", "--cuda-device-only",   # the Makefile's flags
           os.path.join(ROOT, "biapy_amd", "csrc", src + ".hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, f"{' '.join(cmd)} failed:\n{r.stderr[-3000:]}"
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


@pytest.fixture(scope="module")
def bwd_fused_isa(tmp_path_factory):
    return _isa("bwd_fused", tmp_path_factory)


def _blocks(text: str, name_pattern: str):
    """[(kernel, [(label, [instruction lines])])] - every basic block of the kernels whose mangled name matches."""
    found = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        if not re.search(name_pattern, name):
            continue
        blocks, cur = [("entry", [])], None
        for line in body:
            mm = re.match(r"^\.L(BB\d+_\d+):", line)
            if mm:
                blocks.append((mm.group(1), []))
                continue
            t = line.strip()
            if t and not t.startswith((";", ".", "//")):
                blocks[-1][1].append(t)
        found.append((name, blocks))
    return found


def test_role_split_backward_hot_blocks(bwd_fused_isa):
    """conv3_bwd_rs_kernel (round 6), 48-channel instance: what its speed rests on, as properties of the generated code.
    * the D waves' tile is ONE basic block (branch-free epilogue: masks and out-of-range buffer stores instead of predicates) with the 168 dgrad
      MFMAs, 12 64-bit buffer stores and NO vector-memory load (weights come from LDS and registers);
    * the W waves' wgrad phase exists once per wave (blocks of 168 / 147 / 19 wgrad MFMAs), touches no scratch and never waits on vmcnt -
      the next tile's LDS-DMA pieces, requested by inline assembly just before it, stay in flight across the phase;
    * no hot block spills."""
    ks = _blocks(bwd_fused_isa, r"conv3_bwd_rs_kernelILi3ELi1ELb1E")
    assert len(ks) == 1, [k for k, _ in ks]
    _, blocks = ks[0]
    hot = [(lab, ins) for lab, ins in blocks if sum(1 for t in ins if t.startswith("v_mfma")) >= 15]   # (wave 3's phase stays a loop of K-chunks: the bias row)
    d_blocks = [(lab, ins) for lab, ins in hot if any(t.startswith("buffer_store_dwordx2") for t in ins)]
    w_blocks = [(lab, ins) for lab, ins in hot if any(t.startswith("ds_read_b64_tr_b16") for t in ins)]
    assert len(d_blocks) == 1 and len(w_blocks) >= 4, ([l for l, _ in d_blocks], [l for l, _ in w_blocks])
    lab, ins = d_blocks[0]
    assert sum(1 for t in ins if t.startswith("v_mfma")) == 168 and sum(1 for t in ins if t.startswith("buffer_store_dwordx2")) == 12, lab
    assert not any(t.startswith(("buffer_load", "global_load")) for t in ins), f"{lab}: the dgrad step loop requests weights from L2 again"
    assert sum(1 for t in ins if t.startswith("v_exp_f32")) == 48, lab          # the whole epilogue is inside the block
    for lab, ins in hot:
        assert not any(t.startswith("scratch_") for t in ins), f"{lab}: a hot block of the role-split kernel spills"
    for lab, ins in w_blocks:
        assert not any(re.match(r"s_waitcnt vmcnt", t) for t in ins), f"{lab}: the wgrad phase waits for the next tile's DMA pieces"
    dma = [t for _, ins in blocks for t in ins if re.match(r"buffer_load_dwordx4 .* offen lds", t)]
    assert len(dma) >= 4 * 12, len(dma)                                           # 12 pieces per W wave, once per instantiation (+ the prologue's)
