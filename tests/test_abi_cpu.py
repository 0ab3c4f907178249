"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol that
include/mtm_hip.h declares, refuses to run without a GPU, and its host-only NMS matches the
oracle's restatement of cv2.dnn.NMSBoxes."""
import ctypes
import os
import re

import numpy as np
import pytest

import mtm_oracle as O
from helpers import load_golden

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    import build as mtm_build   # multitemplatematching-python_amd/build.py
    mtm_build.build()
    from MTM import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mtm_hip.h")).read()
    body = hdr[hdr.index('extern "C"'):]
    declared = set(re.findall(r"\b(mtm_[a-z_0-9]+)\s*\(", body))
    declared -= {"mtm_ctx", "mtm_templ", "mtm_hit", "mtm_timing"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert lib.load().mtm_abi_version() == lib.ABI_VERSION == 9


def test_struct_layout(lib):
    assert ctypes.sizeof(lib.MtmHit) == 24
    assert ctypes.sizeof(lib.MtmTempl) == 48
    assert lib.HIT_DTYPE.itemsize == 24


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback(lib):
    assert lib.load().mtm_device_count() == 0
    with pytest.raises(lib.MtmError, match="no HIP device|NO_DEVICE|-3"):
        lib.Context(0)
    import MTM
    img = np.zeros((32, 32), np.uint8)
    with pytest.raises(lib.MtmError):
        MTM.matchTemplates([("t", img[:8, :8])], img)


def test_nms_demo_through_abi(lib):
    import MTM
    demo = [("1", (780, 350, 700, 480), 0.8), ("1", (806, 416, 716, 442), 0.6), ("1", (1074, 530, 680, 390), 0.4)]
    out = MTM.NMS(demo, scoreThreshold=0.3, sortAscending=False, maxOverlap=0.5, N_object=2)
    assert [h[1] for h in out] == [demo[0][1], demo[2][1]]
    ref = load_golden()["reference_run"]["nms_demo"]
    # (the fixture stores scores narrowed to float32)
    assert [[h[0], list(h[1]), float(np.float32(h[2]))] for h in out] == ref


def _random_hits(rng, n, size=400, wh=(20, 60)):
    hits = []
    for i in range(n):
        w, h = rng.integers(wh[0], wh[1], 2)
        x, y = rng.integers(0, size, 2)
        hits.append(("t%d" % (i % 3), (int(x), int(y), int(w), int(h)), np.float32(rng.random())))
    return hits


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("ascending", [False, True])
def test_nms_matches_oracle(lib, seed, ascending):
    import MTM
    rng = np.random.default_rng(seed)
    hits = _random_hits(rng, 300)
    # some exact ties and duplicates
    hits += [hits[0], (hits[1][0], hits[1][1], hits[2][2])]
    for thr, ov, nobj in ((0.3, 0.25, float("inf")), (0.5, 0.0, 7), (0.1, 1.0, float("inf")), (0.7, 0.5, 0)):
        got = MTM.NMS(hits, thr, ascending, nobj, ov)
        exp = O.NMS(hits, thr, ascending, nobj, ov)
        assert got == exp


def test_nms_abi_ascending_flag(lib):
    """The C entry point's own 1-score transform (MTM/NMS.py:73-75) == doing it in Python."""
    rng = np.random.default_rng(11)
    hits = _random_hits(rng, 200)
    boxes = [h[1] for h in hits]
    scores = [h[2] for h in hits]
    a = lib.nms_indices(boxes, scores, 0.4, 0.3, ascending=True, n_object=5)
    b = lib.nms_indices(boxes, [1 - s for s in scores], 1 - 0.4, 0.3)[:5]
    assert list(a) == list(b)


def test_nms_edge_cases(lib):
    import MTM
    assert MTM.NMS([]) == []
    one = [("a", (0, 0, 5, 5), 0.1)]
    assert MTM.NMS(one, scoreThreshold=0.9) == one          # a single hit bypasses the threshold
    two = [("a", (0, 0, 5, 5), 0.9), ("b", (0, 0, 5, 5), 0.9)]
    assert MTM.NMS(two, N_object=1) == [two[0]]              # first wins ties
    assert MTM.NMS(two, maxOverlap=0.5) == [two[0]]          # identical boxes: IoU 1 > 0.5
    assert MTM.NMS(two, maxOverlap=1.0) == two
    assert MTM.NMS(two, scoreThreshold=0.95) == []


def test_score_kernel_does_not_spill_accumulators():
    """ncc_mfma_kernel holds 128 (three-row variant: 192) accumulator VGPRs next to inline-asm MFMA steps; an
    instantiation whose register allocation tips over starts spilling around those steps - slow, and (seen once,
    DESIGN 4.1 "packed K") wrong.  No instantiation may have a scratch instruction between its first and its last
    MFMA; the one- and two-row instantiations keep their total scratch small (a few prologue / epilogue values), the
    two-row headline variants have none at all; the three-row ones spill only on the rare paths behind the screen, the
    uint16 kernel (three sets as well) only in its epilogue."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    import build as mtm_build
    mtm_build.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = [k for k in kr.kernels(mtm_build.LIB) if "ncc_mfma_kernel" in k["name"]]
    assert len(ks) >= 100
    in_loop = kr.loop_scratch_ops(mtm_build.LIB)
    assert len(in_loop) == len(ks) and not any(in_loop.values()), {k: v for k, v in in_loop.items() if v}
    # 192 accumulator registers: the three-row variants and the uint16 kernel (three partial-sum sets, METHOD 7)
    three_row = lambda k: re.search(r"ncc_mfma_kernelILi(3E|2ELi7E)", k["name"]) is not None      # noqa: E731
    worst = max((k for k in ks if not three_row(k)), key=lambda k: k["scratch"])
    assert worst["scratch"] <= 200, worst
    assert all(k["scratch"] <= 640 for k in ks if three_row(k))
    # <2, method, exact, false, false, 1, ext, R2 = true, false>: the two-row instantiations - IEEE division (the default
    # since round 5) and reciprocal, with and without the fused extremum; the headline kernel is <2, 5, true, ..., false, true>
    two_row = [k for k in ks if re.search(r"ncc_mfma_kernelILi2ELi[2-5]ELb[01]ELb0ELb0ELi1ELb[01]ELb1ELb0E", k["name"])]
    assert len(two_row) == 16 and all(k["scratch"] == 0 for k in two_row), [k for k in two_row if k["scratch"]]
    assert all(k["vgpr"] <= 256 for k in ks)


def _spill_exec_scan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("spill_exec_scan", os.path.join(ROOT, "tools", "spill_exec_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_spill_ahead_of_an_exec_restore():
    """The code-generation defect behind round 5's state-dependent uint16 scores (profiles/r06_flake/): this
    compiler can place the register allocator's spill stores at the top of a join block AHEAD of the `s_or_b64 exec` that
    re-opens the execution mask, so only the lanes of the divergent region that just ended save their register - and
    every lane reloads it.  The build keeps the device assembly of every unit (-save-temps); no kernel of the library may
    contain that placement, whatever the source looks like."""
    import build as mtm_build
    mtm_build.build()
    scan = _spill_exec_scan()
    files, hits = scan.scan_all()
    assert len(files) >= 11, files                     # every .hip unit left its assembly behind
    names = " ".join(os.path.basename(f) for f in files)
    for unit in ("mtm_mfma_plain", "mtm_mfma_rows", "mtm_mfma_rm", "mtm_mfma_ext", "mtm_mfma_kp", "mtm_bf16", "mtm_api", "mtm_launch"):
        assert unit + "-hip-amdgcn-amd-amdhsa-gfx950.s" in names, unit
    assert not hits, hits[:3]


def test_spill_scan_sees_the_round5_defect(tmp_path):
    """The detector itself, pinned on the block that produced the wrong scores (the uint16 reciprocal packed-K kernel of
    commit 960e744 + profiles/r05_flake/detector_build.patch; excerpt committed under profiles/r06_flake/) and on its
    correctly ordered sibling (the IEEE-division instantiation of the same build)."""
    scan = _spill_exec_scan()
    bad = open(os.path.join(ROOT, "profiles", "r06_flake", "failing_kernel_excerpt.s")).read()
    good = open(os.path.join(ROOT, "profiles", "r06_flake", "exact_div_sibling_excerpt.s")).read()
    for name, text, expect in (("bad.s", bad, 1), ("good.s", good, 0)):
        p = tmp_path / name
        p.write_text("_ZN3mtm15ncc_mfma_kernelTEST:\n" + text + "\n.Lfunc_end0:\n")
        found = scan.scan(str(p))
        assert len(found) == expect, (name, found)
    (_, _, restore, offenders), = scan.scan(str(tmp_path / "bad.s"))
    assert restore.startswith("s_or_b64 exec, exec")
    assert sum("scratch_store_dwordx4" in t for _, t in offenders) == 3      # the three accumulator vectors (pixels 4..6)
