"""bench.py's pure helpers (no GPU): workload sizes of the BASELINE configs, the algorithmic byte / MAC
counts the roofline is computed from, defaults of the command line."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_and_constants(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 20 and a.warmup >= 1 and a.config == "north_star"
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.I8_MFMA_PEAK_TOPS == 5000.0


def test_north_star_workload_and_counts():
    img, units, plants, method, thr, desc = bench.build_workload("north_star", 1)
    assert img.shape == (2160, 3840) and img.dtype == np.uint8 and len(units) == 32
    assert all(u[1].shape == (64, 64) for u in units) and method == 5 and thr == 0.5
    out_px = (2160 - 63) * (3840 - 63)
    assert bench.score_kernel_macs(img, units) == 32 * out_px * 4096 == 1038138605568
    assert bench.masked_stat_macs(img, units) == 0
    assert bench.algorithmic_bytes(img, units) == img.nbytes + 32 * (4096 + 4 * out_px) == 1022232704
    assert bench.algorithmic_bytes_hits_only(img, units) == img.nbytes + 32 * 4096 + 2 * 8 * out_px
    # weak scaling: 32 units per GPU
    _, units8, _, _, _, _ = bench.build_workload("north_star", 8)
    assert len(units8) == 256


def test_masked_mac_accounting():
    """cfg5: the timed score kernel does ONE correlation per masked unit (sum I*(T*M)); sum I^2*M is computed once per
    distinct mask by launches outside the kernel timer and is accounted separately (a roofline fraction above 1 in
    round 1 came from counting it inside)."""
    import synth
    img, units, _ = synth.make_workload(seed=5, image_hw=(300, 400), n_base=3, templ=32, scales=(16, 32), masked=True)
    H, W = img.shape
    per_unit = sum((H - u[1].shape[0] + 1) * (W - u[1].shape[1] + 1) * u[1].shape[0] * u[1].shape[1] for u in units)
    assert bench.score_kernel_macs(img, units) == per_unit
    per_mask = sum(2 * (H - s + 1) * (W - s + 1) * s * s for s in (16, 32))     # two byte planes of I^2, one mask per size
    assert bench.masked_stat_macs(img, units) == per_mask
    assert "cfg4" in bench.CONFIGS


def test_pmc_traffic_table():
    t = bench.pmc_traffic(3, "north_star", 1, hits_only=True)
    m = bench.pmc_traffic(3, "north_star", 1, hits_only=False)
    # no wasted re-reads: at most 1.3 x the algorithmic bytes (hits-only launches read the 16-pixel block ranges of the
    # window statistics instead of the per-pixel planes the algorithmic figure counts - they stay well below it); with the
    # maps written 1.35: the map-mode instantiation spills 36 bytes per lane around its epilogue (1.31 x)
    assert t and m and 0.1 <= t / 135151376 < 1.3 and 1.0 <= m / 1022232704 < 1.35
    # the other BASELINE configs: one launch of the dominant kernel of the workload's own PMC passes; nothing for N > 1
    c5 = bench.pmc_traffic(3, "cfg5", 1)
    assert c5 is not None and c5 > 34e6 and bench.pmc_traffic(3, "cfg5", 8) is None     # (an 8K image alone is 33 MB)
    assert bench.pmc_traffic(3, "no_such_config", 1) is None


def test_synthetic_smooth_image_and_crops():
    """synth.smooth_u8 / cut_templates (the photograph-like workload of bench.py's extra and of the dense-map tests):
    seeded, full byte range, neighbouring pixels correlated; crops are copies at in-range positions."""
    import synth
    a = synth.smooth_u8(7, (120, 200))
    assert a.dtype == np.uint8 and a.shape == (120, 200) and a.min() == 0 and a.max() == 255
    assert np.array_equal(a, synth.smooth_u8(7, (120, 200))) and not np.array_equal(a, synth.smooth_u8(8, (120, 200)))
    d = np.abs(np.diff(a.astype(np.int32), axis=1)).mean()
    w = np.abs(np.diff(synth.rand_u8(7, 0, (120, 200)).astype(np.int32), axis=1)).mean()
    assert d < 0.35 * w                                     # far smoother than white noise
    lt = synth.cut_templates(3, a, 5, 24)
    assert [n for n, _ in lt] == ["t0", "t1", "t2", "t3", "t4"]
    for _, t in lt:
        assert t.shape == (24, 24) and t.flags.c_contiguous and t.base is None
        pos = [(y, x) for y in range(120 - 24 + 1) for x in range(200 - 24 + 1) if a[y, x] == t[0, 0] and np.array_equal(a[y:y + 24, x:x + 24], t)]
        assert pos


def test_gpus_flag_without_launcher_selects_the_device_group():
    """`python bench.py --gpus N` (no torch.distributed.run, WORLD_SIZE unset) is the single-process device-group mode
    (mtm_group) - round 2's bench refused to run that way.  Without a GPU it gets as far as counting the devices."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_GROUP_ALIAS")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode == 0:                       # a box with >= 2 GPUs: the line must say so
        import json
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["multi_gpu"]["mode"].startswith("one process")
    else:
        assert "torch.distributed.run" not in r.stderr
        assert "GPU(s) visible" in r.stderr or "no HIP device" in r.stderr or "No HIP" in r.stderr or "hip" in r.stderr.lower(), r.stderr[-400:]


def test_exchange_flag_and_fallback(monkeypatch):
    """--exchange: rccl is the default of the one-process multi-GPU form, host can be asked for; when the communicators
    cannot be created (a device listed twice, no librccl) the group keeps the host merge and the line says why."""
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    assert bench.parse().exchange is None                      # default: rccl when N > 1 (resolve_group_exchange)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--exchange", "host"])
    assert bench.parse().exchange == "host"

    class FakeGroup:
        def __init__(self, ok):
            self.ok, self.inits = ok, 0

        def __len__(self):
            return 8

        def comm_init(self, strict=True):
            self.inits += 1
            if not self.ok:
                raise RuntimeError("mtm_comm_init_all: device 0 is listed twice")
            return 8

    g = FakeGroup(True)
    assert bench.resolve_group_exchange(g, None)[:2] == ("rccl", 8) and g.inits == 1
    assert bench.resolve_group_exchange(g, "rccl")[:2] == ("rccl", 8)
    g = FakeGroup(True)
    assert bench.resolve_group_exchange(g, "host")[:2] == ("host", 0) and g.inits == 0
    kind, ranks, note = bench.resolve_group_exchange(FakeGroup(False), None)
    assert kind == "host" and ranks == 0 and "listed twice" in note
