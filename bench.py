#!/usr/bin/env python3
"""
bench.py - throughput of the MTM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
per-template score maps (window statistics + sliding-window NCC kernel), peak extraction, D2H of
the hit list, RCCL all-gather of hits (N > 1) and the global NMS - i.e. one MTM.matchTemplates call
minus the H2D upload of image and templates.

Default workload (weak scaling family of BASELINE.json's north_star target line):
    3840x2160 uint8 image x 32*N templates of 64x64, TM_CCOEFF_NORMED, score_threshold 0.5,
    maxOverlap 0.25; units sharded 32 per GPU (N=8 is BASELINE configs[3]).
Other configs (--config cfg2|cfg3|cfg5) are the parity-test cases of BASELINE.json.

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     - achieved algorithmic GB/s of the dominant (score-map) kernel against the 8 TB/s
                 HBM peak, kernel time measured with HIP events on the library's own stream;
  cpu_baseline - the CPU oracle (FFT-based restatement of the reference pipeline, thread pool over
                 templates as in MTM/__init__.py:172) timed on a bounded sample on this host.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
DOT4_PEAK_TMACS = 314.6        # 256 CU x 4 SIMD x 32 lanes x 4 MAC x 2.4 GHz (v_dot4_u32_u8 full rate)
# int8 MFMA, dense: MI355X_MICROARCH.md lists no spec figure for I8, "~2x the bf16 rate" (bf16 ~2.5 PF
# dense) and a micro-benchmark ceiling of >= 3944 TOPS.  One v_mfma_i32_16x16x64_i8 is 16384 MAC in 16
# cycles per SIMD: 1024 SIMDs x 2.4 GHz x 2048 op/cycle = 5.03 POPS.  `peak` below is that figure.
I8_MFMA_PEAK_TOPS = 5000.0
I8_MFMA_UBENCH_TOPS = 3944.0
PREWARM_STEPS = int(os.environ.get("BENCH_PREWARM", "120"))   # untimed, before the W warm-up steps: clock ramp (see main)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="north_star", help="north_star | cfg2 | cfg3 | cfg5")
    ap.add_argument("--kernel", default=os.environ.get("MTM_KERNEL", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true",
                    help="no software pipelining: the host part of a step finishes before the next step's GPU part starts")
    ap.add_argument("--skip-extras", action="store_true",
                    help="only the timed steps (profiling runs): no map-mode / end-to-end / stream side measurements")
    ap.add_argument("--cpu-sample-templates", type=int, default=0)
    return ap.parse_args()


def build_workload(name, n_gpus):
    import synth
    if name == "north_star":
        n_units = 32 * n_gpus
        noisy = 3 if n_units * 4 <= 800 else 1
        img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=n_units, templ=64,
                                                 noisy_per_unit=noisy)
        desc = "3840x2160 u8 image x %d templates 64x64 (32 per GPU), TM_CCOEFF_NORMED" % n_units
        method, thr = 5, 0.5
    else:
        img, units, plants = synth.make_config(name)
        method = 3 if name == "cfg5" else 5
        thr = 0.9 if name == "cfg5" else 0.5
        desc = "BASELINE %s: %dx%d u8 image x %d units" % (name, img.shape[1], img.shape[0], len(units))
    return img, units, plants, method, thr, desc


def algorithmic_bytes(img, units):
    """image + sum(template [+ mask] + 4 * (H-h+1) * (W-w+1)), maps materialised."""
    H, W = img.shape[:2]
    b = img.nbytes
    for u in units:
        t = u[1]
        b += t.nbytes + (u[2].nbytes if len(u) >= 3 else 0) + 4 * (H - t.shape[0] + 1) * (W - t.shape[1] + 1)
    return b


def algorithmic_bytes_hits_only(img, units):
    """Hits-only mode (no map consumers): image + templates [+ masks] + the two float64 window-statistics
    planes each size class reads (S1 and the guarded sqrt), nothing per (pixel, template) is written."""
    H, W = img.shape[:2]
    b = img.nbytes
    classes = set()
    for u in units:
        t = u[1]
        b += t.nbytes + (u[2].nbytes if len(u) >= 3 else 0)
        classes.add(t.shape[:2])
    for (h, w) in classes:
        b += 2 * 8 * (H - h + 1) * (W - w + 1)
    return b


def algorithmic_macs(img, units):
    H, W = img.shape[:2]
    m = 0
    for u in units:
        t = u[1]
        c = 1 if t.ndim == 2 else t.shape[2]
        m += (H - t.shape[0] + 1) * (W - t.shape[1] + 1) * t.shape[0] * t.shape[1] * c * (2 if len(u) >= 3 else 1)
    return m


def pmc_traffic(kernel_used, config, world, hits_only=False):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE, WRITE_SIZE; collected separately, see tools/profile_round.sh), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            table = json.load(f)
        key = "%s/%s/n%d%s" % ({2: "ncc_dot4_kernel", 3: "ncc_mfma_kernel"}.get(kernel_used, "?"), config, world,
                               "/hits_only" if hits_only else "")
        return table.get(key)
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(img, units, method, thr, n_sample):
    """The oracle's throughput port (kind 'port': float32-DFT correlation like cv2, shared image
    spectrum and window statistics, scipy.ndimage peaks, thread pool over templates with
    round(cpu_count/2) workers like the reference, MTM/__init__.py:172) on a bounded sample of the
    same workload.  Masked / multi-channel configs use the exact (slower) oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor
    import mtm_oracle as O
    cores = os.cpu_count() or 1
    workers = max(1, round(cores * 0.5))
    fast = img.ndim == 2 and method in (1, 3, 5) and all(len(u) == 2 and u[1].ndim == 2 for u in units)
    n_sample = n_sample or min(len(units), max(4, min(workers, 64 if fast else 16)))
    sample = units[:n_sample]
    t0 = time.perf_counter()
    if fast:
        fp = O.FastPipeline(img)
        one = lambda tup: fp.find(tup[0], tup[1], method, thr)      # noqa: E731
    else:
        one = lambda tup: O.find_matches([tup], img, method, float("inf"), thr)      # noqa: E731
    with ThreadPoolExecutor(max_workers=workers) as ex:
        hits = [h for part in ex.map(one, sample) for h in part]
    O.NMS(hits, thr, method == 1, float("inf"), 0.25)
    dt = time.perf_counter() - t0
    mpx = img.shape[0] * img.shape[1] * n_sample / 1e6
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"value": round(mpx / dt, 3), "unit": "Mpx-corr/s", "cores": workers, "kind": "port", "cpu_model": model,
            "host_cores": cores,
            "sample": "%d of %d templates on the full %dx%d image, %s (oracle/mtm_oracle.py), "
                      "%d worker threads of %d host cores, %.1f s"
                      % (n_sample, len(units), img.shape[1], img.shape[0],
                         "float32-DFT port of the cv2 pipeline" if fast else "exact float64 oracle",
                         workers, cores, dt),
            "seconds": round(dt, 2)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    # stdout is a protocol: exactly one JSON line, from rank 0.  Libraries print there too (gloo's
    # "[Gloo] Rank 0 is connected ..." at init, librccl's banner at teardown), so file descriptor 1 points
    # to stderr for the whole run and rank 0 writes its line to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch          # plumbing only: device sync + the distributed bootstrap/barrier
    import MTM
    from MTM import _lib
    from MTM.distributed import HitExchange, merge_and_nms, shard_units, unit_cost

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only
    # BENCH_FORCE_DEVICE: control-flow test of the N > 1 path on a single-GPU box (all ranks share one GPU;
    # RCCL refuses that, so the exchange falls back to gloo) - never set by the driver
    device = int(os.environ.get("BENCH_FORCE_DEVICE", local_rank))
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(device)

    img, units, plants, method, thr, desc = build_workload(args.config, world)
    ctx = _lib.Context(device)
    ctx.set_option(_lib.OPT_KERNEL, {"auto": 0, "naive": 1, "dot4": 2, "mfma": 3}[args.kernel])
    exchange_kind = "rccl" if world > 1 else "none"
    try:
        exchange = HitExchange("rccl" if world > 1 else "torch", rank, world, context=ctx)
    except Exception as e:  # noqa: BLE001 - keep the job alive: same records over gloo instead of RCCL
        sys.stderr.write("[bench] RCCL hit exchange unavailable (%s); falling back to gloo\n" % e)
        exchange = HitExchange("torch", rank, world)
        exchange_kind = "gloo-fallback"

    costs = [unit_cost(u[1], img.shape, len(u) >= 3) for u in units]
    mine = shard_units(costs, world)[rank]
    sub = [units[i] for i in mine]
    gidx = np.asarray(mine, dtype=np.int32)

    # inputs resident in HBM before the timed region
    ctx.set_image(img)
    ctx.set_templates([(u[1], u[2] if len(u) >= 3 else None) for u in sub], method)

    def sync():
        if have_torch_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    kernel_ms, total_ms, launches = [], [], 0

    # A step = GPU part (window statistics, score kernel, peak extraction, D2H of the hits), the all-gather of
    # the hit records, and the host part (merge in template order, global NMS, the reference's list of tuples).
    # The steps are software-pipelined: the GPU part of step i+1 is queued (mtm_find_matches_async: no waiting
    # for the GPU) before this thread does the host part of step i, the way
    # MTM.TemplateMatcher.match_stream overlaps the two for an image stream.  Every step is complete - its hit
    # list built - inside the timed region; `--sequential` turns the overlap off.
    def collect(raw, t):
        nonlocal launches
        kernel_ms.append(t["ncc_kernel_ms"])
        total_ms.append(t["total_ms"])
        launches = t["ncc_launches"]
        raw["templ_idx"] = gidx[raw["templ_idx"]]      # (the arrays the context hands out are the caller's own)
        return exchange.allgather(raw)

    def host_part(allhits):
        return merge_and_nms(allhits, units, method, float("inf"), thr, 0.25)

    def run_steps(k, pipelined=True):
        """k steps; returns the hit list and the timing record of the last one."""
        last = (None, None)
        if args.sequential or not pipelined:
            for _ in range(k):
                raw = ctx.find_matches(_lib.PEAKS_LOCAL, thr)
                t = ctx.timing()
                last = (host_part(collect(raw, t)), t)
            return last
        if k > 0:
            ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
        for i in range(k):
            raw = ctx.find_matches_wait()
            t = ctx.timing()
            if world == 1:                               # no collective: the context is free for the next step at once
                if i + 1 < k:
                    ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
                allhits = collect(raw, t)
            else:                                        # the all-gather uses the context's stream: before the next step
                allhits = collect(raw, t)
                if i + 1 < k:
                    ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
            last = (host_part(allhits), t)
        return last

    # The GPU leaves its idle clock only after ~50 ms of load (tools/ramp_probe.py: the first 40 calls run
    # 8 % slower than the steady state).  A fixed number of untimed steps - the same on every rank, the
    # step contains a collective - brings it to the sustained clock before the W warm-up steps.
    run_steps(PREWARM_STEPS)
    run_steps(args.warmup)
    kernel_ms.clear()
    total_ms.clear()
    # like timeit: no cyclic garbage collection inside the timed region (with torch imported a full
    # collection is a 30-40 ms pause that lands in one step at random)
    gc.collect()
    gc.disable()
    sync()
    t0 = time.perf_counter()
    hits, tinfo = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the same K steps one after the other (host part of a step before the GPU part of the next): reported
    # next to `value`
    seq_ms = None
    if world == 1 and not args.sequential and not args.skip_extras:
        km0 = len(kernel_ms)
        gc.disable()
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps, pipelined=False)
        sync()
        seq_ms = (time.perf_counter() - t1) / args.steps * 1e3
        gc.enable()
        del kernel_ms[km0:], total_ms[km0:]

    # the same steps with the score maps written to HBM (MTM_OPT_HITS_ONLY = 0): reported next to `value`
    maps_mode = None
    if world == 1 and tinfo.get("hits_only") and not args.skip_extras:
        ctx.set_option(_lib.OPT_HITS_ONLY, 0)
        run_steps(1)
        km0 = len(kernel_ms)
        sync()
        t1 = time.perf_counter()
        hits_m, _t = run_steps(args.steps)
        sync()
        dtm = time.perf_counter() - t1
        maps_mode = {"value": round(img.shape[0] * img.shape[1] * len(units) * args.steps / dtm / 1e6, 1),
                     "ms_per_step": round(dtm / args.steps * 1e3, 4),
                     "ncc_kernel_ms": round(float(np.mean(kernel_ms[km0:])), 4), "identical_hits": hits_m == hits,
                     "hits_only": int(_t["hits_only"])}
        del kernel_ms[km0 - 1:], total_ms[km0 - 1:]
        ctx.set_option(_lib.OPT_HITS_ONLY, 1)

    # PCIe-inclusive: one full MTM.matchTemplates call, numpy arrays in -> hit list out (never `value`)
    e2e_ms = stream_ms = None
    if world == 1 and not args.skip_extras:
        _lib._default_ctx = ctx
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            MTM.matchTemplates(units, img, method=method, score_threshold=thr, maxOverlap=0.25)
            ts.append((time.perf_counter() - t1) * 1e3)
        e2e_ms = float(np.median(ts))
        # image stream through resident templates (MTM.TemplateMatcher.match_stream): the upload of
        # image i+1 overlaps the kernels of image i; numpy arrays in -> hit lists out, per image
        matcher = MTM.TemplateMatcher(units, method=method, score_threshold=thr, maxOverlap=0.25, context=ctx)
        frames = [np.ascontiguousarray(np.roll(img, 64 * k, axis=1)) for k in range(4)] * 4
        list(matcher.match_stream(frames[:3]))        # warm-up: both image slots allocated
        stamps = [time.perf_counter()]
        for _ in matcher.match_stream(frames):
            stamps.append(time.perf_counter())
        # median of the per-image intervals: a Python garbage-collection pause (tens of ms once torch is
        # imported) lands in one interval and says nothing about the pipeline
        stream_ms = float(np.median(np.diff(stamps))) * 1e3

    # sanity: the timed path found every planted template
    found = {(h[0], h[1]) for h in hits}
    planted_ok = all((p[0], p[1]) in found for p in plants) if method == 5 else True

    if rank == 0:
        px = img.shape[0] * img.shape[1]
        value = px * len(units) * args.steps / dt / 1e6
        my_units = sub
        kms = float(np.mean(kernel_ms)) / max(launches, 1)          # avg duration of ONE ncc launch
        hits_only = bool(tinfo.get("hits_only", 0))
        bytes_launch = (algorithmic_bytes_hits_only if hits_only else algorithmic_bytes)(img, my_units) / max(launches, 1)
        macs = algorithmic_macs(img, my_units)
        achieved = bytes_launch / (kms * 1e-3) / 1e9
        kname = {1: "ncc_naive_kernel", 2: "ncc_dot4_kernel", 3: "ncc_mfma_kernel"}.get(tinfo["kernel_used"], "ncc_f64_kernel")
        tmacs = macs / (float(np.mean(kernel_ms)) * 1e-3) / 1e12
        traffic = pmc_traffic(tinfo["kernel_used"], args.config, world, hits_only)
        hbm = {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": int(bytes_launch),
               "maps_materialised": not hits_only}
        if tinfo["kernel_used"] == 3:
            # the dominant kernel runs on the int8 matrix cores: ~4096 MAC per output, ~1000 MAC per
            # algorithmic byte even when the maps are written - MFMA is the roofline that bounds it
            roof = {"bound": "mfma", "achieved": round(2.0 * tmacs, 1), "peak": I8_MFMA_PEAK_TOPS,
                    "unit": "TOP/s (int8 ops, 2 per MAC; the TFLOP/s slot of an integer kernel)",
                    "frac": round(2.0 * tmacs / I8_MFMA_PEAK_TOPS, 4), "traffic": traffic,
                    "ubench_ceiling": I8_MFMA_UBENCH_TOPS,
                    "frac_of_ubench_ceiling": round(2.0 * tmacs / I8_MFMA_UBENCH_TOPS, 4), "hbm": hbm}
        else:
            roof = dict(hbm, bound="hbm", traffic=traffic, valu_dot4_peak_tmacs=DOT4_PEAK_TMACS,
                        note="direct method, ~1000 MAC per algorithmic byte: VALU-bound by construction")
        roof.update({"kernel": kname, "kernel_ms_per_launch": round(kms, 4), "launches_per_step": launches,
                     "algorithmic_macs_per_launch": int(macs / max(launches, 1)), "achieved_tmacs": round(tmacs, 2)})
        out = {
            "metric": "Mpixel-correlations/s", "value": round(value, 1), "unit": "Mpx-corr/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "image_hw": list(img.shape[:2]), "units": len(units),
                       "units_per_gpu": len(my_units), "method": method, "score_threshold": thr,
                       "max_overlap": 0.25, "prewarm_steps": PREWARM_STEPS, "parallelism": "units sharded over %d rank(s), all-gather of hits: %s" % (world, exchange_kind),
                       "timed_region": "window statistics + correlation/normalisation kernel + peak extraction + D2H hits + "
                                       "all-gather + NMS + hit list; image/templates resident in HBM",
                       "pipelining": "none (--sequential)" if args.sequential else
                                     "the GPU part of step i+1 (mtm_find_matches_async) runs under the host part of step i "
                                     "(merge, NMS, hit list); all K hit lists are built inside the timed region",
                       "score_maps": "not materialised (hits-only mode, MTM_OPT_HITS_ONLY=1: identical hit lists)" if hits_only
                                     else "materialised in HBM"},
            "roofline": roof,
            "gpu_ms": {"kernels_total": round(float(np.mean(total_ms)), 4), "ncc_kernel": round(float(np.mean(kernel_ms)), 4)},
            "hits": len(hits), "planted_found": bool(planted_ok),
            "sequential_ms_per_step": None if seq_ms is None else round(seq_ms, 4),
            "score_maps_materialised": maps_mode,
            "e2e_call_ms": None if e2e_ms is None else round(e2e_ms, 3),
            "e2e_call_mpx_corr_s": None if e2e_ms is None else round(px * len(units) / e2e_ms / 1e3, 1),
            "stream_ms_per_image": None if stream_ms is None else round(stream_ms, 3),
            "stream_mpx_corr_s": None if stream_ms is None else round(px * len(units) / stream_ms / 1e3, 1),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(img, units, method, thr, args.cpu_sample_templates)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
