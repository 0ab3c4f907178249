#!/usr/bin/env python3
"""
bench.py - throughput of the MTM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

N > 1 runs either way: launched by ``torch.distributed.run --nproc-per-node N`` (WORLD_SIZE set) it is one process per
GPU with an RCCL all-gather of the hit records; started plainly (``python bench.py --gpus N``) it is ONE process
driving N GPUs through the library's device group (mtm_group: a context and a native worker thread per device,
units sharded by cost, host merge - no collective needed).

A "step" is ONE ``MTM.matchTemplates`` call as a user of the reference makes it: numpy
arrays in, list of hits out - argument validation, template hand-over (unchanged templates stay resident on the
GPU, as a loop over images leaves them), H2D of the image, window statistics, the sliding-window score kernel,
peak extraction, D2H of the hits, (N > 1: RCCL all-gather of the hit records) and the NMS.  ``value`` is
image pixels x templates x K / the wall-clock of the K timed calls; the per-call median is reported too.
The rate with the inputs already resident in HBM (software-pipelined and call by call), the rate with fresh
template bytes in every call, the image-stream rate and the rate with materialised score maps are extra keys
of the same line.

Default workload (weak-scaling family of BASELINE.json's north_star target line):
    3840x2160 uint8 image x 32*N templates of 64x64, TM_CCOEFF_NORMED, score_threshold 0.5,
    maxOverlap 0.25; units sharded 32 per GPU (N = 8 is BASELINE configs[3]).
Other configs (--config cfg2|cfg3|cfg4|cfg5) are the parity-test cases of BASELINE.json on ONE GPU.

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     - the dominant (score-map) kernel against the dense int8 MFMA peak: 2 x algorithmic MACs of a step /
                 the kernel's time in that step, measured with HIP events on the library's own streams inside
                 the timed region; the shader clock the kernel actually ran at, measured inside the kernel;
                 the HBM view (algorithmic bytes against 8 TB/s) and the PMC-measured HBM traffic;
  cpu_baseline - the CPU oracle (FFT-based restatement of the reference pipeline, thread pool over
                 templates as in MTM/__init__.py:172) timed on a bounded sample on this host.
"""
import argparse
import gc
import glob
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
# tools/ubench/power (round 3): NOTHING but v_mfma_i32_16x16x64_i8, two waves per SIMD, every CU - 4870 TOP/s at 2.39 GHz on
# all-zero operands, 3960 TOP/s at 2.00 GHz on random operands that change with every instruction (what a correlation of
# white-noise bytes feeds the multipliers): the chip's power budget, not the issue rate, caps random-data int8 work there.
I8_MFMA_RANDOM_OPERANDS_TOPS = 3960.0
I8_OPS_PER_CLK = 1024 * 2048   # whole chip, per shader cycle
PREWARM_SECONDS = float(os.environ.get("BENCH_PREWARM_S", "0.6"))   # untimed load before the W warm-up calls (clock ramp)
CONFIGS = ("north_star", "cfg2", "cfg3", "cfg4", "cfg5")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="north_star", choices=CONFIGS)
    ap.add_argument("--kernel", default=os.environ.get("MTM_KERNEL", "auto"))
    ap.add_argument("--exchange", default=None, choices=["rccl", "host"],
                    help="hit exchange of the one-process multi-GPU form (--gpus N without a launcher): rccl = ncclCommInitAll "
                         "over the devices + one all-gather per device inside ncclGroupStart/End (default when N > 1), "
                         "host = merge of the workers' host lists")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-extras", action="store_true",
                    help="only the timed calls (profiling runs): no resident / fresh-template / map-mode / stream side measurements")
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


def score_kernel_macs(img, units):
    """Multiply-accumulates of the launches the library's kernel timer brackets (mtm_timing.ncc_kernel_ms): the
    template correlations sum I*T (or I*(T*M)), out_px * w * h * C per unit.  The second correlation of a masked
    unit, sum I^2*M, is computed ONCE per distinct mask by separate launches of the statistics phase, outside that
    timer: it is counted in `masked_stat_macs`, never in the roofline of the timed kernel."""
    H, W = img.shape[:2]
    m = 0
    for u in units:
        t = u[1]
        c = 1 if t.ndim == 2 else t.shape[2]
        m += (H - t.shape[0] + 1) * (W - t.shape[1] + 1) * t.shape[0] * t.shape[1] * c
    return m


def masked_stat_macs(img, units):
    H, W = img.shape[:2]
    seen, m = set(), 0
    for u in units:
        if len(u) >= 3:
            key = (u[1].shape[:2], u[2].tobytes())
            if key not in seen:
                seen.add(key)
                m += 2 * (H - u[1].shape[0] + 1) * (W - u[1].shape[1] + 1) * u[1].shape[0] * u[1].shape[1]   # two byte planes of I^2
    return m


def pmc_traffic(kernel_used, config, world, hits_only=False):
    """HBM bytes per FULL-IMAGE launch of the dominant kernel from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE, WRITE_SIZE; collected separately, see tools/profile_round.sh), or None.  A table
    look-up, not a measurement of this run."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            table = json.load(f)
        key = "%s/%s/n%d%s" % ({2: "ncc_dot4_kernel", 3: "ncc_mfma_kernel"}.get(kernel_used, "?"), config, world,
                               "/hits_only" if hits_only else "")
        if key in table:
            return table[key]
        # the other BASELINE configs: the dominant kernel of the workload's own PMC passes (tools/profile_workloads.sh):
        # bytes of ONE launch of that kernel (a config with several size classes has several such launches per step)
        wl = table.get("workload/%s" % config)
        return wl["hbm_bytes_per_launch"] if isinstance(wl, dict) and world == 1 else None
    except Exception:  # noqa: BLE001
        return None


def gpu_sensors():
    """Engine clock (MHz) and power (W) the driver reports right now (sysfs hwmon of the first amdgpu card), or {}."""
    out = {}
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        try:
            with open(os.path.join(hw, "freq1_input")) as f:
                out["sclk_mhz_sysfs"] = round(int(f.read()) / 1e6, 1)
            for name in ("power1_average", "power1_input"):
                pth = os.path.join(hw, name)
                if os.path.exists(pth):
                    with open(pth) as f:
                        out["power_w"] = round(int(f.read()) / 1e6, 1)
                    break
            break
        except (OSError, ValueError):
            continue
    return out


def cpu_baseline(img, units, method, thr, n_sample):
    """The reference pipeline on this host's cores, one task per template on round(cpu_count / 2) threads like the
    reference's pool (MTM/__init__.py:172), on a bounded sample of the same workload; kind 'port' - cv2 / skimage are
    not installable here.  uint8 single-channel unmasked workloads (the headline) run oracle/libmtm_cpu.so: a C++ port
    with cv2's structure (block-wise float32 DFT correlation, float64 integral images, per-pixel normalisation, 3x3
    maximum filter) whose image spectra and integral images are shared across templates - faster than the
    reference's own structure, never slower.  Masked / multi-channel configs use the numpy oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtm_oracle as O
    cores = os.cpu_count() or 1
    workers = max(1, round(cores * 0.5))
    cxx = (img.ndim == 2 and img.dtype == np.uint8 and method in (1, 3, 5) and
           all(len(u) == 2 and u[1].ndim == 2 and max(u[1].shape) <= 256 for u in units))
    impl = None
    if cxx:
        try:
            import mtm_cpu
            mtm_cpu.load()
            impl = "cxx"
        except Exception as e:  # noqa: BLE001 - no compiler on this host: fall back to the numpy port
            sys.stderr.write("[bench] oracle/libmtm_cpu.so unavailable (%s): numpy port instead\n" % e)
    fast = img.ndim == 2 and method in (1, 3, 5) and all(len(u) == 2 and u[1].ndim == 2 for u in units)
    n_sample = n_sample or min(len(units), max(4, min(workers, 64 if fast else 16)))
    sample = units[:n_sample]
    t0 = time.perf_counter()
    reps = []
    if impl == "cxx":
        # ~0.5 s per pass: several passes (about 10 s of CPU work in total), the median pass is reported
        for _ in range(7):
            t1 = time.perf_counter()
            hits, info = mtm_cpu.find_matches(sample, img, method, thr, n_threads=workers)
            O.NMS(hits, thr, method == 1, float("inf"), 0.25)
            reps.append(time.perf_counter() - t1)
            if time.perf_counter() - t0 > 20.0:
                break
        what = "C++ port of the cv2 pipeline (oracle/cpu/mtm_cpu.cpp: 512x512 float32 DFT blocks, shared image spectra)"
    else:
        from concurrent.futures import ThreadPoolExecutor
        if fast:
            fp = O.FastPipeline(img)
            one = lambda tup: fp.find(tup[0], tup[1], method, thr)      # noqa: E731
        else:
            one = lambda tup: O.find_matches([tup], img, method, float("inf"), thr)      # noqa: E731
        with ThreadPoolExecutor(max_workers=workers) as ex:
            hits = [h for part in ex.map(one, sample) for h in part]
        what = "numpy float32-DFT port of the cv2 pipeline" if fast else "exact float64 numpy oracle"
    O.NMS(hits, thr, method == 1, float("inf"), 0.25)
    dt = time.perf_counter() - t0
    if reps:
        dt = float(np.median(reps))
        what += ", median of %d passes" % len(reps)
    mpx = img.shape[0] * img.shape[1] * n_sample / 1e6
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    used = min(workers, n_sample) if impl == "cxx" else workers          # the C++ port runs one thread per template
    return {"value": round(mpx / dt, 3), "unit": "Mpx-corr/s", "cores": used,
            "kind": "port", "cpu_model": model, "host_cores": cores,
            "sample": "%d of %d templates on the full %dx%d image, %s, %d worker threads of %d host cores, %.2f s"
                      % (n_sample, len(units), img.shape[1], img.shape[0], what, used, cores, dt),
            "seconds": round(dt, 3)}


def resolve_group_exchange(group, requested):
    """The hit exchange of the one-process multi-GPU form.  Default (None) and "rccl": ncclCommInitAll over the group's
    devices; if that fails (a device listed twice - BENCH_GROUP_ALIAS on a single-GPU box -, no librccl) the group keeps
    the host merge and the line says why.  Returns (kind in use, ncclCommCount, note)."""
    if requested == "host":
        return "host", 0, "host merge requested"
    try:
        ranks = group.comm_init(strict=True)
        return "rccl", ranks, "ncclCommInitAll over %d device(s)" % len(group)
    except Exception as e:  # noqa: BLE001
        return "host", 0, "rccl unavailable, host merge instead: %s" % e


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # --gpus N without a launcher: one process, N devices (mtm_group).  With WORLD_SIZE set: one process per GPU.
    # (BENCH_GROUP_SINGLE=1: the group path with ONE device - on a one-GPU box the only way to run the in-process RCCL
    # exchange end to end, as a communicator of one rank; never set by the driver)
    group_n = args.gpus if ("WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BENCH_GROUP_SINGLE"))) else 0
    if not group_n and world != args.gpus:
        args.gpus = world

    # stdout is a protocol: exactly one JSON line, from rank 0.  Libraries print there too (gloo's
    # "[Gloo] Rank 0 is connected ..." at init, librccl's banner at teardown), so file descriptor 1 points
    # to stderr for the whole run and rank 0 writes its line to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch          # the driver's contract: device synchronisation + torch.distributed barrier / max-over-ranks
    import MTM
    from MTM import _lib
    from MTM.distributed import HitExchange, merge_and_nms, shard_units, unit_cost

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only
    # BENCH_FORCE_DEVICE: control-flow test of the N > 1 path on a single-GPU box (all ranks share one GPU;
    # RCCL refuses that, so the exchange falls back to the control plane) - never set by the driver
    device = int(os.environ.get("BENCH_FORCE_DEVICE", local_rank))
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(device)

    img, units, plants, method, thr, desc = build_workload(args.config, group_n or world)
    group, group_devices = None, []
    if group_n:
        n_vis = _lib.load().mtm_device_count()
        # BENCH_GROUP_ALIAS=1: more contexts than GPUs (control-flow test on a single-GPU box) - never set by the driver
        if n_vis < group_n and not os.environ.get("BENCH_GROUP_ALIAS"):
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (group_n, n_vis))
        group_devices = [i % max(n_vis, 1) for i in range(group_n)]
        if group_n == 1:
            os.environ["MTM_DEVICES_FORCE_GROUP"] = "1"      # (a one-device list normally resolves to a plain context)
        group = _lib.Group(group_devices)
        group.set_option(_lib.OPT_KERNEL, {"auto": 0, "naive": 1, "dot4": 2, "mfma": 3}[args.kernel])
        _lib._engines[tuple(group_devices)] = group          # MTM.matchTemplates(devices=...) below runs on this group
        ctx = None
        group_exchange, group_comm_ranks, group_exchange_note = resolve_group_exchange(group, args.exchange)
    else:
        ctx = _lib.Context(device)
        ctx.set_option(_lib.OPT_KERNEL, {"auto": 0, "naive": 1, "dot4": 2, "mfma": 3}[args.kernel])
        _lib._default_ctx = ctx                  # MTM.matchTemplates below runs on this context
    exchange_kind = "rccl" if world > 1 else ("none" if not group_n else
                                              "rccl all-gather inside ncclGroupStart/End (one process, ncclCommInitAll)"
                                              if group_exchange == "rccl" else "host merge (one process, mtm_group)")
    comm_ranks = 1

    class _ControlPlaneStore:
        """The RCCL unique id travels over the gloo group that exists anyway (the package's own TcpStore would listen on a
        second port, MASTER_PORT + 1 - one more thing that can be taken on a shared node)."""
        def broadcast(self, payload=None):
            box = [payload]
            dist.broadcast_object_list(box, src=0)
            return box[0]

    rccl_error = None
    try:
        exchange = (HitExchange("rccl", rank, world, context=ctx, store=_ControlPlaneStore() if world > 1 else None)
                    if not group_n else None)
        if world > 1:
            comm_ranks = int(getattr(ctx, "n_ranks", 1))           # ranks of the RCCL communicator that was created
    except Exception as e:  # noqa: BLE001 - keep the job alive: same records over gloo instead of RCCL
        rccl_error = e
    if world > 1:
        # every rank takes the same exchange: one that failed to join the communicator would otherwise wait in a gloo
        # all-gather while the others wait in RCCL's
        flags = [None] * world
        dist.all_gather_object(flags, rccl_error is None)
        if rccl_error is None and not all(flags):
            rccl_error = RuntimeError("rank(s) %s could not join the communicator" % [i for i, ok in enumerate(flags) if not ok])
    if rccl_error is not None:
        sys.stderr.write("[bench] RCCL hit exchange unavailable (%s); falling back to gloo\n" % rccl_error)

        def gloo_allgather(payload):
            parts = [None] * world
            dist.all_gather_object(parts, payload)
            return parts
        exchange = HitExchange("custom", rank, world, allgather_bytes=gloo_allgather)
        exchange_kind = "gloo-fallback"

    costs = [unit_cost(u[1], img.shape, len(u) >= 3) for u in units]
    mine = shard_units(costs, group_n or world)[0 if group_n else rank]     # group: device 0's shard (same LPT rule in the library)
    sub = [units[i] for i in mine]
    gidx = np.asarray(mine, dtype=np.int32)
    inf = float("inf")

    def sync():
        if have_torch_gpu:
            for d in (sorted(set(group_devices)) if group_n else [device]):
                torch.cuda.synchronize(d)
        if dist is not None:
            dist.barrier()

    kernel_ms, total_ms, clocks, sum_ms, launches = [], [], [], [], 0
    masked_ms, masked_launches = [], 0                   # sum I^2 M passes of masked classes (their own event pairs)
    dev_kernel_ms = [[] for _ in range(group_n)]         # group: score-kernel time per device and step

    def note_timing():
        nonlocal launches, masked_launches
        if group_n:
            for i in range(group_n):
                dev_kernel_ms[i].append(group.timing(i)["ncc_kernel_ms"])
        t = group.timing(0) if group_n else ctx.timing()
        kernel_ms.append(t["ncc_kernel_ms"])
        sum_ms.append(t["ncc_sum_ms"])
        total_ms.append(t["total_ms"])
        if t["sclk_mhz"] > 0:
            clocks.append(t["sclk_mhz"])
        launches = t["ncc_launches"]
        masked_ms.append(t.get("masked_stat_ms", 0.0))
        masked_launches = t.get("sq_launches", 0)
        return t

    # ---- the timed step: one matchTemplates call, numpy arrays in -> hit list out
    if group_n:
        def call(lt=units):
            return MTM.matchTemplates(lt, img, method=method, score_threshold=thr, maxOverlap=0.25, devices=group_devices)
    elif world == 1:
        def call(lt=units):
            return MTM.matchTemplates(lt, img, method=method, score_threshold=thr, maxOverlap=0.25)
    else:
        from MTM.distributed import _u8_units
        sub_units = _u8_units(sub, img, method) if exchange_kind == "rccl" else None

        def call(lt=None):
            # the same call with the units sharded over the ranks: this rank's templates + the image go to its GPU,
            # hit records are exchanged (RCCL all-gather), every rank runs the global NMS - one native call per step
            # (mtm_find_matches_image_sharded_nms); over the control-plane fallback: step by step from here
            if sub_units is not None:
                return MTM._to_hit_list(ctx.search_sharded_nms(sub_units, img, method, thr, 0.25, -1, gidx), units, 0, 0)
            raw = MTM._raw_matches(sub, img, method, inf, thr, context=ctx).copy()
            raw["templ_idx"] = gidx[raw["templ_idx"]]
            return merge_and_nms(exchange.allgather(raw), units, method, inf, thr, 0.25)

    def run_calls(k, lt_of=None, stamps=None):
        hits = None
        for i in range(k):
            t1 = time.perf_counter()
            hits = call(lt_of(i)) if lt_of else call()
            if stamps is not None:
                stamps.append(time.perf_counter() - t1)
            note_timing()
        return hits

    # ---- resident-input steps (image and templates in HBM before the clock starts), software-pipelined: the GPU
    # part of step i+1 is queued (mtm_find_matches_async) before this thread does the host part of step i
    def collect(raw):
        raw["templ_idx"] = gidx[raw["templ_idx"]]
        return exchange.allgather(raw)

    def host_part(allhits):
        return merge_and_nms(allhits, units, method, inf, thr, 0.25)

    def run_resident(k, pipelined=True):
        last = None
        if not pipelined:
            for _ in range(k):
                raw = ctx.find_matches(_lib.PEAKS_LOCAL, thr)
                note_timing()
                last = host_part(collect(raw))
            return last
        if k > 0:
            ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
        for i in range(k):
            raw = ctx.find_matches_wait()
            note_timing()
            if world == 1:                               # no collective: the context is free for the next step at once
                if i + 1 < k:
                    ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
                allhits = collect(raw)
            else:                                        # the all-gather uses the context's stream: before the next step
                allhits = collect(raw)
                if i + 1 < k:
                    ctx.find_matches_async(_lib.PEAKS_LOCAL, thr)
            last = host_part(allhits)
        return last

    # The GPU leaves its idle clock only after tens of ms of load and then settles to its power budget.  Untimed
    # calls for a fixed TIME (the same on every rank: the step contains a collective, so the count is agreed on
    # rank 0's clock) bring it to the sustained state before the W warm-up calls.
    call()
    # like timeit: no cyclic garbage collection inside the timed region (with torch imported a full collection is a
    # 30-40 ms pause that lands in one step at random).  Collected HERE, before the pre-warm: a pause of that length
    # between the warm-up and the timed calls lets the clock fall back, and K = 20 timed calls (20 ms) would all run
    # in the ramp (measured: 0.97 instead of 0.90 ms per call).
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    n_pre = 0
    while True:
        run_calls(8)
        n_pre += 8
        go_on = (time.perf_counter() - t0) < PREWARM_SECONDS
        if dist is not None:
            flag = torch.tensor([1 if go_on else 0])
            dist.broadcast(flag, src=0)
            go_on = bool(flag.item())
        if not go_on:
            break
    run_calls(args.warmup)
    kernel_ms.clear(), total_ms.clear(), clocks.clear(), sum_ms.clear()
    sensors_before = gpu_sensors()
    per_call = []
    sync()
    t0 = time.perf_counter()
    hits = run_calls(args.steps, stamps=per_call)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    sensors_after = gpu_sensors()
    tinfo = group.timing(0) if group_n else ctx.timing()
    k_ms, t_ms, clk, timed_launches = list(kernel_ms), list(total_ms), list(clocks), tinfo["ncc_launches"]
    s_ms = list(sum_ms)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    px = img.shape[0] * img.shape[1]
    rate = lambda ms: round(px * len(units) / ms / 1e3, 1)        # noqa: E731  ms per call -> Mpx-corr/s
    extras = {}
    if world == 1 and not group_n and not args.skip_extras:
        gc.disable()
        # (a) fresh template bytes in every call: nothing derived from the previous call's templates is reused
        variants = []
        for i in range(4):
            lt = []
            for (name, t, *rest) in units:
                t2 = t.copy()
                t2[0, 0] ^= (i + 1)                      # one different byte: a different template set for the library
                lt.append((name, t2) + tuple(rest))
            variants.append(lt)
        run_calls(2, lt_of=lambda i: variants[i % 4])
        st = []
        run_calls(max(10, min(args.steps, 40)), lt_of=lambda i: variants[i % 4], stamps=st)
        extras["fresh_templates"] = {"median_ms_per_call": round(float(np.median(st)) * 1e3, 4),
                                     "value": rate(float(np.median(st)) * 1e3),
                                     "note": "every call gets template bytes the library has not seen in the previous call: "
                                             "statistics, packing and upload of the templates are inside the call"}
        call()
        # (a2) the same per-call metric with the image in page-locked memory (MTM.pinned_empty): no staging copy, the
        # rows cross PCIe as plain DMA transfers behind the call
        pimg = MTM.pinned_empty(img.shape, img.dtype)
        pimg[...] = img
        for _ in range(3):
            hp = MTM.matchTemplates(units, pimg, method=method, score_threshold=thr, maxOverlap=0.25)
        st = []
        for _ in range(max(10, min(args.steps, 40))):
            t1 = time.perf_counter()
            hp = MTM.matchTemplates(units, pimg, method=method, score_threshold=thr, maxOverlap=0.25)
            st.append(time.perf_counter() - t1)
            note_timing()
        extras["pinned_image"] = {"median_ms_per_call": round(float(np.median(st)) * 1e3, 4),
                                  "value": rate(float(np.median(st)) * 1e3), "identical_hits": hp == hits,
                                  "note": "the image array lives in page-locked host memory (MTM.pinned_empty)"}
        del pimg
        call()
        # (a3) the epilogue's division: IEEE division is the default since round 5 (bit-identical to the oracle); the same
        # per-call metric with the reciprocal normalisation of rounds 1-4 (<= 1 ulp(float32) on ~1e-8 of the outputs)
        ctx.set_option(_lib.OPT_EXACT_DIV, 0)
        for _ in range(3):
            hq = MTM.matchTemplates(units, img, method=method, score_threshold=thr, maxOverlap=0.25)
        st = []
        for _ in range(max(10, min(args.steps, 40))):
            t1 = time.perf_counter()
            hq = MTM.matchTemplates(units, img, method=method, score_threshold=thr, maxOverlap=0.25)
            st.append(time.perf_counter() - t1)
            note_timing()
        ctx.set_option(_lib.OPT_EXACT_DIV, 1)
        extras["reciprocal_normalisation"] = {"median_ms_per_call": round(float(np.median(st)) * 1e3, 4),
                                              "value": rate(float(np.median(st)) * 1e3),
                                              "same_boxes": [(h[0], h[1]) for h in hq] == [(h[0], h[1]) for h in hits],
                                              "note": "MTM_OPT_EXACT_DIV = 0; the default (`value`, median_ms_per_call) divides"}
        call()
        # (b) inputs resident in HBM: round 1's headline (pipelined) and the same call by call
        ctx.set_image(img)
        ctx.set_templates([(u[1], u[2] if len(u) >= 3 else None) for u in sub], method)
        run_resident(3)
        km0 = len(kernel_ms)
        sync()
        t1 = time.perf_counter()
        hits_r = run_resident(args.steps)
        sync()
        d_pipe = (time.perf_counter() - t1) / args.steps * 1e3
        res_kernel = float(np.mean(kernel_ms[km0:])) / max(launches, 1)
        res_total = float(np.mean(total_ms[km0:]))
        res_clk = [c for c in clocks[-args.steps:]]
        sync()
        t1 = time.perf_counter()
        run_resident(args.steps, pipelined=False)
        sync()
        d_seq = (time.perf_counter() - t1) / args.steps * 1e3
        extras["resident_inputs"] = {"pipelined_ms_per_step": round(d_pipe, 4), "pipelined_value": rate(d_pipe),
                                     "sequential_ms_per_step": round(d_seq, 4), "sequential_value": rate(d_seq),
                                     "kernel_ms_per_launch": round(res_kernel, 4), "launches_per_step": launches,
                                     "gpu_ms_kernels_total": round(res_total, 4), "identical_hits": hits_r == hits,
                                     "sclk_mhz_in_kernel": round(float(np.median(res_clk)), 1) if res_clk else None,
                                     "note": "image and templates in HBM before the clock starts (round 1's `value`): "
                                             "statistics + score kernel + peaks + D2H hits + NMS + hit list"}
        # (c) the same with the score maps written to HBM (MTM_OPT_HITS_ONLY = 0)
        if tinfo.get("hits_only"):
            ctx.set_option(_lib.OPT_HITS_ONLY, 0)
            run_resident(2)
            km0 = len(kernel_ms)
            sync()
            t1 = time.perf_counter()
            hits_m = run_resident(args.steps)
            sync()
            dtm = (time.perf_counter() - t1) / args.steps * 1e3
            extras["score_maps_materialised"] = {"ms_per_step": round(dtm, 4), "value": rate(dtm),
                                                 "ncc_kernel_ms": round(float(np.mean(kernel_ms[km0:])), 4),
                                                 "identical_hits": hits_m == hits, "inputs": "resident, pipelined"}
            ctx.set_option(_lib.OPT_HITS_ONLY, 1)
        # (d) image stream through resident templates (MTM.TemplateMatcher.match_stream): the upload of image i+1
        # overlaps the kernels of image i; numpy arrays in -> hit lists out, per image
        matcher = MTM.TemplateMatcher(units, method=method, score_threshold=thr, maxOverlap=0.25, context=ctx)
        frames = [np.ascontiguousarray(np.roll(img, 64 * k, axis=1)) for k in range(4)] * 4
        list(matcher.match_stream(frames[:3]))        # warm-up: both image slots allocated
        stamps = [time.perf_counter()]
        for _ in matcher.match_stream(frames):
            stamps.append(time.perf_counter())
        sm = float(np.median(np.diff(stamps))) * 1e3
        extras["image_stream"] = {"median_ms_per_image": round(sm, 4), "value": rate(sm)}
        # (e) the other regime: an image with the statistics of a photograph (synth.smooth_u8) and templates cut from
        # it.  Its score maps are smooth - at this threshold thousands of pixels per template lie above it, the
        # candidate list overflows and the library settles on map mode + the full peak pass; the raw peaks number in
        # the thousands, so sorting and NMS on the host count too.  (Last: it leaves the context in its back-off state.)
        if img.ndim == 2 and img.dtype == np.uint8 and all(len(u) == 2 for u in units):
            import synth
            simg = synth.smooth_u8(11, img.shape)
            sunits = synth.cut_templates(5, simg, len(units), int(units[0][1].shape[0]))
            for _ in range(6):
                hs = MTM.matchTemplates(sunits, simg, method=method, score_threshold=thr, maxOverlap=0.25)
            st, routes, gms = [], [], []
            for _ in range(12):
                t1 = time.perf_counter()
                hs = MTM.matchTemplates(sunits, simg, method=method, score_threshold=thr, maxOverlap=0.25)
                st.append(time.perf_counter() - t1)
                tms = ctx.timing()
                routes.append(int(tms["hits_only"]))
                gms.append(float(tms["total_ms"]))
            sm = float(np.median(st)) * 1e3
            route = max(set(routes), key=routes.count)          # (every 17th call or so retries the candidate list: route 0)
            extras["photograph_like_image"] = {"median_ms_per_call": round(sm, 4), "value": rate(sm), "hits": len(hs),
                                               "peaks_before_nms": int(tms["n_hits"]), "hits_only": route,
                                               "gpu_ms": round(float(np.median(gms)), 4),
                                               "note": "smooth score maps, thousands of raw peaks: hits_only = 1 -> the candidate list "
                                                       "held them (device-side hash verification); 0 -> the list overflowed: maps in "
                                                       "memory + full peak pass; 2 -> the calls after an overflow: maps in memory, the "
                                                       "score kernel flags the row segments that hold something above the threshold, "
                                                       "the peak pass visits those.  Host sort + NMS of the raw peaks included"}
            ctx.set_option(_lib.OPT_HITS_ONLY, 1)          # clears the back-off
            # (d) the reference's own published benchmark shape (tutorials/Benchmark.ipynb: one 414 x 400 template over a
            # 2048 x 2048 image; 264 ms per call there): large templates run as slabs of one launch on the same kernel
            limg = synth.smooth_u8(21, (2048, 2048))
            lunits = [("big", np.ascontiguousarray(limg[300:700, 500:914]))]
            for _ in range(4):
                hl = MTM.matchTemplates(lunits, limg, method=method, score_threshold=0.9, maxOverlap=0.25)
            st = []
            for _ in range(12):
                t1 = time.perf_counter()
                hl = MTM.matchTemplates(lunits, limg, method=method, score_threshold=0.9, maxOverlap=0.25)
                st.append(time.perf_counter() - t1)
            lm = float(np.median(st)) * 1e3
            tml = ctx.timing()
            extras["large_template_414x400_over_2048x2048"] = {
                "median_ms_per_call": round(lm, 4), "hits": len(hl), "gpu_ms": round(float(tml["total_ms"]), 4),
                "slab_launch_ms": round(float(tml["ncc_kernel_ms"]), 4),
                "value": round(2048 * 2048 / lm / 1e3, 1),
                "note": "Mpx-corr/s of this shape (image pixels x 1 template per second); 7 slabs of 64 taps, one launch"}
            # (f) the same workload as float32 pixels - what the reference turns every non-uint8 input into
            # (MTM/__init__.py:71-74): bf16 matrix cores as a screen (round 6: one piece product where only a list leaves the
            # kernel, two upload bands), exact float64 re-scoring of what the screen lists, the float64 kernel's records
            fimg = img.astype(np.float32) * np.float32(0.731) + np.float32(3.25)
            funits = [(u[0], u[1].astype(np.float32) * np.float32(0.731) + np.float32(3.25)) for u in units]
            for _ in range(4):
                hf = MTM.matchTemplates(funits, fimg, method=method, score_threshold=thr, maxOverlap=0.25)
            st = []
            for _ in range(12):
                t1 = time.perf_counter()
                hf = MTM.matchTemplates(funits, fimg, method=method, score_threshold=thr, maxOverlap=0.25)
                st.append(time.perf_counter() - t1)
            fm = float(np.median(st)) * 1e3
            tmf = ctx.timing()
            extras["float32_image"] = {
                "median_ms_per_call": round(fm, 4), "value": rate(fm), "hits": len(hf),
                "same_boxes_as_uint8": sorted((h[0], h[1]) for h in hf) == sorted((h[0], h[1]) for h in hits),
                "gpu_ms": round(float(tmf["total_ms"]), 4), "bf16_kernel_ms": round(float(tmf["ncc_kernel_ms"]), 4),
                "launches": int(tmf["ncc_launches"]), "f32_route": int(tmf["f32_route"]), "f32_pieces": int(tmf["f32_pieces"]),
                "note": "float32 pixels and templates (an affine map of the uint8 workload), numpy arrays in -> hit list out; "
                        "f32_route 1 = kernel candidates re-scored with the float64 chain, f32_pieces 1 = the one-product screen"}
        gc.enable()

    # sanity: the timed path found every planted template
    found = {(h[0], h[1]) for h in hits}
    planted_ok = all((p[0], p[1]) in found for p in plants) if method == 5 else True

    if rank == 0:
        value = px * len(units) * args.steps / dt / 1e6
        my_units = sub
        launches = timed_launches
        k_step = float(np.mean(k_ms))                               # score-kernel time of one step (all its launches)
        if not k_step > 0.0:                                        # (MTM_NCC_EVENTS=0, an experiment: no kernel timing -> rates of 0)
            k_step = float("inf")
        hits_only = bool(tinfo.get("hits_only", 0))
        bytes_step = (algorithmic_bytes_hits_only if hits_only else algorithmic_bytes)(img, my_units)
        macs = score_kernel_macs(img, my_units)
        achieved = bytes_step / (k_step * 1e-3) / 1e9
        kname = {1: "ncc_naive_kernel", 2: "ncc_dot4_kernel", 3: "ncc_mfma_kernel"}.get(tinfo["kernel_used"], "ncc_f64_kernel")
        tmacs = macs / (k_step * 1e-3) / 1e12
        traffic = pmc_traffic(tinfo["kernel_used"], args.config, group_n or world, hits_only)
        hbm = {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(bytes_step),
               "maps_materialised": not hits_only}
        sclk = float(np.median(clk)) if clk else None
        if tinfo["kernel_used"] == 3:
            # the dominant kernel runs on the int8 matrix cores: ~4096 MAC per output, ~1000 MAC per
            # algorithmic byte even when the maps are written - MFMA is the roofline that bounds it
            roof = {"bound": "mfma", "achieved": round(2.0 * tmacs, 1), "peak": I8_MFMA_PEAK_TOPS,
                    "unit": "TOP/s (int8 ops, 2 per MAC; the TFLOP/s slot of an integer kernel)",
                    "frac": round(2.0 * tmacs / I8_MFMA_PEAK_TOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes of one full-image launch from the committed PMC passes (profiles/), not of this run",
                    "ubench_ceiling": I8_MFMA_UBENCH_TOPS,
                    "frac_of_ubench_ceiling": round(2.0 * tmacs / I8_MFMA_UBENCH_TOPS, 4),
                    "power_limited_ceiling": I8_MFMA_RANDOM_OPERANDS_TOPS,
                    "power_limited_ceiling_note": "pure MFMA stream on random operands (tools/ubench/power): 2.0 GHz under the "
                                                  "power budget instead of 2.4 GHz",
                    "frac_of_power_limited_ceiling": round(2.0 * tmacs / I8_MFMA_RANDOM_OPERANDS_TOPS, 4), "hbm": hbm}
            if sclk:
                peak_at_clk = I8_OPS_PER_CLK * sclk * 1e6 / 1e12
                roof.update({"sclk_mhz_in_kernel": round(sclk, 1),
                             "peak_at_measured_clock": round(peak_at_clk, 1),
                             "frac_at_measured_clock": round(2.0 * tmacs / peak_at_clk, 4)})
        else:
            roof = dict(hbm, bound="hbm", traffic=traffic, valu_dot4_peak_tmacs=DOT4_PEAK_TMACS,
                        note="direct method, ~1000 MAC per algorithmic byte: VALU-bound by construction")
        roof.update({"kernel": kname, "kernel_ms_per_launch": round(float(np.mean(s_ms)) / max(launches, 1), 4),
                     "launches_per_step": launches,
                     "kernel_ms_per_step": round(k_step, 4),
                     "algorithmic_macs_per_step": int(macs), "algorithmic_macs_per_launch": int(macs / max(launches, 1)),
                     "achieved_tmacs": round(tmacs, 2),
                     "launch_note": "the image arrives in row bands: one launch per band over that band's rows (they run under the "
                                    "copy / layout / statistics kernels of the next band)" if launches > 1
                                    and len({u[1].shape[:2] for u in my_units}) == 1 else "one launch per size class"})
        msm = masked_stat_macs(img, my_units)
        if msm:
            # sum I^2*M passes of the masked classes: outside the score kernel's timer and `achieved`, bracketed by event pairs
            # of their own (mtm_timing.masked_stat_ms) and priced against the same int8 MFMA peak
            roof["masked_stat_macs_per_step"] = int(msm)
            ms_ms = float(np.mean(masked_ms[-args.steps:])) if masked_ms else 0.0
            if ms_ms > 0:
                ms_tops = 2.0 * msm / (ms_ms * 1e-3) / 1e12
                roof["masked_stat"] = {"ms": round(ms_ms, 4), "launches_per_step": masked_launches,
                                       "achieved": round(ms_tops, 1), "frac": round(ms_tops / I8_MFMA_PEAK_TOPS, 4)}
                roof["frac_with_masked_stat"] = round(2.0 * (macs + msm) / ((k_step + ms_ms) * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, 4)
        med = float(np.median(per_call)) * 1e3
        out = {
            "metric": "Mpixel-correlations/s", "value": round(value, 1), "unit": "Mpx-corr/s",
            "n_gpus": group_n or world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "image_hw": list(img.shape[:2]), "units": len(units),
                       "units_per_gpu": len(my_units), "method": method, "score_threshold": thr,
                       "max_overlap": 0.25, "prewarm_seconds": PREWARM_SECONDS, "prewarm_calls": n_pre + 1,
                       "parallelism": ("units sharded over %d device(s) of one process (mtm_group), %s" % (group_n, exchange_kind))
                                      if group_n else
                                      "units sharded over %d rank(s), all-gather of hits: %s" % (world, exchange_kind),
                       "processes": 1 if group_n else world,
                       "timed_region": "K x one MTM.matchTemplates call, numpy arrays in -> hit list out (SURVEY 8d): validation, "
                                       "template hand-over (unchanged templates stay resident), image H2D, window statistics, "
                                       "score kernel, peak extraction, D2H hits, all-gather (N > 1), NMS, hit list",
                       "score_maps": "not materialised (hits-only mode, MTM_OPT_HITS_ONLY=1: identical hit lists)" if hits_only
                                     else "materialised in HBM"},
            "median_ms_per_call": round(med, 4), "median_value": rate(med),
            "roofline": roof,
            "gpu_ms": {"kernels_total": round(float(np.mean(t_ms)), 4), "ncc_kernel": round(k_step, 4)},
            "clock": {"sclk_mhz_in_kernel": None if sclk is None else round(sclk, 1),
                      "before": sensors_before, "after": sensors_after},
            "hits": len(hits), "planted_found": bool(planted_ok),
        }
        if group_n:
            out["multi_gpu"] = {"mode": "one process, mtm_group", "devices": group_devices,
                                "exchange": group.exchange_used(), "exchange_requested": args.exchange or "rccl",
                                "communicator_ranks": group_comm_ranks, "exchange_note": group_exchange_note,
                                "kernel_ms_per_step_by_device": [round(float(np.mean(v)), 4) if v else None for v in dev_kernel_ms],
                                "units_by_device": [int((group.shards([(u[1], u[2] if len(u) >= 3 else None) for u in units],
                                                                      img.shape, method) == i).sum()) for i in range(group_n)]}
        elif world > 1:
            out["multi_gpu"] = {"mode": "one process per GPU", "exchange": exchange_kind, "communicator_ranks": comm_ranks}
        out.update(extras)
        ri = extras.get("resident_inputs")
        if ri and tinfo["kernel_used"] == 3 and ri.get("kernel_ms_per_launch"):
            # the same kernel as ONE full-image launch per step (inputs resident: no upload to overlap, so no bands): its
            # efficiency without the second launch tail and the copy / layout / statistics kernels running beside it
            fl = 2.0 * macs / (ri["kernel_ms_per_launch"] * 1e-3) / 1e12
            roof["full_image_launch"] = {"kernel_ms": ri["kernel_ms_per_launch"], "achieved": round(fl, 1),
                                         "frac": round(fl / I8_MFMA_PEAK_TOPS, 4),
                                         "frac_of_power_limited_ceiling": round(fl / I8_MFMA_RANDOM_OPERANDS_TOPS, 4),
                                         "sclk_mhz_in_kernel": ri.get("sclk_mhz_in_kernel")}
            # what the band split of the upload costs the score kernel itself (two launch heads and tails, the copy / layout /
            # statistics kernels running beside launch 1): part of `ms_per_step - kernel_ms_per_step` would otherwise hide here
            if roof.get("launches_per_step", 1) > 1:
                roof["band_split_overhead_ms"] = round(roof["kernel_ms_per_step"] - ri["kernel_ms_per_launch"], 4)
                roof["fixed_cost_ms"] = round(out["ms_per_step"] - roof["kernel_ms_per_step"], 4)
                roof["fixed_cost_incl_band_split_ms"] = round(out["ms_per_step"] - ri["kernel_ms_per_launch"], 4)
        if not args.no_cpu_baseline and world == 1 and not group_n:
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
