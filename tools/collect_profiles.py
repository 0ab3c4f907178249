#!/usr/bin/env python3
"""Copy what tools/profile_round.sh and tools/profile_workloads.sh left under gpurun_out/<tag>/ into profiles/ (tracked)
under the names profiles/README.md lists, and derive profiles/pmc_traffic.json (HBM bytes per launch of the dominant
kernel) from the FETCH_SIZE / WRITE_SIZE passes.   Usage: tools/collect_profiles.py r03 [extra gpurun_out dirs ...]"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {"bench.json": "%s_bench.json", "bench_configs.jsonl": "%s_bench_configs.jsonl",
         "bench_maps_materialised.json": "%s_bench_maps_materialised.json",
         "kernel_stats_banded.csv": "%s_kernel_stats_banded.csv", "kernel_stats_single.csv": "%s_kernel_stats_single_launch.csv",
         "bench_under_rocprof_banded.json": "%s_bench_under_rocprof_banded.json",
         "bench_under_rocprof_single.json": "%s_bench_under_rocprof_single_launch.json",
         "bench_driver.json": "%s_bench_driver_flags.json", "bench_group2.json": "%s_bench_group2_one_gpu.json",
         "bench_group8.json": "%s_bench_group8_one_gpu.json", "bench_group1_rccl.json": "%s_bench_group1_rccl.json"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b % tag))
os.makedirs(os.path.join(dst, "pmc_" + tag), exist_ok=True)
for p in ("p1", "p2", "p3", "p4", "p5", "p3_maps", "p4_maps"):
    f = os.path.join(src, "pmc_%s.csv" % p)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "pmc_" + tag, p + ".csv"))
# secondary workloads: wl_<name>.json, wl_<name>_kernel_stats.csv, wl_<name>_pmc_{sq,fetch,write}.csv
wl_dir = os.path.join(dst, "workloads_" + tag)
wl = sorted(glob.glob(os.path.join(src, "wl_*")))
if wl:
    os.makedirs(wl_dir, exist_ok=True)
    for f in wl:
        shutil.copy(f, os.path.join(wl_dir, os.path.basename(f)[3:]))
# further scratch directories given on the command line: every small text / json / csv file, under profiles/<tag>_<dir>/
for extra in sys.argv[2:]:
    d = os.path.join(ROOT, "gpurun_out", extra)
    out = os.path.join(dst, "%s_%s" % (tag, extra.replace("/", "_")))
    os.makedirs(out, exist_ok=True)
    for f in glob.glob(os.path.join(d, "*")):
        if os.path.isfile(f) and os.path.getsize(f) < 200000 and f.rsplit(".", 1)[-1] in ("txt", "json", "jsonl", "csv", "log"):
            shutil.copy(f, out)


def counter(path, name, kernel):
    """average of counter `name` per dispatch of the kernel with the most dispatches whose name contains `kernel`"""
    best = None
    for row in csv.DictReader(open(path)):
        if kernel in row["kernel"] and row["counter"] == name:
            if best is None or int(row["dispatches"]) > best[1]:
                best = (float(row["avg_per_dispatch"]), int(row["dispatches"]))
    return None if best is None else best[0]


table = json.load(open(os.path.join(dst, "pmc_traffic.json")))
d = os.path.join(dst, "pmc_" + tag)
if os.path.exists(os.path.join(d, "p3.csv")):
    f, w = counter(os.path.join(d, "p3.csv"), "FETCH_SIZE", "ncc_mfma_kernel"), counter(os.path.join(d, "p4.csv"), "WRITE_SIZE", "ncc_mfma_kernel")
    table["ncc_mfma_kernel/north_star/n1/hits_only"] = int(round((2 * f + w) * 1024))
    f, w = counter(os.path.join(d, "p3_maps.csv"), "FETCH_SIZE", "ncc_mfma_kernel"), counter(os.path.join(d, "p4_maps.csv"), "WRITE_SIZE", "ncc_mfma_kernel")
    table["ncc_mfma_kernel/north_star/n1"] = int(round((2 * f + w) * 1024))
# per-workload traffic of the dominant score kernel (the kernel with the largest share in the workload's kernel trace)
for js in sorted(glob.glob(os.path.join(wl_dir, "*.json"))) if wl else []:
    name = os.path.basename(js)[:-5]
    ks = os.path.join(wl_dir, name + "_kernel_stats.csv")
    fe, wr = os.path.join(wl_dir, name + "_pmc_fetch.csv"), os.path.join(wl_dir, name + "_pmc_write.csv")
    if not (os.path.exists(ks) and os.path.exists(fe) and os.path.exists(wr)):
        continue
    rows = list(csv.DictReader(open(ks)))
    if not rows:
        continue
    dom = rows[0]["kernel"].split("(")[0].replace("void ", "").strip()
    f, w = counter(fe, "FETCH_SIZE", dom.split("<")[0]), counter(wr, "WRITE_SIZE", dom.split("<")[0])
    if f is not None and w is not None:
        table["workload/%s" % name] = {"kernel": dom, "hbm_bytes_per_launch": int(round((2 * f + w) * 1024)),
                                       "launch_us": float(rows[0]["avg_us"]), "share_of_gpu_time_percent": float(rows[0]["percent"])}
table["_note"] = ("HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from the rocprofv3 --pmc passes under profiles/pmc_%s/ "
                  "(headline: full-image launches, MTM_UPLOAD_BANDS=1; hits-only and with the maps materialised) and "
                  "profiles/workloads_%s/ (the dominant kernel of every secondary workload).  FETCH_SIZE is doubled per "
                  "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE uncorrected.  Algorithmic bytes of "
                  "the headline launch: 135151376 (hits-only) / 1022232704 (maps).") % (tag, tag)
json.dump(table, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in table.items() if not k.startswith("_")}, indent=1)[:3000])
