#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked) under the names
profiles/README.md lists, and derive profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.
Usage: tools/collect_profiles.py r02"""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {"bench.json": "%s_bench.json", "bench_configs.jsonl": "%s_bench_configs.jsonl",
         "bench_maps_materialised.json": "%s_bench_maps_materialised.json",
         "kernel_stats_banded.csv": "%s_kernel_stats_banded.csv", "kernel_stats_single.csv": "%s_kernel_stats_single_launch.csv",
         "bench_under_rocprof_banded.json": "%s_bench_under_rocprof_banded.json",
         "bench_under_rocprof_single.json": "%s_bench_under_rocprof_single_launch.json"}
for a, b in names.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b % tag))
os.makedirs(os.path.join(dst, "pmc_" + tag), exist_ok=True)
for p in ("p1", "p2", "p3", "p4", "p5", "p3_maps", "p4_maps"):
    shutil.copy(os.path.join(src, "pmc_%s.csv" % p), os.path.join(dst, "pmc_" + tag, p + ".csv"))


def counter(path, name):
    for row in csv.DictReader(open(path)):
        if "ncc_mfma_kernel" in row["kernel"] and row["counter"] == name:
            return float(row["avg_per_dispatch"])
    raise SystemExit("no %s in %s" % (name, path))


d = os.path.join(dst, "pmc_" + tag)
hits_only = (2 * counter(os.path.join(d, "p3.csv"), "FETCH_SIZE") + counter(os.path.join(d, "p4.csv"), "WRITE_SIZE")) * 1024
maps = (2 * counter(os.path.join(d, "p3_maps.csv"), "FETCH_SIZE") + counter(os.path.join(d, "p4_maps.csv"), "WRITE_SIZE")) * 1024
old = json.load(open(os.path.join(dst, "pmc_traffic.json")))
old["ncc_mfma_kernel/north_star/n1/hits_only"] = int(round(hits_only))
old["ncc_mfma_kernel/north_star/n1"] = int(round(maps))
json.dump(old, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("hits-only %.2f MB, maps %.2f MB per full-image launch" % (hits_only / 1e6, maps / 1e6))
