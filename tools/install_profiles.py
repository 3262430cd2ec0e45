#!/usr/bin/env python3
"""Copies the summaries written by tools/profile_round.sh (gpurun_out/r01b/)
into profiles/ and recomputes the cfg3 entry of profiles/traffic.json."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r01b")
P = os.path.join(ROOT, "profiles")


def last_json(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def counter_sum(name, kernel):
    """(sum, dispatches) of one counter for one kernel from the summary that
    tools/pmc_summary.py wrote on the GPU box (the merged gpurun_out/ directories
    may also hold files of earlier runs)."""
    total, calls, current, n = 0.0, 0, False, 0
    for line in open(os.path.join(src, "fetch_write.txt")):
        if not line.startswith(" "):
            current = kernel in line
            n = int(line.split()[-1]) if current else 0
        elif current and line.split()[0] == name:
            total += float(line.split()[1])
            calls = n
    return total, calls


shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(P, "r01_bench_default_kernel_stats.csv"))
for a, b in [("bench_profiled.json", "r01_bench_default_profiled.json"), ("bench.json", "r01_bench_default.json"),
             ("bench_strict.json", "r01_bench_strict_16spp.json")]:
    with open(os.path.join(P, b), "w") as f:
        f.write(json.dumps(last_json(os.path.join(src, a))) + "\n")
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(P, "r01_pmc_k_trace_fast.txt"))
shutil.copy(os.path.join(src, "fetch_write.txt"), os.path.join(P, "r01_pmc_fetch_write_size.txt"))
b = last_json(os.path.join(src, "bench.json"))
fetch_kb, launches = counter_sum("FETCH_SIZE", "k_trace_fast<false")
write_kb, _ = counter_sum("WRITE_SIZE", "k_trace_fast<false")
rays = b["config"]["rays_per_step"] / b["config"]["spp"] * 4   # the counter passes render 4 spp
t = json.load(open(os.path.join(P, "traffic.json")))
t["cfg3"].update({"launches": launches, "rays": rays, "FETCH_SIZE_KB_sum": fetch_kb, "WRITE_SIZE_KB_sum": write_kb,
                  "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024 / launches,
                  "hbm_bytes_per_ray": (fetch_kb + write_kb) * 1024 / rays,
                  "algorithmic_bytes_per_launch": b["roofline"]["bytes_per_ray"] * rays / launches})
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
s = last_json(os.path.join(src, "bench_strict.json"))
print("default", b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_step"], "cpu", b["cpu_baseline"]["value"])
print("strict", s["value"], "traffic/launch", t["cfg3"]["hbm_bytes_per_launch"], "per ray", t["cfg3"]["hbm_bytes_per_ray"])
