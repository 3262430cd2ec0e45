#!/usr/bin/env python3
"""Copies what tools/profile_round3.sh left under gpurun_out/r03_<cfg>/ into
profiles/ (run here, after the gpurun call has merged its output back):

    python tools/install_round3.py cfg3|cfg4

cfg3: the counter files (r03_valu_per_ray.json, r03_traffic.json, which the
counter passes updated on the GPU box), the summary of every kernel's counters,
the two bench lines and the kernel statistics of the profiled default run.
cfg4: the counter files and the summary.  The per-kernel digests
(r03_pmc_k_trace_*.txt) are rewritten from the cfg3 summary.
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def last_json(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def main():
    wl = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "r03_" + wl)
    # only this workload's entries: the other workload's may be newer here than in that run's copy
    for name in ("r03_valu_per_ray.json", "r03_traffic.json"):
        new = json.load(open(os.path.join(src, name)))
        dst = os.path.join(P, name)
        cur = json.load(open(dst)) if os.path.exists(dst) else {}
        cur[wl] = new[wl]
        with open(dst, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
            f.write("\n")
    shutil.copy(os.path.join(src, "pmc_all_kernels.txt"), os.path.join(P, f"r03_pmc_{wl}_all_kernels.txt"))
    if wl == "cfg3":
        shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(P, "r03_bench_default_kernel_stats.csv"))
        for a, b in (("bench.json", "r03_bench_cfg3.json"), ("bench_profiled.json", "r03_bench_profiled_cfg3.json")):
            with open(os.path.join(P, b), "w") as f:
                json.dump(last_json(os.path.join(src, a)), f, indent=1)
                f.write("\n")
        for mode in ("fast", "strict"):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_kernel_summary.py"), "r03", mode])
    b = json.load(open(os.path.join(P, "r03_valu_per_ray.json")))[wl]
    t = json.load(open(os.path.join(P, "r03_traffic.json")))[wl]
    for mode in b:
        print(wl, mode, "wave-inst/ray", b[mode]["wave_inst_per_ray"], "lanes", b[mode]["avg_active_lanes"],
              "valu busy", b[mode]["valu_busy"], "fabric B/ray", t[mode]["hbm_bytes_per_ray"], "head", b[mode]["git_head"])


if __name__ == "__main__":
    main()
