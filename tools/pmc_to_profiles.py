#!/usr/bin/env python3
"""Turns rocprofv3 --pmc counter passes over bench.py into the per-ray figures
bench.py reads from profiles/:

    pmc_to_profiles.py valu    DIR BENCH_LOG WORKLOAD MODE   -> profiles/<round>_valu_per_ray.json
    pmc_to_profiles.py traffic FETCH_DIR WRITE_DIR BENCH_LOG WORKLOAD MODE -> profiles/<round>_traffic.json

<round> is $RT_PROFILE_ROUND (default r03).  Every entry is stamped with what it
was measured on: the library's rt_version(), the hash of the kernel sources
(bench.py: kernel_source_sha, the same hash bench.py compares against when it
reads the entry) and the git commit given in $RT_GIT_HEAD (the GPU box has no
.git; the caller passes `git rev-parse --short HEAD`).

DIR holds the counter CSVs of ONE run of `bench.py --steps 1 --warmup 0 ...`
(SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE in
one pass); BENCH_LOG is that run's stdout, whose JSON line gives the rays of the
step.  Only the dispatches of the timed kernel (k_trace_fast<false, ...> or
k_trace_ref<..., false, ...>) are summed: the counting variants that bench.py
runs on a sample tile have other template arguments.

Normalisation, written into the file as well:
  lane_inst_per_ray = SQ_THREAD_CYCLES_VALU / rays   (active lanes summed over VALU instructions)
  wave_inst_per_ray = SQ_INSTS_VALU / rays
  avg_active_lanes  = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU          (of 64)
  valu_busy         = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE * 128)
      GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM/8 is the kernel's cycles;
      the chip has 1024 SIMDs, and a VALU instruction occupies its SIMD for 4 cycles
  hbm_bytes_per_ray = (FETCH_SIZE + WRITE_SIZE) * 1024 / rays         (counters are in KB; no x2:
      for scattered 64-byte records FETCH_SIZE reads the requested bytes as they are, calibrated
      with the gather probe, see the calibration note in profiles/traffic.json)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("RT_PROFILE_ROUND", "r03")


def provenance():
    """What the counters were measured on (see the module docstring)."""
    sys.path.insert(0, ROOT)
    import bench
    from raytracer_amd import api
    return {"git_head": os.environ.get("RT_GIT_HEAD") or None, "rt_version": api.version(),
            "kernel_source_sha": bench.kernel_source_sha()}


def timed_kernel(name, mode):
    if mode == "fast":
        return "k_trace_fast<false" in name
    real = "double" if mode == "fp64" else "float"
    return f"k_trace_ref<{real}, false" in name


def sums(d, mode):
    tot, calls = defaultdict(float), set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if timed_kernel(r["Kernel_Name"], mode):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                calls.add(r["Dispatch_Id"])
    return tot, len(calls)


def bench_line(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {path}")


def merge(fname, workload, mode, entry):
    path = os.path.join(ROOT, "profiles", fname)
    data = json.load(open(path)) if os.path.exists(path) else {}
    entry.update(provenance())
    data.setdefault(workload, {})[mode] = entry
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


def main():
    what = sys.argv[1]
    if what == "valu":
        d, log, workload, mode = sys.argv[2:6]
        c, n = sums(d, mode)
        doc = bench_line(log)
        rays = doc["config"]["rays_per_step"] * doc["steps"]
        merge(f"{ROUND}_valu_per_ray.json", workload, mode, {
            "command": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE ... -- "
                       f"python3 bench.py --workload {workload} --mode {mode} --steps 1 --warmup 0 --spp {doc['config']['spp']} --cpu-seconds 0",
            "kernel": doc["roofline"]["kernel"], "dispatches": n, "rays": rays,
            "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"],
            "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"],
            "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"),
            "lane_inst_per_ray": round(c["SQ_THREAD_CYCLES_VALU"] / rays, 1),
            "wave_inst_per_ray": round(c["SQ_INSTS_VALU"] / rays, 2),
            "avg_active_lanes": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 2),
            "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] * 128), 4),
            "normalisation": "lane_inst_per_ray = SQ_THREAD_CYCLES_VALU / rays; valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / "
                             "(GRBM_GUI_ACTIVE summed over 8 XCDs x 128 SIMDs per XCD); peak issue = 256 CUs x 4 SIMDs x 16 "
                             "lanes per clock x 2.4 GHz = 39.3 T lane-instructions/s; valu_busy is an indicator only "
                             "(roofline.frac does not use it): a value above 1, as the reference-order kernel shows, "
                             "means that not every counted instruction occupies its SIMD for four cycles"})
    elif what == "traffic":
        fd, wd, log, workload, mode = sys.argv[2:7]
        f, n = sums(fd, mode)
        w, _ = sums(wd, mode)
        doc = bench_line(log)
        rays = doc["config"]["rays_per_step"] * doc["steps"]
        fetch = f.get("FETCH_SIZE", f.get("FETCH_SIZE_sum", 0.0))
        write = w.get("WRITE_SIZE", w.get("WRITE_SIZE_sum", 0.0))
        merge(f"{ROUND}_traffic.json", workload, mode, {
            "command": f"rocprofv3 --pmc FETCH_SIZE (and, in a run of its own, WRITE_SIZE) -- python3 bench.py --workload "
                       f"{workload} --mode {mode} --steps 1 --warmup 0 --spp {doc['config']['spp']} --cpu-seconds 0",
            "kernel": doc["roofline"]["kernel"], "dispatches": n, "rays": rays,
            "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
            "hbm_bytes_per_ray": round((fetch + write) * 1024 / rays, 2),
            "note": "fabric-side bytes of the L2 (Infinity Cache hits included), per ray of the timed traversal kernel"})
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
