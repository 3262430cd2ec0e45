#!/usr/bin/env python3
"""Sweep of the traversal kernels' wave-scheduling parameters on the bench frame:
one process, one scene, an 8-spp frame per setting (whole pipeline and
traversal-only Mrays/s).  GPU only; for development.

    python tools/tune_frame.py [fast|strict] "stack,refill,blocks,descend_min,leaf_min;..."
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
    sweeps = sys.argv[2] if len(sys.argv) > 2 else "8,60,8,32,16"
    sys.argv = [sys.argv[0], "--cpu-seconds", "0", "--no-extra"]
    args = bench.parse_args()
    job = bench.Job(args)
    W = bench.Workload(job, os.environ.get("WORKLOAD", "cfg3"))
    spp = int(os.environ.get("SPP", "8"))
    for rep in range(int(os.environ.get("REPS", "2"))):
        for setting in sweeps.split(";"):
            W.scene.set_tuning(*[int(x) for x in setting.split(",")])
            leg = bench.run_leg(job, W, mode, spp, 3, 1)
            print(f"{mode} {setting:18s} rep {rep}: {leg['value']:8.1f} Mrays/s, traversal {leg['rays_per_s_in_kernel'] / 1e6:8.1f}",
                  flush=True)


if __name__ == "__main__":
    main()
