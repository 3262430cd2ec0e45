#!/usr/bin/env python3
"""profiles/r02_pmc_k_trace_fast.txt: the ordered traversal kernel's section of
profiles/r02_pmc_cfg3_all_kernels.txt with the per-ray figures derived from
profiles/r02_valu_per_ray.json and profiles/r02_traffic.json beside it.

    python tools/pmc_kernel_summary.py        (after tools/profile_round2.sh cfg3)
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def main():
    src = open(os.path.join(P, "r02_pmc_cfg3_all_kernels.txt")).read().split("\n")
    i = [k for k, l in enumerate(src) if l.startswith("void rt::k_trace_fast<false, 8>")][0]
    sec = [src[i]]
    for l in src[i + 1:]:
        if not l.startswith("   "):
            break
        sec.append(l)
    c = {l.split()[0]: float(l.split()[1]) for l in sec[1:]}
    v = json.load(open(os.path.join(P, "r02_valu_per_ray.json")))["cfg3"]["fast"]
    t = json.load(open(os.path.join(P, "r02_traffic.json")))["cfg3"]["fast"]
    out = ["k_trace_fast<false, 8>, bench frame (1M-triangle room, 3840x2160) at 4 spp: 6 dispatches, %d rays" % v["rays"],
           "counters: tools/profile_round2.sh cfg3 (one rocprofv3 --pmc pass per group), summed over the dispatches",
           ""] + sec + [
        "",
        "derived (single pass each, so numerator and denominator come from the same run):",
        "   VALU wave-level instructions per ray      %.2f   (round 1: 127; start of round 2: 122.2)" % v["wave_inst_per_ray"],
        "   VALU lane-instructions per ray            %.1f" % v["lane_inst_per_ray"],
        "   lanes per VALU instruction                %.2f of 64   (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU; round 1: 31)" % v["avg_active_lanes"],
        "   VALU busy                                 %.3f   (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD))" % v["valu_busy"],
        "   SALU instructions per VALU instruction    %.2f   (round 1: 0.58)" % (v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]),
        "   fabric bytes per ray                      %.1f   (FETCH_SIZE + WRITE_SIZE)" % t["hbm_bytes_per_ray"],
        "   L2 hit rate                               %.2f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))]
    open(os.path.join(P, "r02_pmc_k_trace_fast.txt"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
