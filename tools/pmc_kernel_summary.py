#!/usr/bin/env python3
"""profiles/<round>_pmc_<kernel>.txt: one traversal kernel's section of
profiles/<round>_pmc_cfg3_all_kernels.txt with the per-ray figures derived from
profiles/<round>_valu_per_ray.json and profiles/<round>_traffic.json beside it.

    python tools/pmc_kernel_summary.py [round] [fast|strict]     (after tools/profile_round3.sh cfg3)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
    prefix = "void rt::k_trace_fast<false, 8" if mode == "fast" else "void rt::k_trace_ref<float, false"
    src = open(os.path.join(P, f"{rnd}_pmc_cfg3_all_kernels.txt")).read().split("\n")
    i = [k for k, l in enumerate(src) if l.startswith(prefix)][0]
    sec = [src[i]]
    for l in src[i + 1:]:
        if not l.startswith("   "):
            break
        sec.append(l)
    c = {l.split()[0]: float(l.split()[1]) for l in sec[1:]}
    v = json.load(open(os.path.join(P, f"{rnd}_valu_per_ray.json")))["cfg3"][mode]
    t = json.load(open(os.path.join(P, f"{rnd}_traffic.json")))["cfg3"][mode]
    rays = v["rays"]
    out = [f"{sec[0].split(' dispatches')[0][10:]}, bench frame (1M-triangle room, 3840x2160) at 4 spp: 6 dispatches, {int(rays)} rays",
           f"counters: tools/profile_round3.sh cfg3 (one rocprofv3 --pmc pass per group), summed over the dispatches; "
           f"commit {v.get('git_head')}, {v.get('rt_version')}, kernel sources {v.get('kernel_source_sha')}",
           ""] + sec + [
        "",
        "derived (single pass each, so numerator and denominator come from the same run):",
        "   VALU wave-level instructions per ray      %.2f" % v["wave_inst_per_ray"],
        "   VALU lane-instructions per ray            %.1f" % v["lane_inst_per_ray"],
        "   lanes per VALU instruction                %.2f of 64   (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU)" % v["avg_active_lanes"],
        "   VALU busy                                 %.3f   (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD))" % v["valu_busy"],
        "   SALU instructions per VALU instruction    %.2f" % (v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]),
        "   vector memory read instructions per ray   %.2f wave-level (SQ_INSTS_VMEM_RD / rays)" % (c["SQ_INSTS_VMEM_RD"] / rays),
        "   TA busy                                   %.2f   (TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 x 256 CUs))" % (c["TA_TA_BUSY_sum"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256)),
        "   L1 lookups per ray                        %.1f   (TCP_TOTAL_CACHE_ACCESSES_sum / rays: one per lane and 16-byte load)" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays),
        "   L1 lookups per clock and CU               %.2f   (the same / (GRBM_GUI_ACTIVE / 8 x 256 CUs))" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256)),
        "   LDS cycles lost to bank conflicts         %s" % ("%.2f of the LDS's busy cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else "not collected"),
        "   fabric bytes per ray                      %.1f   (FETCH_SIZE + WRITE_SIZE)" % t["hbm_bytes_per_ray"],
        "   L2 hit rate                               %.2f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))]
    name = "k_trace_fast" if mode == "fast" else "k_trace_ref"
    open(os.path.join(P, f"{rnd}_pmc_{name}.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-10:]))


if __name__ == "__main__":
    main()
