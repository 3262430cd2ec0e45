"""The counter passes under profiles/ that bench.py's roofline constants come from
must belong to the kernels in the tree: every entry carries a hash of the kernel
sources it was measured with, and a kernel change without a new counter pass would
silently keep the older kernel's per-ray figures in the bench line (the line itself
says so in roofline.profile_warning; this test says it before anything is run on a
GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_counter_passes_were_made_with_the_kernels_in_the_tree():
    import bench
    now = bench.kernel_source_sha()
    for kind in ("valu_per_ray", "traffic"):
        for workload, mode in (("cfg3", "fast"), ("cfg3", "strict"), ("cfg4", "fast")):
            entry, name = bench.profile_entry(kind, workload, mode)
            assert entry is not None, (kind, workload, mode)
            assert name.startswith(bench.PROFILE_ROUNDS[0]), name        # the newest round's file
            assert entry["kernel_source_sha"] == now, (
                f"profiles/{name} [{workload}][{mode}] was measured at commit {entry.get('git_head')} with other "
                f"kernel sources: run tools/profile_round3.sh {workload} and tools/install_round3.py {workload}")


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r03_bench_cfg3.json")) as f:
        line = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra"):
        assert key in line, key
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["profile_warning"] is None
    for leg in ("strict_cfg3", "fast_cfg4"):
        r = line["extra"][leg]["roofline"]
        assert r["profile_warning"] is None and r["frac"] is not None, leg
    base = line["cpu_baseline"]
    assert base["kind"] == "port" and base["cores"] >= 1 and base["value"] > 0
