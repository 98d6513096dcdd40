"""bench.py's labels follow its arguments (no GPU needed): only the default sizes may be called BASELINE configs[1], and
roofline.traffic is taken from a PMC file only when that file was measured on the current kernel sources and workload."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workload_label_follows_the_arguments():
    import bench

    assert bench.spmv_workload_name(10_000_000, 160_000_000, 10_000_000, 162_698_392) == \
        "csr_spmv_f64_rmat_10Mx10M_160Mnnz (BASELINE configs[1])"
    other = bench.spmv_workload_name(4_000_000, 64_000_000, 4_000_000, 80_318_267)
    assert "BASELINE configs[1]" not in other and "NOT a BASELINE config" in other and "4Mx4M" in other


def test_traffic_is_only_read_for_matching_sources_and_workload(tmp_path, monkeypatch):
    import bench

    good = bench.kernel_source_hash("spam_spmv.hip", "spam_spmv_rg.hip")
    prof = os.path.join(ROOT, "profiles", "_test_pmc.json")
    try:
        json.dump({"kernel_source_sha16": good, "workload": {"rows": 7, "plan": "p"}, "traffic_bytes_per_launch": 123.4}, open(prof, "w"))
        tr, src = bench.verified_traffic("_test_pmc.json", ("spam_spmv.hip", "spam_spmv_rg.hip"), {"rows": 7, "plan": "p"})
        assert tr == 123 and "_test_pmc.json" in src
        tr, src = bench.verified_traffic("_test_pmc.json", ("spam_spmv.hip", "spam_spmv_rg.hip"), {"rows": 8, "plan": "p"})
        assert tr is None and "another workload" in src
        json.dump({"kernel_source_sha16": "0" * 16, "workload": {"rows": 7}, "traffic_bytes_per_launch": 1.0}, open(prof, "w"))
        tr, src = bench.verified_traffic("_test_pmc.json", ("spam_spmv.hip", "spam_spmv_rg.hip"), {"rows": 7})
        assert tr is None and "other kernel sources" in src
    finally:
        if os.path.exists(prof):
            os.remove(prof)


def test_committed_pmc_files_match_the_committed_kernel_sources():
    """profiles/*_pmc_latest.json are stamped with the hash of the kernel sources they were measured on: a commit that
    touches a kernel without re-taking them would make bench.py print traffic = null."""
    import bench

    d = json.load(open(os.path.join(ROOT, "profiles", "spmv_pmc_latest.json")))
    assert d["kernel_source_sha16"] == bench.kernel_source_hash("spam_spmv.hip", "spam_spmv_rg.hip")
    d = json.load(open(os.path.join(ROOT, "profiles", "spgemm_pmc_latest.json")))
    assert d["kernel_source_sha16"] == bench.kernel_source_hash("spam_spgemm.hip")
