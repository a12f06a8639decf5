"""CPU suite: the bench line's PMC constants are bound to the build they were measured on.

profiles/traffic.json records, per workload, the SHA-256 of the engine library its rocprofv3 --pmc passes ran on (tools/pmc_classes.py --lib);
bench.load_traffic hands the counters out only when the library this process loads has that hash, and says `traffic_stale` otherwise."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_record_carries_the_library_hash_for_every_workload():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert set(t) >= {"dac44k", "encodec48k", "snac44k"}
    hashes = {k: v.get("_lib_sha256") for k, v in t.items()}
    assert all(isinstance(h, str) and len(h) == 64 for h in hashes.values()), hashes
    assert len(set(hashes.values())) == 1, "the three workloads were profiled on different builds"
    k7 = t["dac44k"]["conv_k7"]
    assert {"hbm_bytes_per_launch", "mfma_busy", "valu_busy", "pipe_busy"} <= set(k7)
    assert abs(k7["pipe_busy"] - (k7["mfma_busy"] + k7["valu_busy"])) < 1e-3 and 0.5 < k7["pipe_busy"] <= 1.02


def test_load_traffic_withholds_counters_of_another_build(monkeypatch):
    import bench
    from neuralcodecs_amd import _lib
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["dac44k"]["_lib_sha256"]
    monkeypatch.setattr(_lib, "lib_sha256", lambda: rec)
    t = bench.load_traffic("dac44k")
    assert t is not None and "conv_k7" in t and not bench.traffic_stale("dac44k")
    monkeypatch.setattr(_lib, "lib_sha256", lambda: "0" * 64)
    assert bench.load_traffic("dac44k") is None and bench.traffic_stale("dac44k")
    assert bench.load_traffic("no-such-workload") is None


def test_compact_class_table_shape():
    import bench
    classes = {"conv_k7": {"ms_per_step": 40.1, "launches_per_step": 24.0, "tflops": 125.6, "algo_GBps": 0.0},
               "stem": {"ms_per_step": 0.064, "launches_per_step": 1.0, "tflops": 0.0, "algo_GBps": 5762.2}}
    c = bench.compact_classes(classes)
    assert c["conv_k7"] == [40.1, 24.0, 125.6, round(125.6 / bench.FP32_MFMA_PEAK_TFLOPS, 3)]
    assert c["stem"][2] == 5762.2 and c["stem"][3] == round(5762.2 / bench.HBM_PEAK_GBS, 3)
