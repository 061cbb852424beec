"""CPU suite: the schema and the invariants of bench.py's JSON line, built from made-up measurements (no GPU): the contract
fields of the driver, the roofline / valu_int objects (every fraction at most 1: VERDICT r02 item 2), the N > 1 diagnostics
 and the traffic stamp check (item 6)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PLAN = {"R": 1, "qblocks": 256, "chunk": 2048, "chunks": 34, "lead_rows": 8192, "lead_chunks": 1, "tail_chunks": 8, "cus": 256, "sgpr_feed": 1}


def test_n1_line_has_the_contract_fields_and_no_fraction_above_one():
    b = _bench()
    rec = b.bench_line(world=1, steps=50, warmup=5, loop_closure=False, n_query=65536, n_train=65536, n_local=65536, wall_ms=53.5,
                       dev_ms=53.4, rank_kernel_ms=[1.0647], launches=50, collective="none", fallback_reason=None, rccl_version=22707,
                       plan=PLAN, ok=True, train_replication="per-rank upload")
    json.dumps(rec)                                            # serialisable as is
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "parity_spot_check"):
        assert key in rec, key
    assert rec["metric"].startswith("descriptor pairs/sec BF-Hamming knn=2") and rec["unit"] == "pairs/s" and rec["n_gpus"] == 1
    assert rec["vs_baseline"] is None and rec["dtype"] == "u32" and rec["higher_is_better"] is True and "workload" in rec["config"]
    assert abs(rec["value"] - 65536 * 65536 / (53.5 / 50 * 1e-3)) < 1e3 and abs(rec["ms_per_step"] - 1.07) < 1e-9
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["kernel"] == "bf_top2_kernel"
    assert roof["algorithmic_bytes_per_launch"] == 5242880 and abs(roof["achieved"] - 5242880 / 1.0647e-3 / 1e9) < 1e-9
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-15 and 0 < roof["frac"] < 1
    v = roof["valu_int"]
    assert 0 < v["frac_of_32lane_peak"] <= 1 and 0 < v["frac_of_issue_floor"] <= 1
    assert abs(v["frac_of_32lane_peak"] - v["frac_of_issue_floor"]) < 1e-12      # 16 x 4 / 2 cycles per row IS the 32-lane peak
    assert abs(v["issue_floor_ms"] - 65536 * 32 / 2.4e9 * 1e3) < 1e-9
    assert not any("sustained" in k or "isolated" in k for k in v)             # the wall-clock x 2.4 GHz "floors" are gone
    # an impossibly fast kernel would show as a fraction above 1 - the line does not clamp, it states the model: check the model's edge
    fast = b.bench_line(world=1, steps=1, warmup=0, loop_closure=False, n_query=65536, n_train=65536, n_local=65536, wall_ms=0.9,
                        dev_ms=0.9, rank_kernel_ms=[65536 * 32 / 2.4e9 * 1e3], launches=1, collective="none", fallback_reason=None,
                        rccl_version=None, plan=PLAN, ok=True, train_replication="per-rank upload")
    assert abs(fast["roofline"]["valu_int"]["frac_of_issue_floor"] - 1.0) < 1e-12


def test_n_gt_1_and_loop_closure_lines_explain_themselves():
    b = _bench()
    rec = b.bench_line(world=8, steps=20, warmup=2, loop_closure=True, n_query=1 << 20, n_train=1 << 20, n_local=1 << 17, wall_ms=800.0,
                       dev_ms=790.0, rank_kernel_ms=[33.1, 33.0, 34.2, 33.3, 33.1, 33.2, 33.0, 33.4], launches=20, collective="xgmi-p2p-copies",
                       fallback_reason="rank 3: rccl: SlamHipError: ncclCommInitRank failed", rccl_version=22707, plan=dict(PLAN, qblocks=512),
                       ok=True, train_replication="per-rank upload")
    cfg, roof = rec["config"], rec["roofline"]
    assert "loop-closure" in rec["metric"] and "BASELINE configs[3]" in cfg["workload"] and "64 per rank" in cfg["sharding"]
    assert cfg["collective"] == "xgmi-p2p-copies" and "rank 3" in cfg["collective_fallback_reason"] and cfg["rccl_version"] == 22707
    assert roof["kernel_ms"] == 34.2 and roof["kernel_ms_per_rank"] == {"min": 33.0, "max": 34.2}
    assert roof["algorithmic_bytes_per_launch"] == 32 * ((1 << 17) + (1 << 20)) + 16 * (1 << 17)      # the SHARD's bytes
    assert roof["traffic"] is None and "N=1" in roof["traffic_source"]
    assert 0 < roof["valu_int"]["frac_of_32lane_peak"] <= 1
    assert rec["value"] == (1 << 40) / (800.0 / 20 * 1e-3) and rec["n_gpus"] == 8 and rec["scaling"] == "strong"


def test_traffic_is_withheld_when_a_kernel_source_changed(tmp_path, monkeypatch):
    b = _bench()
    good = {"source_sha": {n: b.source_sha(n) for n in ("bf_hamming.hip", "bf_scan_sgpr.h", "reproj.hip")},
            "bf_top2_kernel": {"traffic_bytes": 123.0}, "reproj_rj_kernel": {"traffic_bytes": 456.0}}
    f = tmp_path / "hbm_counters.json"
    f.write_text(json.dumps(good))
    monkeypatch.setattr(b, "HBM_COUNTERS", str(f))
    assert b.profiled_traffic("bf_top2_kernel")[0] == 123.0 and b.profiled_traffic("reproj_rj_kernel")[0] == 456.0
    stale = dict(good, source_sha=dict(good["source_sha"], **{"bf_scan_sgpr.h": "0" * 16}))
    f.write_text(json.dumps(stale))
    val, why = b.profiled_traffic("bf_top2_kernel")
    assert val is None and "bf_scan_sgpr.h" in why
    assert b.profiled_traffic("reproj_rj_kernel")[0] == 456.0                  # its own source is unchanged
    f.write_text("{}")
    assert b.profiled_traffic("bf_top2_kernel")[0] is None
