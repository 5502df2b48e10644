"""bench.py's launcher and line shaping, without a GPU: `--gpus N` outside torchrun starts N ranks itself and the line
carries n_gpus = N; a launcher that started another number of ranks is refused; the nested workload lines are cut down
and summarised so that the END of the one JSON line shows every workload."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 3 and line["scaling"] == "strong"


def test_world_size_other_than_gpus_is_refused():
    e = _env(); e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode != 0 and "n_gpus" in r.stderr


def test_nested_lines_are_compact_and_summarised():
    sys.path.insert(0, ROOT)
    import bench
    full = {"metric": "m", "value": 12.5, "unit": "MB/s", "ms_per_step": 80.0, "verified_all_blocks": True, "verified_dedup": True,
            "identity": "x" * 500, "roofline_all": [{"kernel": "k"}] * 40,
            "kernels_ms_per_step": {"k%d" % i: float(100 - i) for i in range(30)},
            "roofline": {"bound": "hbm", "kernel": "sha1_chain_kernel", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.000125,
                         "traffic": None, "avg_launch_ms": 215.0, "note": "y" * 300},
            "cpu_baseline": {"value": 6.1, "unit": "MB/s", "cores": 16, "kind": "reference", "sample": "whole job", "extra": list(range(100))},
            "config": {"workload": "w", "input_bytes": 1, "fragments": 2}}
    c = bench.compact_line(full)
    assert len(json.dumps(c)) < 1200
    assert c["value"] == 12.5 and c["roofline"]["kernel"] == "sha1_chain_kernel" and "note" not in c["roofline"]
    assert len(c["kernels_ms_per_step"]) == 6 and c["cpu_baseline"]["kind"] == "reference"
    assert bench.summary_row(full) == [12.5, 80.0, True, 0.000125, 6.1]
    assert bench.summary_row({"error": "boom", "rc": 1}) == [None, None, False, None, None]
    bad = dict(full, verified_dedup=False)
    assert bench.summary_row(bad)[2] is False
    # round 5: what a reader of the one line needs of a nested workload survives the cut -- the lone-job figure, the three rooflines
    # with their PMC traffic, the product call of config 4
    r = {"bound": "hbm", "kernel": "k", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": 123, "traffic_over_algorithmic": 1.2,
         "avg_launch_ms": 1.0, "launches_per_step": 1.0, "ms_per_step": 1.0, "measured": "one job alone", "note": "z" * 200, "integer_issue_ceiling_GBps": 1.0}
    more = dict(full, roofline=r, roofline_in_flight=r, roofline_longest_chain=dict(r, waves=13), single_job={"ms": 391.0, "value": 172.0, "note": "n" * 300},
                product_one_call={"ms": 4710.6, "archive_bytes": 5728947211, "note": "n" * 200}, archive_blocks={"c": 1, "d": 13, "h": 13, "i": 175})
    c = bench.compact_line(more)
    assert c["single_job"] == {"ms": 391.0, "value": 172.0} and c["product_one_call"]["ms"] == 4710.6 and c["archive_blocks"]["i"] == 175
    for k in ("roofline", "roofline_in_flight", "roofline_longest_chain"):
        assert c[k]["traffic"] == 123 and c[k]["measured"] == "one job alone" and "note" not in c[k]
    assert c["roofline_longest_chain"]["waves"] == 13 and len(json.dumps(c)) < 2600


def test_steady_window_counts_exactly_k_completions_with_the_pipeline_full():
    """bench.steady_window: depth + warm + steps + depth jobs between the barriers, the clock from the completion of job
    depth+warm to the completion of job depth+warm+steps -- independent of how long the pipeline takes to fill or drain."""
    import bench
    calls = []

    def fake_run(n):                       # a pipeline that fills slowly (first `depth` completions late), then one job per 10 ms
        calls.append(n)
        t, done = 100.0, []
        for i in range(n):
            t += 0.5 if i < 4 else 0.010
            done.append(t)
        done[-1] += 3.0                    # ... and drains slowly
        return 7, done[::-1]               # (completion order is not job order)
    bars = []
    dt, n, out, ct = bench.steady_window(fake_run, lambda: bars.append(1), 4, 5, 20)
    assert calls == [4 + 5 + 20 + 4] and n == 33 and out == 7 and len(bars) == 2
    assert abs(dt - 20 * 0.010) < 1e-9
    # one step in flight: the contract's literal bracket
    calls.clear(); bars.clear()
    dt, n, out, ct = bench.steady_window(lambda k: (calls.append(k), (3, [0.0] * k))[1], lambda: bars.append(1), 1, 2, 6)
    assert calls == [2, 6] and n == 6 and out == 3 and ct is None and len(bars) == 2
