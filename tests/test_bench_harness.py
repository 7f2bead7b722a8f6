"""CPU tier: bench.py's timing/aggregation harness, including the N>1 path over gloo (world_size 2).
The fake-quant path does not shard (replicas only), so the only distributed logic is
barrier + max-over-ranks + whole-job aggregation -- that is what runs here."""
import json
import os
import socket
import subprocess
import sys
import time

from conftest import ROOT

import bench


def test_timed_region_counts_exactly_k_steps():
    calls = []
    dt = bench.timed_region(lambda i: calls.append(i), steps=7, warmup=3, sync=lambda: None)
    assert calls == list(range(10)) and dt >= 0


def test_sustained_region_runs_whole_chunks_until_the_clock_says_so():
    calls, now = [], [0.0]

    def step(i):
        calls.append(i)
        now[0] += 0.001            # each step "takes" 1 ms of the fake clock

    n, secs, chunks = bench.sustained_region(step, 0.25, lambda: None, chunk=100, clock=lambda: now[0])
    assert n == 300 and calls == list(range(300)) and len(chunks) == 3          # 0.1, 0.2 < 0.25 <= 0.3
    assert abs(secs - 0.3) < 1e-9 and all(abs(c - 1.0) < 1e-6 for c in chunks)   # ms per step of each chunk
    assert bench.sustained_region(step, 0.0, lambda: None) == (0, 0.0, [])       # --sustain-seconds 0 skips it


def test_aggregate_value_is_whole_job():
    assert bench.aggregate_value(1e9, 10, 1, 10.0) == 1.0
    assert bench.aggregate_value(1e9, 10, 8, 10.0) == 8.0       # N replicas: N x the elements over the same time


def test_roofline_entry_math():
    e = bench.roofline_entry("k", 8_000_000_000, 1000.0)
    assert e["achieved"] == 8.0 and e["frac"] == 0.001 and e["peak"] == 8000.0 and e["bound"] == "hbm"
    assert "frac_algorithmic" not in e            # the kernel moves exactly its algorithmic bytes: one figure only
    e = bench.roofline_entry("k", 8_000_000_000, (1000.0, [900.0, 1000.0, 1100.0]), traffic=4_000_000_000)
    assert e["traffic_gbs"] == 4.0 and e["traffic_frac"] == 0.0005 and e["us_p10_p50_p90"] == [900000.0, 1000000.0, 1100000.0]
    assert e["traffic_source"]                    # a traffic figure always says where it was measured


def test_frac_is_a_byte_rate_never_above_one():
    """the mask backward moves 4 B/elem, not the 6 B/elem of the reference's data flow: `frac` follows the bytes moved,
    the 6 B/elem accounting is reported under its own, labelled key (frac 1.14 was an accounting figure)"""
    n = 45_088_768
    ms = 30.4e-3                                   # 30.4 us: the measured mask backward of the metric tensor
    e = bench.roofline_entry("ste_bwd", n * 6, ms, moved_bytes=n * 4)
    assert e["frac"] <= 1.0 and abs(e["achieved"] - n * 4 / 30.4e-6 / 1e9) < 0.1
    assert e["frac_algorithmic"] > 1.0 and "not a byte rate" in e["frac_algorithmic_note"]
    assert e["bytes_moved_per_launch"] == n * 4 and e["algorithmic_bytes_per_launch"] == n * 6


WORKER = r"""
import os, sys, time, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank = dist.get_rank()
def step(i):
    time.sleep(0.02 if rank == 0 else 0.05)      # rank 1 is the slow one: max-over-ranks must see it
dt = bench.timed_region(step, steps=4, warmup=1, sync=lambda: None, dist_mod=dist)
val = bench.aggregate_value(1000, 4, dist.get_world_size(), dt)
if rank == 0:
    print(json.dumps({{"dt": dt, "value": val, "world": dist.get_world_size()}}))
dist.destroy_process_group()
"""


def test_two_rank_gloo_max_over_ranks(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2
    assert res["dt"] >= 4 * 0.05 * 0.95                   # the slow rank's time, not rank 0's
    assert abs(res["value"] - 1000 * 4 * 2 / res["dt"] / 1e9) < 1e-12


# ---- `python bench.py --gpus N` with NO launcher: bench.py starts the N ranks itself ----
def _run_bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_without_launcher_spawns_two_ranks():
    p, out = _run_bench(["--gpus", "2", "--stub", "--steps", "6", "--warmup", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dist_backend"] == "gloo"
    assert out["data"] == "stub" and out["stub"] is True          # can never be mistaken for a measurement
    assert out["steps"] == 6 and out["warmup"] == 2
    # the slow rank (rank 1 sleeps 4 ms per step) sets the time: max over ranks, whole-job value = 2 ranks' elements
    assert out["ms_per_step"] >= 4.0 * 0.9
    assert abs(out["value"] - 1000 * 6 * 2 / (out["ms_per_step"] * 6 / 1e3) / 1e9) < 1e-9


def test_gpus_4_stub_counts_every_rank():
    p, out = _run_bench(["--gpus", "4", "--stub", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 4 and out["ranks_seen"] == 4


def test_gpus_8_stub_is_the_drivers_config_5_command_end_to_end():
    """BASELINE configs[4] (LLaMA-13B, 8 x MI355X, outer DDP loop only): the exact shape of the driver's command, `bench.py --gpus 8 --steps K
    --warmup W`, rehearsed on CPU with the stub step -- eight spawned ranks, one line, every rank counted, weak scaling (replicas)"""
    p, out = _run_bench(["--gpus", "8", "--stub", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["scaling"] == "weak" and out["data"] == "stub"
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1
    # whole-job value: eight ranks' elements over the slowest rank's time
    assert abs(out["value"] - 1000 * 3 * 8 / (out["ms_per_step"] * 3 / 1e3) / 1e9) < 1e-9


def test_gpus_8_under_the_launcher_command(tmp_path):
    """the same through `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py ...`,
    as the driver launches N > 1"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]      # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8


def test_never_reports_fewer_ranks_than_asked():
    # a launcher that started the wrong number of ranks: refuse, print no JSON line
    p, out = _run_bench(["--gpus", "2", "--stub", "--steps", "2", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert p.returncode != 0 and out is None and "refusing" in p.stderr


def test_a_failing_rank_fails_the_job(tmp_path):
    # rank 1 cannot join (its rendezvous port is closed to it): the parent must exit non-zero, not hang, not print n_gpus 1
    p, out = _run_bench(["--gpus", "2", "--stub", "--steps", "2", "--warmup", "0"], {"BENCH_INIT_TIMEOUT_S": "5", "BENCH_TEST_KILL_RANK": "1"})
    assert p.returncode != 0 and out is None


def test_one_gpu_default_has_no_process_group_keys():
    p, out = _run_bench(["--stub", "--steps", "2", "--warmup", "0"])
    assert p.returncode == 0 and out["n_gpus"] == 1 and out["ranks_seen"] == 1 and "dist_backend" not in out


# ---- the line the driver parses (round 2's single 27 KB line was cut by the driver's 8 KB tail) ----
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "roofline_step", "cpu_baseline", "ranks_seen")


def _full_record():
    """a realistic full record: round 2's 27 KB line, re-keyed the way round 4 reports (headline = the step that writes both gradients)"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    full.update({"value": 744.34, "ms_per_step": 0.1212, "value_product_default": 966.66, "ms_per_step_product_default": 0.0933,
                 "backward_elements_touched": 2 * 45088768})
    full["config"]["elements_per_step"] = 2 * 45088768
    full["roofline_step"].update({"us_per_step_launches": 118.2, "bytes_moved_per_step": 732692480, "achieved": 6198.8, "frac": 0.7748})
    full["roofline_step"]["product_default"] = {"bytes_moved_per_step": 552337408, "us_per_step_launches": 90.7, "achieved": 6089.7, "frac": 0.7612}
    full["roofline_step"].pop("out_of_place", None)
    return full


def test_headline_is_compact_and_complete():
    full = _full_record()
    assert len(json.dumps(full)) > 20000                       # the record itself is far beyond what the driver's tail keeps
    line = bench.compact_headline(full, "bench_extras.json")
    assert len(line.encode()) < bench.HEADLINE_MAX_BYTES == 4096
    h = json.loads(line)
    for k in HEADLINE_KEYS + ("value_product_default", "backward_elements_touched", "self_check"):
        assert k in h, k
    assert len(json.dumps(h["config"])) <= 400 and "model" not in h["config"] and h["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "kernel"):
        assert k in h["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "parity_gate"):
        assert k in h["cpu_baseline"], k
    assert h["roofline_step"]["frac"] <= 1.0 and h["roofline_step"]["product_default"]["frac"] <= 1.0
    assert h["self_check"]["ok"] is True
    assert "hbm_gbs_algorithmic" not in h                      # a GB/s above the 8 TB/s peak has no place in the headline
    assert h["value"] == full["value"] and h["ms_per_step"] == full["ms_per_step"] and h["roofline"]["frac"] == full["roofline"]["frac"]


def test_headline_stays_compact_whatever_the_extras_grow_to():
    full = _full_record()
    for i in range(60):                                        # a future round registers 60 more kernel families
        full[f"kernels_future_{i}"] = [dict(full["kernels"][0]) for _ in range(10)]
    full["config"]["workload"] = "W" * 5000
    full["roofline"]["kernel"] = "K" * 5000
    full["cpu_baseline"]["sample"] = "S" * 5000
    assert len(bench.compact_headline(full, "x").encode()) < 4096


def test_emit_prints_one_stdout_line_and_the_extras_on_stderr(capsys):
    import argparse
    full = _full_record()
    bench.emit(full, argparse.Namespace(no_sidecar=True))
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1, "stdout must carry exactly ONE JSON line (the launch contract)"
    last = json.loads(lines[0])
    assert len(lines[0].encode()) < 4096 and "bench_extras" not in last and last["value"] == full["value"]
    extras = cap.err.strip().splitlines()
    assert len(extras) > 5 and all(json.loads(ln).get("bench_extras") for ln in extras)
    fams = {json.loads(ln)["bench_extras"] for ln in extras}
    assert {"kernels", "kernels_step", "kernels_model_shapes", "kernels_export", "gpu_eager"} <= fams


def test_stub_line_is_compact_too():
    p, out = _run_bench(["--stub", "--steps", "2", "--warmup", "0"])
    assert p.returncode == 0
    last = p.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 4096 and json.loads(last)["data"] == "stub"


# ---- the cross-checks a reader can redo from the line alone --------------------------------------------
def test_self_check_flags_round_3s_headline_and_passes_the_like_for_like_one():
    """BENCH_r03.json: value 966.66 Gelem/s at 0.0933 ms/step is 10 B x 90.2 M elements / 93.3 us = 9.67 TB/s under SURVEY §8d's
    accounting -- above the 8 TB/s peak (the timed step's W4 backward moved nothing); and its separately timed launches summed to
    101.81 us, more than the 93.3 us step that contains them (20 launches after 4 warm-ups).  Both must be flagged."""
    elems = 2 * 45088768
    r03 = {"ms_per_step": 0.0933, "value": 966.66, "n_gpus": 1, "config": {"elements_per_step": elems},
           "roofline": {"frac": 0.7443}, "roofline_step": {"frac": 0.678, "us_per_step_launches": 101.81}}
    sc = bench.self_check(r03)
    assert sc["ok"] is False and sc["accounting_below_peak"] is False and sc["launch_sum_within_step"] is False
    assert 9600 < sc["accounting_gbs_10B_per_elem"] < 9700
    good = {"ms_per_step": 0.1212, "value": 744.34, "n_gpus": 1, "config": {"elements_per_step": elems},
            "roofline": {"frac": 0.757}, "roofline_step": {"frac": 0.775, "us_per_step_launches": 118.2}}
    sc = bench.self_check(good)
    assert sc["ok"] is True and sc["accounting_gbs_10B_per_elem"] <= bench.HBM_PEAK_GBS and sc["launch_sum_vs_step"] <= 1.03
    assert bench.self_check({"stub": True, "ms_per_step": 1.0}) is None


def test_kernel_timing_protocol_is_independent_of_steps():
    """every per-kernel entry: >= 200 launches after >= 20 warm-ups whatever --steps says; us_per_launch = their back-to-back mean"""
    class _Ev:
        clock = [0.0]

        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = _Ev.clock[0]

        def elapsed_time(self, other):
            return other.t - self.t

    class _Cuda:
        Event = _Ev

        @staticmethod
        def synchronize():
            pass

    class _Torch:
        cuda = _Cuda

    calls = []

    def fn(s):
        calls.append(s)
        _Ev.clock[0] += 0.05 if len(calls) % 10 else 0.5     # every tenth launch is an outlier: the p50 ignores it, the mean does not

    del bench.PROFILE_MANIFEST[:]
    t = bench.time_launches(_Torch, fn, 20, [0, 1, 2, 3], name="k")        # the driver's --steps 20
    assert t.iters >= 200 and t.warmup >= 20 and len(calls) == t.warmup + 2 * t.iters
    ms, pct = t
    assert abs(pct[1] - 0.05) < 1e-9 and abs(ms - (0.05 * 0.9 + 0.5 * 0.1)) < 1e-6
    assert bench.PROFILE_MANIFEST[-1] == ("k", t.warmup + 2 * t.iters)
    e = bench.roofline_entry("k", 1000, t)
    assert e["launches_timed"] >= 200 and e["warmup_launches"] >= 20 and abs(e["us_per_launch"] - 95.0) < 0.01
