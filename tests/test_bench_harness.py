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


def test_aggregate_value_is_whole_job():
    assert bench.aggregate_value(1e9, 10, 1, 10.0) == 1.0
    assert bench.aggregate_value(1e9, 10, 8, 10.0) == 8.0       # N replicas: N x the elements over the same time


def test_roofline_entry_math():
    e = bench.roofline_entry("k", 8_000_000_000, 1000.0)
    assert e["achieved"] == 8.0 and e["frac"] == 0.001 and e["peak"] == 8000.0 and e["bound"] == "hbm"
    e = bench.roofline_entry("k", 8_000_000_000, (1000.0, [900.0, 1000.0, 1100.0]), traffic=4_000_000_000)
    assert e["traffic_gbs"] == 4.0 and e["traffic_frac"] == 0.0005 and e["us_p10_p50_p90"] == [900000.0, 1000000.0, 1100000.0]


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
