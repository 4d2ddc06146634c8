"""bench.py as the driver runs it: the one-GPU line, and the N = 2 launch line of the scaling run rehearsed on ONE GPU
(`--backend gloo`: the same ShardedPair control flow -- disparity shards, one MIN all-reduce of the packed keys per step,
a fixed number of pre-heat steps, fences, MAX over ranks -- with the keys staged through host memory).  The RCCL leg itself
needs one GPU per rank and has never run on this pool."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def test_single_gpu_line_tsukuba():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tsukuba", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--preheat-s", "0.05"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _last_json_line(p.stdout)
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["unit"] == "MPix/s" and r["value"] > 0
    assert r["roofline"]["scope"] == "walker_kernel" and 0 < r["roofline"]["frac"] < 1
    # the chunk that ran: all 16 slices in one walker launch
    assert r["config"]["slices_in_flight"] == 16 and r["config"]["walker_launches_per_call"] == 1


def test_slices_in_flight_is_what_ran():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tsukuba", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--preheat-s", "0", "--slices-in-flight", "5"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _last_json_line(p.stdout)
    assert r["config"]["slices_in_flight"] == 5 and r["config"]["walker_launches_per_call"] == 4


def test_cost_volume_source_line():
    """`--source cost`: the reference's data flow -- the raw volumes resident in HBM, the comb walker reads p and writes q."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tsukuba", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--preheat-s", "0", "--source", "cost"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _last_json_line(p.stdout)
    assert r["source"] == "cost" and "not the headline" in r["metric"] and r["value"] > 0
    assert r["config"]["walker_launches_per_call"] == 1 and 0 < r["roofline"]["frac"] < 1


def test_two_ranks_launch_line_on_one_gpu():
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank) with two ranks sharing cuda:0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload",
           "tsukuba", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = _last_json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 2 and r["scaling"] == "strong"
    assert r["config"]["sharding"] == "disparity slices / 2 ranks" and r["config"]["backend"] == "gloo"
    assert r["config"]["slices_in_flight"] == 8            # rank 0's shard: 8 of the 16 slices in one launch
    assert r["preheat_steps"] == 200 and r["value"] > 0 and "cpu_baseline" not in r
