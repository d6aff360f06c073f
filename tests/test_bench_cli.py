"""bench.py as the driver invokes it: `python bench.py --gpus N` starts its own ranks, prints one JSON
line on stdout and exits 0; a WORLD_SIZE that contradicts --gpus is an error before anything is loaded."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """Two ranks on the one GPU of the box, ghosts staged through the host (gloo): the N > 1 code path of
    bench.py from the plain command line, strong scaling of a 128 x 512 x 512 grid."""
    env = dict(os.environ, FDTD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "512", "--global-x", "128", "--steps", "6",
                        "--warmup", "2", "--reps", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2
    assert out["scaling"] == "strong" and out["steps"] == 6 and out["reps"]["n"] == 2
    assert out["value"] > 0 and "128x512x512" in out["config"]["workload"]
    assert "exchange" in out
    assert out["config"]["transport"] == "rccl" and out["ranks_over_rccl"]["value"] == out["value"]
    assert "skipped" in out["one_process_peer_copies"]      # one GPU visible, two ranks


@pytest.mark.gpu
def test_bench_under_the_distributed_launcher():
    """The other way a driver starts N ranks: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
    (both ranks on the one GPU, ghosts over gloo); rank 0 prints the one line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, FDTD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--grid", "512",
                        "--global-x", "96", "--steps", "4", "--warmup", "2", "--reps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and "96x512x512" in out["config"]["workload"]


@pytest.mark.gpu
def test_bench_ranks_then_the_one_process_session():
    """After the ranks' run, rank 0 measures the same workload through the multi-device session in a process of its own;
    the line carries both and `value` is the faster one.  (Forced here: the box has one GPU, the ranks share it and so
    do the session's slabs.  By default the second run happens when rank 0 sees as many GPUs as there are ranks.)"""
    env = dict(os.environ, FDTD_BENCH_BACKEND="gloo", FDTD_BENCH_PEER_LEG="force")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "512", "--global-x", "128", "--steps", "6",
                        "--warmup", "2", "--reps", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    both = (out["ranks_over_rccl"], out["one_process_peer_copies"])
    assert both[0]["value"] > 0 and both[1]["value"] > 0, both
    assert "128x512x512" in both[1]["workload"]
    assert out["value"] == max(b["value"] for b in both)
    assert out["config"]["transport"] == ("rccl" if both[0]["value"] >= both[1]["value"] else "peer copies (one process)")
    assert out["n_gpus"] == 2 and out["steps"] == 6


@pytest.mark.gpu
def test_bench_single_gpu_line():
    """The N = 1 line at a reduced size: repetitions, roofline and CPU baseline objects are present."""
    r = subprocess.run([sys.executable, BENCH, "--grid", "256", "--steps", "10", "--warmup", "5", "--reps", "3",
                        "--cpu-steps", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["reps"]["n"] == 3 and len(out["reps"]["Mcells_per_s"]) == 3
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"]
    assert out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["one_thread"]["cores"] == 1


@pytest.mark.gpu
def test_bench_at_a_higher_space_order():
    """--space-order 8: arrays padded by 8 cells, one step per sweep through the library's time loop."""
    r = subprocess.run([sys.executable, BENCH, "--grid", "256", "--steps", "10", "--warmup", "5", "--reps", "2",
                        "--space-order", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "8th-order" in out["config"]["workload"] and out["value"] > 0 and "cpu_baseline" not in out
    assert out["roofline"]["cell_updates_per_launch"] == 256 ** 3


@pytest.mark.gpu
def test_bench_peer_transport_in_one_process():
    """--transport peer: the library's multi-device session, here two slabs on the one GPU; the line prices the
    exchange against the same schedule without pushes and reports the host's issue time per pair."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FDTD_MULTIGPU_DEVICES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--transport", "peer", "--slabs", "2", "--grid", "512", "--global-x", "128",
                        "--steps", "6", "--warmup", "2", "--reps", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    ex = out["exchange"]
    assert out["config"]["slabs"] == 2 and "128x512x512" in out["config"]["workload"] and out["value"] > 0
    assert "hipMemcpyPeerAsync" in ex["transport"] and ex["issuing"] == "one host thread per slab"
    for key in ("exposed_ms_per_step", "sweep_only_ms_per_step", "host_issue_us_per_pair", "one_issuing_thread"):
        assert key in ex, key


@pytest.mark.gpu
def test_bench_both_transports_in_one_line():
    """--transport both: the RCCL run (here the one-rank rehearsal with itself as neighbour) and the peer run, each in
    a process of its own; one line, the peer run's figures under `exchange_peer`."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FDTD_MULTIGPU_DEVICES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--self-exchange", "--transport", "both", "--slabs", "2",
                        "--grid", "512", "--global-x", "128", "--steps", "12", "--warmup", "2", "--reps", "1",
                        "--no-breakdown"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert "RCCL" in out["exchange"]["transport"]
    assert "hipMemcpyPeerAsync" in out["exchange_peer"]["transport"] and out["exchange_peer"]["value"] > 0
