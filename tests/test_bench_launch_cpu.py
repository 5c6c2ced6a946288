"""bench.py's launch paths, rehearsed on CPU ranks (gloo + tests/cpu_backend.py through bench.py's --cpu-rehearsal
test seam): `python bench.py --gpus N` with no launcher must start N ranks itself, form an N-rank group, print ONE
JSON line from rank 0 and fail loudly when a rank fails; the torch.distributed.run path must keep working."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_prints_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-rehearsal"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    d = line["config"]["distributed"]
    assert d["world_size_formed"] == 2 and d["ranks_counted_by_allreduce"] == 2
    assert d["launcher"].startswith("self") and d["backend"] == "gloo"
    assert line["metric"].startswith("REHEARSAL ONLY")       # a rehearsal can never pass for a measurement
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert all(v == v for v in line["config"]["train_metrics_last_step"])


def test_self_launch_fails_when_a_rank_fails():
    # without the rehearsal seam the ranks need a GPU; here there is none (and on a GPU box rank 1 has no device of
    # its own): every rank exits non-zero, the parent must report failure and print no JSON line
    env = _clean_env()
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c2_uci"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "failed" in r.stderr


def test_torchrun_launch_still_works():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--cpu-rehearsal"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    d = line["config"]["distributed"]
    assert d["world_size_formed"] == 2 and d["ranks_counted_by_allreduce"] == 2
    assert d["launcher"].startswith("torch.distributed.run")


def test_parent_of_a_self_launch_never_imports_torch():
    """The parent relays; the ranks compute.  Importing bench must not pull torch in (that happens in run_rank)."""
    code = ("import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; "
            "a = bench.parse_args(['--gpus', '4']); assert a.gpus == 4" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
