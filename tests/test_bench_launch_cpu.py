"""`python bench.py --gpus N` must start its own N ranks (one command per node, as the reference's `composer train.py`
does, /root/reference/train_e2e.sh:10), and the torch.distributed.run form the driver uses for N > 1 must keep working.
No GPU: `--rank-probe` makes every rank report itself and exit before it touches a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _probes(stdout):
    out = []
    for line in stdout.splitlines():
        if line.startswith("{") and "rank_probe" in line:
            out.append(json.loads(line))
    return sorted(out, key=lambda d: d["rank_probe"])


def test_no_env_invocation_spawns_its_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rank-probe"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    p = _probes(r.stdout)
    assert [d["rank_probe"] for d in p] == [0, 1] and all(d["world"] == 2 for d in p), r.stdout
    assert all(d["master"] == "127.0.0.1" for d in p)


def test_torchrun_form_still_works():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-probe"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [d["rank_probe"] for d in _probes(r.stdout)] == [0, 1]


def test_single_gpu_default_does_not_spawn():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and "127.0.0.1" in cmd
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rank-probe"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0 and _probes(r.stdout) == [{"rank_probe": 0, "world": 1, "local_rank": 0, "master": None}]


def test_world_size_mismatch_is_a_clear_error():
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
