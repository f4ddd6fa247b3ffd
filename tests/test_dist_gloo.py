"""The N>1 path on CPUs: world_size 2, gloo, launched exactly like the
driver launches bench.py (python -m torch.distributed.run)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather_matches_reference(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "result.txt").read_text() == "OK"


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus 2` (no launcher, as the driver starts --gpus 1) must start its
    own ranks under torch.distributed.run as child processes: here, without a GPU, both ranks
    get through the gloo rendezvous and stop at the library's device check -- loudly, and the
    launcher passes the failure on."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("no HIP device visible") >= 2, r.stderr[-3000:]  # one per rank
    assert "must be launched with" not in r.stderr
    # no JSON line from a run that measured nothing
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")], r.stdout[-2000:]
