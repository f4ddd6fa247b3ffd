"""The N > 1 host path on CPUs: the product's own rendezvous (peakachu_amd.rendezvous: TCP,
standard library), its launcher (peakachu_amd.launch) and the sharding / gather / merge code of
peakachu_amd.dist + score_genome, with the CPU oracle as the per-rank scorer.  The reference has
one process (peakachu/score_genome.py:46-84); nothing here has a counterpart there."""
import os
import socket
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peakachu_amd import launch, rendezvous  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(**kw):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PK_RDZV_FILE", "PK_RDZV_ENDPOINT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_two_ranks_under_the_drivers_launcher(tmp_path):
    """Launched exactly like the driver launches bench.py (python -m torch.distributed.run):
    the ranks find each other through the file named after MASTER_ADDR / MASTER_PORT / run id
    (MASTER_PORT itself belongs to that launcher's store) and the merged bedpe equals the
    reference's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "result.txt").read_text() == "OK"


def test_three_ranks_under_the_products_launcher(tmp_path):
    """peakachu_amd.launch.spawn: three children, RANK / WORLD_SIZE / PK_RDZV_FILE from the
    launcher; same worker, same result (LPT dealing and block ranges for three ranks)."""
    code = subprocess.run([sys.executable, "-c",
                           "import sys; sys.path.insert(0, %r); from peakachu_amd import launch; "
                           "sys.exit(launch.spawn(3, argv=[%r, %r]))"
                           % (ROOT, os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path))],
                          cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert code.returncode == 0, code.stdout[-2000:] + code.stderr[-2000:]
    assert (tmp_path / "result.txt").read_text() == "OK"


def test_rendezvous_collectives_in_threads(tmp_path, monkeypatch):
    monkeypatch.setenv("PK_RDZV_FILE", str(tmp_path / "rdzv.json"))
    monkeypatch.delenv("PK_RDZV_ENDPOINT", raising=False)
    world, out, errs = 4, {}, []

    def rank_main(rank):
        try:
            r = rendezvous.Rendezvous(rank=rank, world=world, timeout=30)
            res = [r.all_gather(b"x" * rank), r.broadcast(b"id-128" if rank == 0 else None),
                   r.gather(bytes([rank])), r.all_gather_obj(None if rank % 2 else "odd? no: %d" % rank)]
            r.barrier()
            big = os.urandom(3 << 20) if rank == 1 else b""      # a payload of several MB
            res.append([len(p) for p in r.all_gather(big)])
            r.close()
            out[rank] = res
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    assert not errs, errs
    for rank in range(world):
        ag, bc, ga, objs, sizes = out[rank]
        assert ag == [b"x" * k for k in range(world)] and bc == b"id-128"
        assert ga == ([bytes([k]) for k in range(world)] if rank == 0 else None)
        assert objs == ["odd? no: 0", None, "odd? no: 2", None]
        assert sizes == [0, 3 << 20, 0, 0]
    assert not (tmp_path / "rdzv.json").exists()   # rank 0 removes what it published


def test_a_stale_file_is_not_followed(tmp_path, monkeypatch):
    """A file left by a job that is gone (dead port, another token): the peers keep reading
    until rank 0 has published the new one."""
    f = tmp_path / "rdzv.json"
    f.write_text('{"host": "127.0.0.1", "port": %d, "token": "old", "world": 2, "pid": 1}' % _free_port())
    monkeypatch.setenv("PK_RDZV_FILE", str(f))
    res = {}

    def peer():
        r = rendezvous.Rendezvous(rank=1, world=2, timeout=30)
        res["peer"] = r.broadcast()
        r.close()

    t = threading.Thread(target=peer)
    t.start()
    time.sleep(0.5)   # the peer meets the stale file first
    r0 = rendezvous.Rendezvous(rank=0, world=2, timeout=30)
    r0.broadcast(b"fresh")
    r0.close()
    t.join(30)
    assert res.get("peer") == b"fresh"


def _run_ranks(mode, world, tmp_path):
    env = _clean_env(PK_RDZV_FILE=str(tmp_path / ("rdzv_%s.json" % mode)), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rdzv_worker.py"), mode],
                              env=dict(env, RANK=str(k), LOCAL_RANK=str(k)), stderr=subprocess.PIPE, text=True)
             for k in range(world)]
    t0 = time.monotonic()
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], [o[1] for o in outs], time.monotonic() - t0


def test_a_failing_rank_and_a_failing_root_strand_nobody(tmp_path):
    codes, errs, _ = _run_ranks("ok", 3, tmp_path)
    assert codes == [0, 0, 0], errs
    # the last rank dies: rank 0 notices the closed socket in the next operation and tells rank 1
    codes, errs, dt = _run_ranks("peer_dies", 3, tmp_path)
    assert codes == [7, 7, 3], (codes, errs)
    assert "rank 2 left the job" in errs[0] and "rank 2 left the job" in errs[1]
    assert dt < 60
    # rank 0 dies: both peers see the end of their stream
    codes, errs, dt = _run_ranks("root_dies", 3, tmp_path)
    assert codes == [4, 7, 7], (codes, errs)
    assert "lost rank 0" in errs[1] and "lost rank 0" in errs[2]
    assert dt < 60


def test_launcher_passes_a_failure_on_and_stops_the_rest(tmp_path):
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); from peakachu_amd import launch; "
                        "sys.exit(launch.spawn(3, argv=[%r, 'peer_dies']))"
                        % (ROOT, os.path.join(ROOT, "tests", "_rdzv_worker.py"))],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode in (3, 7), (r.returncode, r.stderr[-1500:])
    assert "stopping the others" in r.stderr


def test_gpu_count_without_touching_hip(monkeypatch, tmp_path):
    """visible_gpus reads the KFD topology and the *_VISIBLE_DEVICES lists; wanted_ranks never
    fans out under a launcher or when told not to."""
    n = launch.visible_gpus()
    assert n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "-1")
    assert launch.visible_gpus() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,-1,3")
    assert launch.visible_gpus() <= 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert launch.wanted_ranks() == 1
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setenv("PK_NO_SPAWN", "1")
    assert launch.wanted_ranks() == 1
    monkeypatch.delenv("PK_NO_SPAWN")
    # the fan-out is opt-in, and what sysfs lists must be confirmed by a probe child that asks HIP
    assert launch.wanted_ranks() == 1
    monkeypatch.setenv("PK_DEVICES", "3")
    want = 3 if n == 0 else min(n, 3)
    monkeypatch.setattr(launch, "probed_gpus", lambda: 8)
    assert launch.wanted_ranks() == want
    monkeypatch.setattr(launch, "probed_gpus", lambda: None)       # the probe could not run
    assert launch.wanted_ranks() == 1
    monkeypatch.setattr(launch, "probed_gpus", lambda: want - 1)    # HIP opens fewer than are listed
    assert launch.wanted_ranks() == (1 if want > 1 else want)
    monkeypatch.undo()
    assert launch.probed_gpus() in (None, 0) or launch.probed_gpus() >= 1   # (runs: a child, no HIP in this process)


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus 2` (no launcher, as the driver starts --gpus 1) must start its
    own ranks as child processes: here, without a GPU, both ranks get through the rendezvous
    and stop at the library's device check -- loudly, and the launcher passes the failure on."""
    env = _clean_env(OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("no HIP device visible") >= 2, r.stderr[-3000:]  # one per rank
    assert "must be launched with" not in r.stderr
    # no JSON line from a run that measured nothing
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")], r.stdout[-2000:]
