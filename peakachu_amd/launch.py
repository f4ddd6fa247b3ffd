"""One command, every GPU of the node: `peakachu-amd score_genome ...` started bare on a box with
several GPUs becomes a launcher that runs one copy of itself per GPU as CHILD processes --
decided before anything has touched HIP (a process that has initialised the GPU must never
exec or fork another GPU user) -- with RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous file of
their own in the environment; an external launcher (the bench driver's, srun, mpirun) that sets the same
variables works as before and makes this module a no-op.

The reference has nothing like it (one process, one thread: peakachu/score_genome.py:46); its
command line, documented at README.md:127, is what stays.
"""
import os
import subprocess
import sys
import tempfile
import time


def visible_gpus():
    """GPUs this process would see, counted WITHOUT a HIP call: the KFD topology's nodes with
    SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(root, node, "properties")) if line.strip())
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(",") if t.strip() != ""]
            listed = 0
            for t in ids:
                if t.strip().startswith("-"):
                    break  # (-1 ends the list: how one hides every device)
                listed += 1
            n = min(n, listed)
    return n


def wanted_ranks():
    """How many ranks a bare start should fan out to: PK_DEVICES (a number, or `all`), else every
    visible GPU; 1 (no fan-out) when a launcher has set WORLD_SIZE already or PK_NO_SPAWN=1."""
    if "WORLD_SIZE" in os.environ or os.environ.get("PK_NO_SPAWN") == "1":
        return 1
    want = os.environ.get("PK_DEVICES", "all").strip().lower()
    n = visible_gpus()
    if want not in ("", "all"):
        n = min(n, int(want)) if n else int(want)
    return max(1, n)


def spawn(n, argv=None, env_extra=None):
    """Runs `n` copies of this very command (sys.orig_argv) as children, rank r on GPU r; returns
    the exit code to leave with: 0 when all ranks did, else the first failure's (the others are
    told to stop: a rank that is gone can no longer take part in the gather)."""
    argv = list(argv if argv is not None else sys.orig_argv[1:])
    fd, rfile = tempfile.mkstemp(prefix="pk_rdzv_", suffix=".json")
    os.close(fd)
    os.unlink(rfile)  # (rank 0 creates it; the name is what the children share)
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "PK_RDZV_FILE": rfile, "PK_LAUNCHED_BY": "peakachu_amd.launch"})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
            if env_extra:
                env.update(env_extra)
            procs.append(subprocess.Popen([sys.executable] + argv, env=env))
        code = 0
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    sys.stderr.write("peakachu_amd.launch: rank %d ended with code %d; stopping the others\n" % (r, rc))
                    deadline = time.monotonic() + 15.0   # they notice by themselves (rendezvous) ...
                    while time.monotonic() < deadline and any(procs[q].poll() is None for q in alive):
                        time.sleep(0.1)
                    for q in alive:                      # ... or are told
                        if procs[q].poll() is None:
                            procs[q].terminate()
            time.sleep(0.05)
        return code
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink(rfile)
        except OSError:
            pass
