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


def probed_gpus():
    """GPUs the HIP runtime itself opens, asked of a short-lived CHILD (pk_device_count of the
    package's library): sysfs lists every GPU of the host even where a container or lease may only
    open some of them.  The launcher itself stays free of HIP.  None when the probe cannot run."""
    code = "from peakachu_amd import _lib; print(_lib.load().pk_device_count())"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=root)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return None


def wanted_ranks():
    """How many ranks a bare start should fan out to.  The fan-out is OPT-IN: PK_DEVICES=<n> or
    PK_DEVICES=all asks for it (an N > 1 run over RCCL has not been seen on hardware by the builder:
    the default stays the single process the reference is); 1 when a launcher has set WORLD_SIZE
    already or PK_NO_SPAWN=1.  The count is what sysfs lists, confirmed by a probe child that asks the
    HIP runtime; where the two disagree the command stays one process (and says why)."""
    if "WORLD_SIZE" in os.environ or os.environ.get("PK_NO_SPAWN") == "1":
        return 1
    want = os.environ.get("PK_DEVICES", "").strip().lower()
    if want in ("", "0", "1"):
        return 1
    n = visible_gpus()
    if want != "all":
        n = min(n, int(want)) if n else int(want)
    if n > 1:
        seen = probed_gpus()
        if seen is None or seen < n:
            sys.stderr.write("peakachu_amd.launch: %s GPUs are listed but the HIP runtime opens %s: staying one process\n"
                             % (n, "none (probe failed)" if seen is None else seen))
            return 1
    return max(1, n)


def spawn(n, argv=None, env_extra=None):
    """Runs `n` copies of this very command (sys.orig_argv) as children, rank r on GPU r; returns
    the exit code to leave with: 0 when all ranks did, else the first failure's (the others are
    told to stop: a rank that is gone can no longer take part in the gather)."""
    argv = list(argv if argv is not None else sys.orig_argv[1:])
    # a directory of the launcher's own (mode 0700): rank 0 creates the file in it, nobody else can
    rdir = tempfile.mkdtemp(prefix="pk_rdzv_")
    rfile = os.path.join(rdir, "rendezvous.json")
    procs = []
    # a launcher that is told to stop tells its ranks (they hold the GPUs), then leaves with the
    # signal's code; without this its `finally` never ran and the children lived on
    import signal

    def _forward(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        raise SystemExit(128 + signum)
    previous = {}
    try:
        for sig in (signal.SIGTERM, signal.SIGINT):
            previous[sig] = signal.signal(sig, _forward)
    except ValueError:  # (not the main thread: the caller's own business)
        previous = {}
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
                p.terminate()
        deadline = time.monotonic() + 5.0
        for p in procs:
            while p.poll() is None and time.monotonic() < deadline:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
        for sig, old in previous.items():
            signal.signal(sig, old)
        for path in (rfile,):
            try:
                os.unlink(path)
            except OSError:
                pass
        try:
            os.rmdir(rdir)
        except OSError:
            pass
