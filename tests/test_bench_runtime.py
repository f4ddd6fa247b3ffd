"""bench.py's ranks run on the product's own ROCm, not on a copy a framework bundles.

torch 2.10+rocm7.0 ships libamdhip64 / librccl / libhsa-runtime64 under the SAME sonames as
/opt/rocm's; a soname binds once per process, so a rank that imported torch first would run
libpeakachu_hip.so on torch's HIP and torch's RCCL.  The ranks therefore import no framework
(torch.distributed.run stays the launcher: a parent that never touches HIP), the JSON line says
which runtime ran, and a run on a foreign copy is refused (exit 3).  No device needed here."""
import ast
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _imports(path):
    names = set()
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.Import):
            names.update(a.name for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            names.add(node.module or "")
    return names


def test_bench_and_the_package_import_no_framework():
    for path in [BENCH, os.path.join(ROOT, "__graft_entry__.py")] + \
            [os.path.join(ROOT, "peakachu_amd", f) for f in os.listdir(os.path.join(ROOT, "peakachu_amd"))
             if f.endswith(".py")]:
        bad = [n for n in _imports(path) if n.split(".")[0] in ("torch", "triton", "jax", "tensorflow", "cupy")]
        assert not bad, (path, bad)
    # the launcher's command line is the only mention of torch in bench.py's code
    src = open(BENCH).read()
    code_hits = [ln for ln in src.splitlines() if "torch" in ln and not ln.lstrip().startswith("#")
                 and '"torch.distributed.run"' in ln]
    assert len(code_hits) == 1 and "cmd = [" in code_hits[0]


_PROBE = r"""
import json, os, sys
sys.path.insert(0, %r)
%s
import numpy
from peakachu_amd import _lib, dist, forest            # what bench.py's ranks import
from peakachu_amd.rendezvous import Rendezvous
_lib.load()
info = _lib.runtime_info()
info["torch_mapped"] = sorted({l.split()[-1] for l in open("/proc/self/maps") if "/torch/lib/" in l})
print(json.dumps(info))
"""


def _probe(first=""):
    r = subprocess.run([sys.executable, "-c", _PROBE % (ROOT, first)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_the_ranks_bind_the_rocm_ldd_names():
    info = _probe()
    assert info["torch_mapped"] == []
    assert info["product_runtime"] is True, info
    for k in ("librccl.so", "libamdhip64.so", "libhsa-runtime64.so"):
        p = info["rocm_libs"][k]
        assert isinstance(p, str) and os.path.dirname(os.path.realpath(p)) == info["rocm_dir_ldd"], info
    assert info["hip_runtime_version"] > 0 and info["rccl_version"] > 0


def test_a_foreign_runtime_is_noticed_and_refused():
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        import pytest
        pytest.skip("no framework with a bundled ROCm in this environment")
    info = _probe("import torch.distributed")
    if not info["torch_mapped"]:
        import pytest
        pytest.skip("this torch build bundles no ROCm")
    assert info["product_runtime"] is False, info
    # bench.py itself, entered in a process that mapped the foreign copy first: exit 3 before any device call
    code = ("import torch.distributed, runpy, sys; sys.argv = ['bench.py', '--steps', '1']; "
            "runpy.run_path(%r, run_name='__main__')" % BENCH)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PK_BENCH_ANY_RUNTIME")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "refusing to measure another runtime" in r.stderr


def test_profile_stamp_ignores_comments_and_layout():
    """tools/srchash.py: the stamp of profiles/pmc*.json changes with the code, not with a note."""
    sys.path.insert(0, ROOT)
    from tools.srchash import normalised
    code = 'int f(int a) { return a /* the input */ + 1; }  // adds one\nconst char *s = "// kept /* too */";\n'
    same = 'int f(int a)\n{\n    return a + 1;   // a different note\n}\nconst char *s = "// kept /* too */";'
    other = 'int f(int a) { return a + 2; }\nconst char *s = "// kept /* too */";'
    assert normalised(code) == normalised(same) != normalised(other)
    assert normalised('x = "a  b";') != normalised('x = "a b";')      # literals are code (inline asm)
    import bench
    from tools import make_traffic
    assert bench.source_sha() == make_traffic.source_sha() and len(bench.source_sha()) == 16
