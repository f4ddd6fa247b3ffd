"""The gather protocol of pk_comm.hip (peakachu_amd/csrc/pk_comm_protocol.h: what every rank tells
the others before anybody posts a send or a receive) run by 1, 2, 3 and 8 THREADS over a fabric whose
send / recv block like RCCL's, with local failures injected: root capacity too small, a staging area
that cannot grow, a refused copy, a peer that cannot stage its bytes.  Nobody may be left waiting
and the next call must work.  The RCCL build uses the same header; only what moves the bytes differs.
(The reference gathers nothing: one process, peakachu/score_genome.py:46-84.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_protocol_with_injected_failures(tmp_path):
    exe = str(tmp_path / "test_comm_protocol")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "test_comm_protocol.cpp")], check=True, cwd=ROOT)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:] + r.stderr[-1000:]
