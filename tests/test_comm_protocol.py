"""The gather protocol of pk_comm.hip (peakachu_amd/csrc/pk_comm_protocol.h: what every rank tells
the others before anybody posts a send or a receive) run by 1, 2, 3 and 8 THREADS over a fabric whose
send / recv block like RCCL's, with local failures injected: root capacity too small, a staging area
that cannot grow, a refused copy, a peer that cannot stage its bytes.  Nobody may be left waiting
and the next call must work.  The RCCL build uses the same header; only what moves the bytes differs.
(The reference gathers nothing: one process, peakachu/score_genome.py:46-84.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


def _sanitizer_works(flag, tmp_path):
    """g++ in this image ships libtsan / libasan / libubsan; elsewhere the build may lack them."""
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    r = subprocess.run(["g++", flag, "-o", str(tmp_path / "probe"), str(probe)], capture_output=True)
    return r.returncode == 0


@pytest.mark.parametrize("flags,scale", [((), 1), (("-fsanitize=thread",), 10), (("-fsanitize=address,undefined",), 4)],
                         ids=["plain", "tsan", "asan-ubsan"])
def test_gather_protocol_with_injected_failures(tmp_path, flags, scale):
    """Plain, under ThreadSanitizer (the protocol's threads share mailboxes, counters and staging
    areas) and under Address + UndefinedBehaviour sanitizers: a report of any of them fails the run
    (halt_on_error / exitcode), waits that count as deadlocks are stretched for the slower builds."""
    if flags and not _sanitizer_works(flags[0], tmp_path):
        pytest.skip("g++ cannot link %s here" % flags[0])
    exe = str(tmp_path / "test_comm_protocol")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-Wall", *flags, *(["-fno-sanitize-recover=all"] if flags else []), "-o", exe, os.path.join(ROOT, "tests", "native", "test_comm_protocol.cpp")], check=True, cwd=ROOT)
    env = dict(os.environ, PK_TEST_WAIT_SCALE=str(scale), TSAN_OPTIONS="halt_on_error=1 exitcode=66",
               ASAN_OPTIONS="detect_leaks=1 exitcode=67", UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:] + r.stderr[-3000:]
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
