"""The host C++ of the contact-map reader under sanitizers ("race detection /
sanitizers ... new work").  peakachu_amd/csrc/pk_hostio.hip has no device code: it is compiled
here with plain g++ under Address + UndefinedBehaviour sanitizers and under ThreadSanitizer and
driven by tests/native/test_hostio.cpp with truncated / corrupt deflate streams, slices at the
chunk edges, unaligned destinations and 1-8 threads.  CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [("-fsanitize=address,undefined", "-fno-sanitize-recover=all"), ("-fsanitize=thread",)],
                         ids=["asan-ubsan", "tsan"])
def test_chunk_pipeline_under_sanitizers(tmp_path, flags):
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run(["g++", *flags, "-o", str(tmp_path / "probe"), str(probe)], capture_output=True).returncode:
        pytest.skip("g++ cannot link %s here" % flags[0])
    exe = str(tmp_path / "test_hostio")
    subprocess.run(["g++", "-x", "c++", "-O1", "-g", "-std=c++17", "-pthread", "-Wall", *flags, "-o", exe,
                    os.path.join(ROOT, "tests", "native", "test_hostio.cpp"), "-lz", "-ldl"], check=True, cwd=ROOT)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67",
               UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:] + r.stderr[-3000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
