"""CPU: the host-side components under ThreadSanitizer and AddressSanitizer / UBSan.

tests/host_tsan/host_tsan_main.cpp drives gsalign_amd/csrc/host (thread pool, exact_sort, index builder and loaders, MAF / VCF emitters, ordered writer) from several threads at once
the way GSAlign_hip's main() does -- results handed over by two worker threads out of order while a formatter thread writes -- with made-up alignment results, so no GPU is needed.
(Round 5's advisor found a late-waking-worker race in HostPool by reading; this keeps the next one from needing a reader.)"""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gsalign_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "host_tsan", "host_tsan_main.cpp")] + [os.path.join(CSRC, "host", f) for f in ("index_io.cpp", "emit.cpp", "host_api.cpp", "synth.cpp")]


def _build(tmp, name, flags):
    exe = os.path.join(tmp, name)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"), "-o", exe] + SOURCES + ["-lz", "-lpthread"]
    return exe, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_host_components_under_sanitizers():
    with tempfile.TemporaryDirectory(prefix="gsa_host_san_") as tmp:
        probe = os.path.join(tmp, "probe.cpp")
        with open(probe, "w") as f:
            f.write("int main(){return 0;}\n")
        for flag in ("-fsanitize=thread", "-fsanitize=address,undefined"):
            if subprocess.run(["g++", flag, probe, "-o", os.path.join(tmp, "probe")], capture_output=True).returncode != 0:
                pytest.skip(f"this toolchain has no runtime for {flag}")
        builds = [_build(tmp, "host_tsan", ["-fsanitize=thread"]), _build(tmp, "host_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])]
        for exe, p in builds:
            out, _ = p.communicate(timeout=600)
            assert p.returncode == 0, out[-3000:]
        # positive control: the same ThreadSanitizer build does report an unsynchronised counter
        r = subprocess.run([builds[0][0], tmp, "1000", "race"], capture_output=True, text=True, timeout=120, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr, (r.returncode, r.stderr[-1000:])
        for exe, _ in builds:
            work = os.path.join(tmp, os.path.basename(exe) + "_w"); os.makedirs(work)
            env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0", GSA_HOST_THREADS="6")
            r = subprocess.run([exe, work, "300000"], capture_output=True, text=True, timeout=900, env=env)
            assert r.returncode == 0 and "HOST_TSAN_OK" in r.stdout and "Sanitizer" not in r.stderr, (os.path.basename(exe), r.returncode, r.stdout[-500:], r.stderr[-4000:])
