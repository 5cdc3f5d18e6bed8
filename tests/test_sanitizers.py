"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the code that can run without a GPU: the product's host side
(jefferson-2.0_amd/csrc/jf_host.cpp -- geometry, index/weight rules, WAV I/O incl. truncated and corrupted files, the KEMAR
directory loader on a complete synthetic set and on broken ones, the reverb's schedule and gain) and the C oracle (every
entry point, ragged and empty signals, positions outside the range, the reverb stage).  GPU sanitizers are not available on
the pool; the kernels' bounds are the parity tests' and the host-side shape checks' business."""
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")


def _have_sanitizers(cc):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write("int main(void){return 0;}\n")
        return subprocess.run([cc, "-x", "c", "-fsanitize=address,undefined", src, "-o", os.path.join(d, "t")],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 0


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, **kw)
    text = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "runtime error" not in text and "AddressSanitizer" not in text and "LeakSanitizer" not in text, text[-3000:]
    return text


@pytest.mark.skipif(not (shutil.which("g++") and _have_sanitizers("gcc")), reason="gcc with libasan/libubsan not available")
def test_host_side_under_asan_and_ubsan():
    build = os.path.join(ROOT, "tests", "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "host_san")
    _run(["g++", "-std=c++17", *SAN, "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
          os.path.join(ROOT, "tests", "san", "host_san_driver.cpp"), os.path.join(ROOT, "jefferson-2.0_amd", "csrc", "jf_host.cpp"),
          "-o", exe])
    with tempfile.TemporaryDirectory() as scratch:
        out = _run([exe, scratch], env=ENV)
    assert "0 failed checks" in out


@pytest.mark.skipif(not _have_sanitizers("gcc"), reason="gcc with libasan/libubsan not available")
def test_oracle_under_asan_and_ubsan():
    build = os.path.join(ROOT, "tests", "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "oracle_san")
    _run(["gcc", "-std=c11", *SAN, "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "san", "oracle_san_driver.c"),
          os.path.join(ROOT, "oracle", "jf_oracle.c"), "-lm", "-o", exe])
    out = _run([exe], env=dict(ENV, OMP_NUM_THREADS="2"))
    assert "0 bad values" in out


@pytest.mark.skipif(not _have_sanitizers("gcc"), reason="gcc with libasan/libubsan not available")
def test_hdf5_reader_under_asan_and_ubsan():
    """jf_hdf5.c over the five h5py-written containers, 400 damaged copies of each and crafted B-tree headers: no fault, no leak, no undefined step"""
    build = os.path.join(ROOT, "tests", "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "hdf5_san")
    csrc = os.path.join(ROOT, "jefferson-2.0_amd", "csrc")
    _run(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", *SAN, "-I" + csrc, os.path.join(ROOT, "tests", "san", "hdf5_san_driver.c"),
          os.path.join(csrc, "jf_hdf5.c"), "-o", exe, "-lz"])
    sofa = os.path.join(ROOT, "tests", "golden", "sofa")
    with tempfile.TemporaryDirectory() as scratch:
        out = _run([exe, scratch, "400"] + [os.path.join(sofa, n + ".sofa") for n in ("nc4", "symtab", "latest", "cartesian", "mono")],
                   env=ENV)
    assert "dataset reads succeeded" in out
    # the crafted version-2 B-trees (record size below what the record type's callback reads) were run
    assert "directed B-tree cases" in out and not out.startswith("0 directed")
