"""The library used from a plain-C host program (no Python, no torch): the
boundary is a C ABI, so a cgo / JNI / C++ caller sees exactly this."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "host_check.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "host_check")


def _build():
    stale = (not os.path.exists(EXE)) or os.path.getmtime(EXE) < os.path.getmtime(SRC)
    if stale:
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                               "-L", os.path.join(ROOT, "rag_amd"), "-lrag_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "rag_amd"), "-lm"])


def test_c_host_program_compiles_against_the_header():
    """CPU: the header is valid C11 and every symbol the program uses links."""
    _build()
    assert os.access(EXE, os.X_OK)


@pytest.mark.gpu
def test_c_host_program_runs():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "ok" in out.stdout
