"""The drop-in boundary from plain C: include/wvn_hip.h is valid C99 (and C++17), and examples/c_host.c -- a C host with no
HIP header and no Python -- links against libwvn_hip.so and gets SimpleMLP.forward right.  The CPU half checks that the
library links and answers its size queries; the GPU half runs the forward against the C reference inside the example."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "wild_visual_navigation_amd", "lib")
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")


def _build(tmp_path):
    exe = str(tmp_path / "c_host")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host.c"), "-o", exe, "-L", LIBDIR, "-lwvn_hip", "-L", ROCM_LIB, "-lamdhip64", "-lm",
           f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROCM_LIB}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_and_cxx17(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "wvn_hip.h"\nint probe(void) { return (int)sizeof(wvn_vit_model) + (int)sizeof(wvn_mlp_desc); }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "b.o")]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_c_host_links_and_answers(tmp_path):
    from wild_visual_navigation_amd import _lib

    _lib.lib()   # built (raises with the build hint otherwise)
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "parameters 34523" in r.stdout and ("no GPU: the C-ABI links and answers" in r.stdout or "\nok" in r.stdout), r.stdout


@pytest.mark.gpu
def test_c_host_forward_matches_its_c_reference(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), r.stdout + r.stderr
