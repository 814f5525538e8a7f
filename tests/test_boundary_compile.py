"""The drop-in boundary is usable from include/ alone (INTEGRATION.md route A): a caller that
holds `VolumeRenderCL` by value and walks the reference's call order compiles with nothing but
-I include, and links against libvrhost + libvrhip."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "caller_by_value.cpp")
PKG = os.path.join(ROOT, "volumerenderercl_amd")


def test_caller_compiles_against_include_alone():
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), SRC])


def test_every_public_header_is_self_contained():
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        cmd = ["g++", "-std=c++17"] if h in ("volumerendercl.h", "datrawreader.h") else ["gcc", "-std=c11"]
        subprocess.check_call(cmd + ["-fsyntax-only", "-x", "c++" if cmd[0] == "g++" else "c",
                                     "-I", inc, os.path.join(inc, h)])


def _build(tmp_path):
    exe = str(tmp_path / "caller_by_value")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", PKG, "-lvrhost", "-lvrhip", "-Wl,-rpath," + PKG])
    return exe


def test_caller_links(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_caller_runs(tmp_path):
    """The by-value caller renders a .dat/.raw volume through the C++ class on the GPU."""
    import numpy as np
    vol = (np.random.default_rng(3).random((32, 32, 32)) * 255).astype(np.uint8)
    vol.tofile(tmp_path / "v.raw")
    (tmp_path / "v.dat").write_text("ObjectFileName: v.raw\nResolution: 32 32 32\nSliceThickness: 1 1 1\n"
                                    "Format: UCHAR\n")
    out = subprocess.run([_build(tmp_path), str(tmp_path / "v.dat")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "1 timesteps, 32x32x32, 12288 floats" in out.stdout
