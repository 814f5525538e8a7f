"""The C-ABI library loads on a GPU-less host and exports every symbol include/vrhip.h
declares (no compute calls here); the product fails loudly when it cannot run."""
import ctypes as C
import os
import re

import pytest

from volumerenderercl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vrhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vrhip_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.vrhip_abi_version() == 1


def test_struct_layouts_match_reference_kernel_args():
    """128 / 64 / 32 / 4 bytes, field offsets of volumeraycast.cl:540-582."""
    assert C.sizeof(_lib.CameraParams) == 128
    assert _lib.CameraParams.bbox_bl.offset == 64 and _lib.CameraParams.bbox_tr.offset == 80
    assert _lib.CameraParams.ortho.offset == 96
    assert C.sizeof(_lib.RenderingParams) == 64
    for name, off in (("modelScale", 16), ("illumType", 32), ("imgEss", 36), ("showEss", 40),
                      ("useLinear", 44), ("useGradient", 48), ("technique", 52), ("seed", 56),
                      ("iteration", 60)):
        assert getattr(_lib.RenderingParams, name).offset == off
    assert C.sizeof(_lib.RaycastParams) == 32 and _lib.RaycastParams.brickRes.offset == 16
    assert C.sizeof(_lib.PathtraceParams) == 4


def test_no_cpu_fallback():
    """Without a GPU vrhip_create must fail with a message, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from volumerenderercl_amd import VolumeRenderCL
    vr = VolumeRenderCL()
    with pytest.raises(RuntimeError):
        vr.initialize()
    with pytest.raises(RuntimeError):
        VolumeRenderCL().initialize(useCPU=True)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: no product source may reference oracle/."""
    pkg = os.path.join(ROOT, "volumerenderercl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vr_oracle" not in text and "from oracle" not in text and \
                    "import oracle" not in text and "libvroracle" not in text, os.path.join(dirpath, f)


def test_host_library_exports_every_declared_symbol():
    """libvrhost.so (loader, .hdr decoder) exports what include/vrhost.h declares."""
    text = open(os.path.join(ROOT, "include", "vrhost.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(vr(?:dr|host)_[a-z_0-9]+)\s*\(", text)))
    assert "vrhost_load_hdr" in names and "vrdr_load" in names
    _lib.load()   # libvrhost links against libvrhip
    host = C.CDLL(os.path.join(ROOT, "volumerenderercl_amd", "libvrhost.so"))
    for name in names:
        assert hasattr(host, name), name


def test_library_carries_the_hash_of_its_sources():
    """vrhip_build_source_hash(): what bench.py checks before it measures (a stale build is refused)."""
    from volumerenderercl_amd import _lib, _srchash
    lib = _lib.load()
    built = lib.vrhip_build_source_hash().decode()
    assert len(built) == 16 and int(built, 16) >= 0
    assert built == _srchash.source_hash(), "libvrhip.so is older than its sources: run __graft_entry__.build()"
