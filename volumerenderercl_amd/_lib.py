"""ctypes binding of libvrhip.so (include/vrhip.h).

The HIP library is the product: there is no CPU fallback.  Loading fails loudly when the
in-tree shared library has not been built (`python -c "import __graft_entry__ as g; g.build()"`
or `make -C volumerenderercl_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VRHIP_LIB_PATH selects an alternative build of the same library (A/B kernel experiments)
LIB_PATH = os.environ.get("VRHIP_LIB_PATH") or os.path.join(_HERE, "libvrhip.so")

OK, ERR_INVALID, ERR_HIP, ERR_NODATA, ERR_UNSUPPORTED = range(5)
UCHAR, USHORT, FLOAT = 0, 1, 2


class CameraParams(C.Structure):
    """vrhip_camera_params == VolumeRenderCL::camera_params (volumerendercl.h:43-49)."""
    _fields_ = [("viewMat", C.c_float * 16), ("bbox_bl", C.c_float * 4),
                ("bbox_tr", C.c_float * 4), ("ortho", C.c_uint32), ("_pad", C.c_uint32 * 7)]


class RenderingParams(C.Structure):
    """vrhip_rendering_params == VolumeRenderCL::rendering_params (volumerendercl.h:51-66)."""
    _fields_ = [("backgroundColor", C.c_float * 4), ("modelScale", C.c_float * 4),
                ("illumType", C.c_uint32), ("imgEss", C.c_uint32), ("showEss", C.c_uint32),
                ("useLinear", C.c_uint32), ("useGradient", C.c_uint32),
                ("technique", C.c_uint32), ("seed", C.c_uint32), ("iteration", C.c_uint32)]


class RaycastParams(C.Structure):
    """vrhip_raycast_params == VolumeRenderCL::raycast_params (volumerendercl.h:68-76)."""
    _fields_ = [("samplingRate", C.c_float), ("useAO", C.c_uint32), ("contours", C.c_uint32),
                ("aerial", C.c_uint32), ("brickRes", C.c_float * 4)]


class PathtraceParams(C.Structure):
    """vrhip_pathtrace_params == VolumeRenderCL::pathtrace_params (volumerendercl.h:78-81)."""
    _fields_ = [("max_extinction", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples_taken", C.c_uint64), ("samples_nominal", C.c_uint64),
                ("samples_shaded", C.c_uint64), ("bricks_visited", C.c_uint64),
                ("bricks_skipped", C.c_uint64), ("rays_hit", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class LaunchInfo(C.Structure):
    """vrhip_launch_info: what the last render call launched."""
    _fields_ = [(n, C.c_uint32) for n in (
        "technique", "frames", "work_items", "prepass", "ray_list", "phase1_waves", "phase2_waves", "round_budget",
        "footprint", "empty_skip", "skip_in_lds", "instrumented", "extras", "patch_classes", "sorted_phase2")] + [("reserved", C.c_uint32 * 17)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


assert C.sizeof(LaunchInfo) == 128
assert C.sizeof(CameraParams) == 128 and C.sizeof(RenderingParams) == 64
assert C.sizeof(RaycastParams) == 32 and C.sizeof(PathtraceParams) == 4

_H = C.c_void_p
_U3 = C.POINTER(C.c_uint32)

# name -> (restype, argtypes): every symbol include/vrhip.h declares
SYMBOLS = {
    "vrhip_abi_version": (C.c_int, []),
    "vrhip_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "vrhip_destroy": (None, [_H]),
    "vrhip_last_error": (C.c_char_p, [_H]),
    "vrhip_device_name": (C.c_int, [_H, C.c_char_p, C.c_size_t]),
    "vrhip_set_stream": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "vrhip_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "vrhip_download_cells": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vrhip_download_empty_cells": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32)]),
    "vrhip_assemble_batch": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p]),
    "vrhip_pack_tiles": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vrhip_message_positions": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32,
                                          C.c_uint32, C.c_void_p]),
    "vrhip_assemble_frame": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p]),
    "vrhip_upload_volume": (C.c_int, [_H, C.c_void_p, _U3, C.c_int, C.c_uint32]),
    "vrhip_upload_volume_channels": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_uint32), C.c_int,
                                               C.c_int, C.c_uint32]),
    "vrhip_share_volumes": (C.c_int, [_H, _H]),
    "vrhip_render_batch": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "vrhip_set_round_budget": (C.c_int, [_H, C.c_uint32]),
    "vrhip_upload_volume_device": (C.c_int, [_H, C.c_void_p, _U3, C.c_int, C.c_uint32]),
    "vrhip_synth_volume": (C.c_int, [_H, C.c_int, _U3, C.c_int, C.c_uint32]),
    "vrhip_download_volume": (C.c_int, [_H, C.c_uint32, C.c_void_p, C.c_size_t]),
    "vrhip_downsample_volume": (C.c_int, [_H, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, _U3]),
    "vrhip_clear_volumes": (C.c_int, [_H]),
    "vrhip_set_timestep": (C.c_int, [_H, C.c_uint32]),
    "vrhip_get_resolution": (C.c_int, [_H, C.POINTER(C.c_uint32)]),
    "vrhip_set_transfer_function": (C.c_int, [_H, C.c_void_p, C.c_uint32]),
    "vrhip_set_tff_prefix_sum": (C.c_int, [_H, C.c_void_p, C.c_uint32]),
    "vrhip_build_bricks": (C.c_int, [_H]),
    "vrhip_get_brick_info": (C.c_int, [_H, _U3, C.POINTER(C.c_float), _U3]),
    "vrhip_download_bricks": (C.c_int, [_H, C.c_uint32, C.c_void_p, C.c_size_t]),
    "vrhip_last_bricks_seconds": (C.c_double, [_H]),
    "vrhip_set_camera_params": (C.c_int, [_H, C.POINTER(CameraParams)]),
    "vrhip_set_rendering_params": (C.c_int, [_H, C.POINTER(RenderingParams)]),
    "vrhip_set_raycast_params": (C.c_int, [_H, C.POINTER(RaycastParams)]),
    "vrhip_set_pathtrace_params": (C.c_int, [_H, C.POINTER(PathtraceParams)]),
    "vrhip_set_object_ess": (C.c_int, [_H, C.c_int]),
    "vrhip_render_frame": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]),
    "vrhip_render_tiles": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_uint32, C.c_void_p]),
    "vrhip_set_environment_map": (C.c_int, [_H, C.c_void_p, C.c_uint32, C.c_uint32]),
    "vrhip_reset_image_ess": (C.c_int, [_H]),
    "vrhip_get_image_ess": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vrhip_set_image_ess": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vrhip_last_kernel_seconds": (C.c_double, [_H]),
    "vrhip_last_phase_seconds": (C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vrhip_last_launch_info": (C.c_int, [_H, C.POINTER(LaunchInfo)]),
    "vrhip_download_cost_map": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "vrhip_set_phase_timing": (C.c_int, [_H, C.c_int]),
    "vrhip_set_frame_timing": (C.c_int, [_H, C.c_int]),
    "vrhip_build_source_hash": (C.c_char_p, []),
    "vrhip_set_stats_enabled": (C.c_int, [_H, C.c_int]),
    "vrhip_get_stats": (C.c_int, [_H, C.POINTER(Stats)]),
    "vrhip_count_touched": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                      C.c_void_p, C.c_size_t]),
    "vrhip_count_fetched": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "vrhip_count_touched_tiles": (C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
}

_lib = None


def load():
    """Load libvrhip.so and bind every declared symbol.  Raises (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvrhip.so is not built (%s missing). Build it with "
            "`make -C volumerenderercl_amd/csrc` or __graft_entry__.build(); there is no "
            "CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.vrhip_abi_version() != 1:
        raise RuntimeError("libvrhip.so ABI version mismatch")
    _lib = lib
    return lib
