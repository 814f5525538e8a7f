"""ctypes binding of the CPU oracle (oracle/vr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvroracle.so")

UCHAR, USHORT, FLOAT = 0, 1, 2
_NP_DTYPE = {UCHAR: np.uint8, USHORT: np.uint16, FLOAT: np.float32}


class CameraParams(C.Structure):
    _fields_ = [("viewMat", C.c_float * 16), ("bbox_bl", C.c_float * 4),
                ("bbox_tr", C.c_float * 4), ("ortho", C.c_uint32), ("_pad", C.c_uint32 * 7)]


class RenderingParams(C.Structure):
    _fields_ = [("backgroundColor", C.c_float * 4), ("modelScale", C.c_float * 4),
                ("illumType", C.c_uint32), ("imgEss", C.c_uint32), ("showEss", C.c_uint32),
                ("useLinear", C.c_uint32), ("useGradient", C.c_uint32),
                ("technique", C.c_uint32), ("seed", C.c_uint32), ("iteration", C.c_uint32)]


class RaycastParams(C.Structure):
    _fields_ = [("samplingRate", C.c_float), ("useAO", C.c_uint32), ("contours", C.c_uint32),
                ("aerial", C.c_uint32), ("brickRes", C.c_float * 4)]


class PathtraceParams(C.Structure):
    _fields_ = [("max_extinction", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("voxels", C.c_void_p), ("res", C.c_uint32 * 3), ("format", C.c_int32),
                ("bricks", C.c_void_p), ("bricks_res", C.c_uint32 * 3),
                ("tff", C.c_void_p), ("tff_n", C.c_uint32),
                ("prefix", C.c_void_p), ("prefix_n", C.c_uint32), ("channels", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("samples_taken", C.c_uint64), ("samples_nominal", C.c_uint64),
                ("samples_shaded", C.c_uint64), ("bricks_visited", C.c_uint64),
                ("bricks_skipped", C.c_uint64), ("rays_hit", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


assert C.sizeof(CameraParams) == 128 and C.sizeof(RenderingParams) == 64
assert C.sizeof(RaycastParams) == 32 and C.sizeof(PathtraceParams) == 4


def build(force=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vr_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(
            os.path.join(_HERE, "_ref", "libref_kernel.so")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.vro_parallel_rng.restype = C.c_uint32
        L.vro_parallel_rng.argtypes = [C.c_uint32]
        L.vro_parallel_rng3.restype = C.c_uint32
        L.vro_parallel_rng3.argtypes = [C.c_uint32] * 3
        L.vro_map_uint_float.restype = C.c_float
        L.vro_map_uint_float.argtypes = [C.c_uint32]
        L.vro_powr.restype = C.c_float
        L.vro_powr.argtypes = [C.c_float, C.c_float]
        L.vro_round_pow2.restype = C.c_uint32
        L.vro_round_pow2.argtypes = [C.c_uint32]
        L.vro_padded.restype = C.c_uint32
        L.vro_padded.argtypes = [C.c_uint32]
        L.vro_intersect_bbox.restype = C.c_int
        L.vro_intersect_bbox.argtypes = [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_float)] * 2
        L.vro_render_tile.restype = C.c_int
        L.vro_render_tile.argtypes = [
            C.POINTER(Scene), C.POINTER(CameraParams), C.POINTER(RenderingParams),
            C.POINTER(RaycastParams), C.POINTER(PathtraceParams), C.c_int,
            C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
            C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_void_p, C.c_int]
        L.vro_render_tile_ex.restype = C.c_int
        L.vro_render_tile_ex.argtypes = L.vro_render_tile.argtypes + [C.POINTER(FrameExtras)]
        L.vro_hit_image_init.restype = None
        L.vro_hit_image_init.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.vro_generate_bricks.restype = C.c_int
        L.vro_generate_bricks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int,
                                          C.POINTER(C.c_uint32), C.c_void_p]
        L.vro_synth_volume.restype = C.c_int
        L.vro_synth_volume.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_void_p]
        L.vro_num_threads.restype = C.c_int
        L.vro_atan2f.restype = C.c_float
        L.vro_atan2f.argtypes = [C.c_float, C.c_float]
        L.vro_acosf.restype = C.c_float
        L.vro_acosf.argtypes = [C.c_float]
        U4 = C.POINTER(C.c_uint32)
        L.vro_ui_rand_step.restype = C.c_uint32
        L.vro_ui_rand_step.argtypes = [U4, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.vro_lcg_step.restype = C.c_uint32
        L.vro_lcg_step.argtypes = [U4, C.c_uint32, C.c_uint32]
        L.vro_hybrid_rand.restype = C.c_float
        L.vro_hybrid_rand.argtypes = [U4]
        L.vro_check_bounding_box.restype = C.c_int
        L.vro_check_bounding_box.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]
        _lib = L
    return _lib


def _u3(v):
    return (C.c_uint32 * 3)(*[int(x) for x in v])


def brick_layout(res):
    edge = (C.c_uint32 * 3)()
    brf = (C.c_float * 3)()
    tex = (C.c_uint32 * 3)()
    lib().vro_brick_layout(_u3(res), edge, brf, tex)
    return list(edge), list(brf), list(tex)


def calc_scaling(res, thickness):
    ms = (C.c_float * 3)()
    lib().vro_calc_scaling(_u3(res), (C.c_double * 3)(*thickness), ms)
    return list(ms)


def prefix_sum(tff):
    tff = np.ascontiguousarray(tff, dtype=np.uint8).reshape(-1)
    n = tff.size // 4
    out = np.zeros(n, dtype=np.uint32)
    lib().vro_prefix_sum(tff.ctypes.data_as(C.c_void_p), C.c_uint32(n),
                         out.ctypes.data_as(C.c_void_p))
    return out


def generate_bricks(vol, fmt):
    """vol: ndarray [z, y, x] of the format's dtype. Returns ndarray [bz, by, bx, 2]."""
    vol = np.ascontiguousarray(vol, dtype=_NP_DTYPE[fmt])
    res = [vol.shape[2], vol.shape[1], vol.shape[0]]
    _, _, tex = brick_layout(res)
    out = np.zeros((tex[2], tex[1], tex[0], 2), dtype=_NP_DTYPE[fmt])
    rc = lib().vro_generate_bricks(vol.ctypes.data_as(C.c_void_p), _u3(res), fmt, _u3(tex),
                                   out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def downsample(vol, fmt, factor):
    """volumeDownsampling's kernel + size rule. vol: ndarray [z, y, x]; returns the low-res ndarray."""
    vol = np.ascontiguousarray(vol, dtype=_NP_DTYPE[fmt])
    res = [vol.shape[2], vol.shape[1], vol.shape[0]]
    lo = [-(-r // factor) for r in res]
    out = np.zeros((lo[2], lo[1], lo[0]), dtype=_NP_DTYPE[fmt])
    out_res = (C.c_uint32 * 3)()
    f = lib().vro_downsample
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32)]
    rc = f(vol.ctypes.data_as(C.c_void_p), _u3(res), fmt, int(factor), out.ctypes.data_as(C.c_void_p), out_res)
    assert rc == 0 and list(out_res) == lo
    return out


def synth_volume(kind, res, fmt):
    """kind: 'sphere' | 'shells'. Returns ndarray [z, y, x]."""
    out = np.zeros((res[2], res[1], res[0]), dtype=_NP_DTYPE[fmt])
    rc = lib().vro_synth_volume({"sphere": 0, "shells": 1}[kind], _u3(res), fmt,
                                out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


class FrameExtras(C.Structure):
    _fields_ = [("hit_in", C.c_void_p), ("hit_out", C.c_void_p), ("env_rgba", C.c_void_p),
                ("env_w", C.c_uint32), ("env_h", C.c_uint32)]


def hit_image_init(W, H):
    """(hit_in, hit_out) as updateOutputImg leaves them: uint8 [(H/8+1), (W/8+1)] each."""
    shape = (H // 8 + 1, W // 8 + 1)
    a, b = np.zeros(shape, np.uint8), np.zeros(shape, np.uint8)
    lib().vro_hit_image_init(W, H, a.ctypes.data, b.ctypes.data)
    return a, b


def render_tile(vol, fmt, tff, cam, rp, rc, pt=None, use_ess=True, W=64, H=64, tile=None,
                in_accum=None, bricks=None, prefix=None, want_touched=False, threads=0,
                hit_in=None, hit_out=None, env=None, literal=False):
    """Render the tile (x0, y0, w, h) of a W x H frame with the oracle.

    hit_in / hit_out: image-order ESS state (uint8 [(H/8+1), (W/8+1)], hit_out is updated in
    place); env: environment map, float32 [h, w, 4].
    literal: evaluate the reference's source text literally (vro_set_literal) instead of the parity
    definitions shared with the HIP kernel -- CPU tests bound the difference.
    Returns (rgba float32 [h, w, 4], stats dict, touched-bitmap or None).
    """
    vol = np.ascontiguousarray(vol, dtype=_NP_DTYPE[fmt])
    res = [vol.shape[2], vol.shape[1], vol.shape[0]]
    channels = vol.shape[3] if vol.ndim == 4 else 1    # [z, y, x, c]: CL_RG / CL_RGBA volumes
    tff = np.ascontiguousarray(tff, dtype=np.uint8).reshape(-1)
    if prefix is None:
        prefix = prefix_sum(tff)
    prefix = np.ascontiguousarray(prefix, dtype=np.uint32)
    if bricks is None and use_ess:
        # generateBricks reads the .x component (volumeraycast.cl:947-957)
        bricks = generate_bricks(vol[..., 0] if vol.ndim == 4 else vol, fmt)
    x0, y0, w, h = tile if tile is not None else (0, 0, W, H)
    sc = Scene()
    sc.voxels = vol.ctypes.data
    sc.res = _u3(res)
    sc.format = fmt
    sc.channels = channels
    if bricks is not None:
        bricks = np.ascontiguousarray(bricks, dtype=_NP_DTYPE[fmt])
        sc.bricks = bricks.ctypes.data
        sc.bricks_res = _u3([bricks.shape[2], bricks.shape[1], bricks.shape[0]])
    sc.tff = tff.ctypes.data
    sc.tff_n = tff.size // 4
    sc.prefix = prefix.ctypes.data
    sc.prefix_n = prefix.size
    out = np.zeros((h, w, 4), dtype=np.float32)
    st = Stats()
    touched = None
    if want_touched:
        mb = [(r + 3) // 4 for r in res]
        touched = np.zeros((mb[0] * mb[1] * mb[2] + 7) // 8, dtype=np.uint8)
    acc = None
    if in_accum is not None:
        acc = np.ascontiguousarray(in_accum, dtype=np.float32)
    pt = pt if pt is not None else PathtraceParams(100.0)
    ex = FrameExtras()
    if hit_in is not None:
        assert hit_in.dtype == np.uint8 and hit_out.dtype == np.uint8
        assert hit_in.flags.c_contiguous and hit_out.flags.c_contiguous
        assert hit_in.shape == (H // 8 + 1, W // 8 + 1) == hit_out.shape
        ex.hit_in, ex.hit_out = hit_in.ctypes.data, hit_out.ctypes.data
    if env is not None:
        env = np.ascontiguousarray(env, dtype=np.float32)
        ex.env_rgba, ex.env_w, ex.env_h = env.ctypes.data, env.shape[1], env.shape[0]
    lib().vro_set_literal(1 if literal else 0)
    r = lib().vro_render_tile_ex(C.byref(sc), C.byref(cam), C.byref(rp), C.byref(rc), C.byref(pt),
                                 1 if use_ess else 0, W, H, x0, y0, w, h,
                                 acc.ctypes.data if acc is not None else None,
                                 out.ctypes.data, C.byref(st),
                                 touched.ctypes.data if touched is not None else None, threads,
                                 C.byref(ex))
    lib().vro_set_literal(0)
    if r != 0:
        raise RuntimeError("vro_render_tile failed: %d" % r)
    return out, st.as_dict(), touched
