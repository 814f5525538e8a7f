"""Shared helpers for the parity tests: scene construction and oracle<->product glue.

The oracle (oracle/vro.py) is the checker only; the product is driven through the C ABI
via volumerenderercl_amd.VolumeRenderCL.
"""
import numpy as np

from oracle import vro
from volumerenderercl_amd import frontend

NP_DTYPE = {0: np.uint8, 1: np.uint16, 2: np.float32}


def to_oracle_params(cam, rp, rc, pt):
    """Byte-copy the product's kernel-argument structs into the oracle's own types."""
    return (vro.CameraParams.from_buffer_copy(bytes(cam)),
            vro.RenderingParams.from_buffer_copy(bytes(rp)),
            vro.RaycastParams.from_buffer_copy(bytes(rc)),
            vro.PathtraceParams.from_buffer_copy(bytes(pt)))


def noise_volume(res, fmt, seed=0, smooth=True):
    """Deterministic pseudo-random field [z, y, x] with structure at several scales."""
    rng = np.random.default_rng(seed)
    x, y, z = res
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, z), np.linspace(-1, 1, y), np.linspace(-1, 1, x),
                             indexing="ij")
    r = np.sqrt(xx * xx + yy * yy + zz * zz)
    f = np.clip(1.0 - r / 0.95, 0, 1) * (0.6 + 0.4 * np.cos(9 * xx) * np.cos(7 * yy + 1) *
                                          np.cos(5 * zz + 2))
    if not smooth:
        f = f * (0.7 + 0.3 * rng.random(f.shape))
    f = np.clip(f, 0, 1)
    if fmt == 0:
        return np.round(f * 255).astype(np.uint8)
    if fmt == 1:
        return np.round(f * 65535).astype(np.uint16)
    return f.astype(np.float32)


def views():
    q30 = frontend.quat_from_axis_angle((1, 1, 0), 30.0)
    return {
        "default": frontend.view_matrix(),
        "rot30": frontend.view_matrix(q30, (0.0, 0.0, 2.0)),
        "close": frontend.view_matrix(frontend.quat_from_axis_angle((0.2, 1, 0.1), 75.0),
                                      (0.1, -0.05, 1.2)),
        "inside": frontend.view_matrix(frontend.quat_from_axis_angle((0, 1, 0), 20.0),
                                       (0.0, 0.0, 0.4)),
    }


def tffs():
    return {
        "default": frontend.tff_from_stops(),
        "opaque": frontend.opaque_ramp_tff(),
        "haze": frontend.haze_tff(),
    }


def oracle_frame(vr, vol, fmt, tff, W, H, use_ess=True, tile=None, in_accum=None,
                 want_touched=False, prefix=None, env=None):
    cam, rp, rc, pt = to_oracle_params(*vr.params())
    return vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, use_ess=use_ess, W=W, H=H, tile=tile,
                           in_accum=in_accum, want_touched=want_touched, prefix=prefix, env=env)
