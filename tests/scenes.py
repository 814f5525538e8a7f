"""The parity scenes, shared by the GPU tests (HIP path vs oracle, tests/test_gpu_parity.py) and
the CPU tests (literal vs parity-definition oracle, tests/test_oracle_literal.py), and a
renderer-free builder of the four kernel-argument structs for them."""
import numpy as np

from oracle import vro
from tests import common

UCHAR, USHORT, FLOAT = 0, 1, 2
SEED = 3499211612


CASES = [
    # fmt, res, (W, H), view, tff, kwargs
    (UCHAR, (48, 48, 48), (96, 80), "default", "default", {}),
    (UCHAR, (48, 48, 48), (96, 80), "rot30", "default", {"ess": False}),
    (UCHAR, (64, 40, 52), (120, 72), "rot30", "opaque", {"gradient_bg": True}),
    (UCHAR, (33, 47, 29), (64, 64), "close", "haze", {"illum": 0}),
    (UCHAR, (48, 48, 48), (64, 64), "inside", "default", {}),
    (UCHAR, (48, 48, 48), (72, 56), "rot30", "default", {"linear": False}),
    (UCHAR, (48, 48, 48), (72, 56), "rot30", "default", {"ortho": True}),
    (UCHAR, (48, 48, 48), (64, 64), "default", "default",
     {"bbox": (-0.5, -0.8, -1.0, 0.7, 0.6, 0.2)}),
    (UCHAR, (40, 40, 40), (64, 48), "rot30", "haze", {"contours": True, "aerial": True}),
    (UCHAR, (40, 40, 40), (64, 48), "close", "opaque", {"illum": 0, "contours": True}),
    (UCHAR, (32, 32, 80), (64, 64), "rot30", "default", {"thickness": (1.0, 1.0, 2.5)}),
    (UCHAR, (128, 128, 128), (128, 128), "rot30", "default", {"rate": 0.7}),
    (USHORT, (48, 48, 48), (80, 64), "rot30", "default", {}),
    (USHORT, (40, 56, 36), (64, 64), "close", "opaque", {"ess": False}),
    (FLOAT, (48, 48, 48), (80, 64), "rot30", "default", {}),
    (FLOAT, (36, 36, 36), (64, 64), "default", "haze", {"illum": 0}),
    # shading modes 2-5: TF-opacity gradient, Sobel, gradient-magnitude TF, cel
    (UCHAR, (48, 48, 48), (80, 64), "rot30", "default", {"illum": 2}),
    (UCHAR, (40, 44, 36), (64, 64), "rot30", "opaque", {"illum": 3, "contours": True}),
    (UCHAR, (40, 40, 40), (64, 64), "close", "default", {"illum": 4}),
    (USHORT, (40, 40, 40), (64, 64), "rot30", "opaque", {"illum": 4, "ess": False}),
    (FLOAT, (40, 40, 40), (72, 56), "rot30", "default", {"illum": 5}),
    (FLOAT, (32, 32, 32), (64, 48), "default", "opaque", {"illum": 3, "ess": False}),
    (UCHAR, (40, 40, 40), (64, 48), "inside", "default", {"illum": 5, "aerial": True}),
    # ambient occlusion at early ray termination (calcAO, volumeraycast.cl:368-392, :870-876)
    (UCHAR, (48, 48, 48), (80, 64), "rot30", "opaque", {"ao": True}),
    (FLOAT, (40, 40, 40), (64, 64), "close", "opaque", {"ao": True, "ess": False, "illum": 0}),
    (USHORT, (40, 44, 36), (64, 56), "rot30", "opaque", {"ao": True, "illum": 3}),
    # showEss (:888-896): rays without a sample and rays ending next to a box edge are marked
    (UCHAR, (48, 48, 48), (96, 80), "rot30", "default",
     {"show_ess": True, "background": (0.25, 0.5, 1.0, 1.0)}),
    (USHORT, (40, 44, 36), (64, 56), "default", "haze", {"show_ess": True, "ess": False}),
    (FLOAT, (40, 40, 40), (64, 64), "close", "opaque", {"show_ess": True, "illum": 0}),
]


PT_CASES = [
    # fmt, res, (W, H), view, tff, smooth, kwargs -- technique 1 (Woodcock-tracking path tracer)
    (FLOAT, (48, 48, 48), (96, 80), "rot30", "default", True, {}),
    (FLOAT, (48, 48, 48), (64, 64), "default", "haze", False, {"ext": 30.0}),
    (UCHAR, (64, 40, 52), (72, 56), "rot30", "opaque", True, {"ext": 250.0, "gradient_bg": True}),
    (USHORT, (40, 56, 36), (64, 64), "close", "default", True, {"ext": 60.0}),
    (FLOAT, (48, 48, 48), (64, 64), "inside", "default", True, {}),
    (UCHAR, (48, 48, 48), (64, 64), "default", "default", False,
     {"bbox": (-0.5, -0.8, -1.0, 0.7, 0.6, 0.2), "ortho": True}),
]


MC_CASES = [
    # fmt, channels, res, view, kwargs -- CL_RGBA / CL_RG volumes (volumeraycast.cl:838-855)
    (UCHAR, 4, (40, 40, 40), "rot30", {}),
    (FLOAT, 4, (36, 40, 32), "close", {"ess": False, "linear": False}),
    (USHORT, 2, (40, 36, 44), "rot30", {"aerial": True}),
    (UCHAR, 2, (40, 40, 40), "default", {"ess": False, "illum": 0}),
    (UCHAR, 4, (40, 40, 40), "rot30", {"illum": 4}),               # gradient magnitude of .x
    (FLOAT, 4, (32, 32, 32), "inside", {"ao": True, "show_ess": True}),
]


FP_CASES = [
    # fmt, res, (W, H), view, tff, kwargs -- frames the default kernels render from the footprint
    # volume (un-instrumented, volume not much wider than the viewport)
    (UCHAR, (48, 48, 48), (96, 80), "rot30", "default", {}),
    (UCHAR, (33, 47, 29), (64, 64), "close", "haze", {"illum": 0}),
    (UCHAR, (48, 48, 48), (64, 64), "inside", "default", {}),        # edge-clamped fetches
    (USHORT, (40, 56, 36), (64, 64), "close", "opaque", {"ess": False}),
    (USHORT, (45, 45, 45), (80, 64), "rot30", "default", {"contours": True}),
    (FLOAT, (48, 48, 48), (80, 64), "rot30", "default", {}),
    (FLOAT, (37, 37, 37), (64, 64), "inside", "haze", {"ess": False}),
]


def oracle_params(res, view, kw, seed=SEED):
    """(cam, rp, rc, pt) for a scene, built the way VolumeRenderCL builds them
    (volumerendercl.cpp:347-362 calcScaling, :620-631 brickRes, the setters of :922-1047) --
    tests/test_gpu_parity.py checks them byte for byte against the product's own structs."""
    cam = vro.CameraParams()
    cam.viewMat[:] = common.views()[view]
    bb = kw.get("bbox", (-1, -1, -1, 1, 1, 1))
    cam.bbox_bl[:] = [bb[0], bb[1], bb[2], 0]
    cam.bbox_tr[:] = [bb[3], bb[4], bb[5], 0]
    cam.ortho = 1 if kw.get("ortho", False) else 0
    rp = vro.RenderingParams()
    bg = kw.get("background")
    rp.backgroundColor[:] = [bg[0], bg[1], bg[2], 0.0] if bg is not None else [1.0, 1.0, 1.0, 1.0]
    rp.modelScale[:] = vro.calc_scaling(list(res), list(kw.get("thickness", (1.0, 1.0, 1.0)))) + [0]
    rp.illumType = kw.get("illum", 1)
    rp.imgEss = 1 if kw.get("img_ess", False) else 0
    rp.showEss = 1 if kw.get("show_ess", False) else 0
    rp.useLinear = 1 if kw.get("linear", True) else 0
    rp.useGradient = 1 if kw.get("gradient_bg", False) else 0
    rp.technique = kw.get("technique", 0)
    rp.seed = seed
    rp.iteration = 0
    rc = vro.RaycastParams()
    rc.samplingRate = kw.get("rate", 1.5)
    rc.useAO = 1 if kw.get("ao", False) else 0
    rc.contours = 1 if kw.get("contours", False) else 0
    rc.aerial = 1 if kw.get("aerial", False) else 0
    _, brf, _ = vro.brick_layout(list(res))
    rc.brickRes[:] = brf + [0]
    pt = vro.PathtraceParams(kw.get("ext", 100.0))
    return cam, rp, rc, pt


def multichannel_volume(fmt, nch, res):
    planes = [common.noise_volume(res, fmt, seed=20 + c, smooth=False) for c in range(nch)]
    vol = np.stack(planes, axis=-1)
    if nch == 4:   # keep the opacity channel moderate so that rays are not cut at once
        vol[..., 3] = (vol[..., 3] * 0.2).astype(vol.dtype)
    return vol
