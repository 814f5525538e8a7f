"""Front-end formulas that feed the hot path: the 16-float view
matrix and the 1024-entry RGBA8 transfer-function table the reference's Qt widget hands to
`VolumeRenderCL::updateView` / `setTransferFunction`.

Restated from Qt 5 semantics (QMatrix4x4 / QPropertyAnimation); Qt is not available here,
so these two are PARITY UNPINNED -- the pinned contract of the hot path is its numeric
inputs (16 floats, 4096 bytes), which every test feeds identically to oracle and GPU.
"""
import math

import numpy as np

# resetCam, volumerenderwidget.cpp:91-92,1069-1074
DEFAULT_ROTATION = (1.0, 0.0, 0.0, 0.0)     # quaternion w, x, y, z
DEFAULT_TRANSLATION = (0.0, 0.0, 2.0)

# TransferFunctionWidget::resetTransferFunction, transferfunctionwidget.cpp:338-346
DEFAULT_STOPS = [(0.00, (0, 0, 0, 0)), (0.10, (125, 125, 125, 0)), (1.00, (0, 0, 0, 255))]

# first outputs of a default-seeded std::mt19937 (reference frame seeds)
MT19937_FIRST_SEEDS = (3499211612, 581869302, 3890346734)


class Mt19937:
    """std::mt19937 (32-bit Mersenne twister, init_genrand seeding); default seed 5489.
    The reference draws one output per frame as the jitter seed (volumerendercl.cpp:212)."""

    def __init__(self, seed=5489):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self):
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def quat_from_axis_angle(axis, degrees):
    """QQuaternion::fromAxisAndAngle (axis normalised)."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = math.radians(degrees) / 2.0
    s = math.sin(h)
    return (math.cos(h), a[0] * s, a[1] * s, a[2] * s)


def quat_mul(p, q):
    pw, px, py, pz = p
    qw, qx, qy, qz = q
    return (pw * qw - px * qx - py * qy - pz * qz,
            pw * qx + px * qw + py * qz - pz * qy,
            pw * qy - px * qz + py * qw + pz * qx,
            pw * qz + px * qy - py * qx + pz * qw)


def rotation_matrix(q):
    w, x, y, z = q
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def view_matrix(rotation=DEFAULT_ROTATION, translation=DEFAULT_TRANSLATION):
    """updateViewMatrix, volumerenderwidget.cpp:1079-1098: M = R(q) * T(t) * S(t.z), handed
    over row-major.  Upper 3x3 = t.z * R, last column = R * t."""
    R = rotation_matrix(rotation)
    t = np.asarray(translation, dtype=np.float64)
    M = np.eye(4)
    M[:3, :3] = R * t[2]
    M[:3, 3] = R @ t
    return [float(np.float32(v)) for v in M.reshape(-1)]


def _ease(kind, t):
    """QEasingCurve::valueForProgress for the curves the GUI offers (mainwindow.cpp:945-957), with the
    operation order of Qt's src/3rdparty/easing/easing.cpp (easeNone, easeInOutQuad, easeInOutCubic)."""
    t = min(1.0, max(0.0, t))
    if kind == "linear":
        return t
    if kind == "quad":     # QEasingCurve::InOutQuad
        t *= 2.0
        if t < 1:
            return t * t / 2.0
        t -= 1
        return -0.5 * (t * (t - 2) - 1)
    if kind == "cubic":    # QEasingCurve::InOutCubic
        t *= 2.0
        if t < 1:
            return 0.5 * t * t * t
        t -= 2.0
        return 0.5 * (t * t * t + 2)
    raise ValueError(kind)


def tff_from_stops(stops=DEFAULT_STOPS, n=1024, easing="linear"):
    """updateTransferFunction, volumerenderwidget.cpp:916-938: entry i samples the key-value
    animation at time qRound(i/n*8192) of 8192; QVariantAnimation interpolates QColor per channel as
    qBound(0, int(f + (t - f) * localProgress), 255) (_q_interpolate<int> truncates) with
    localProgress = (progress - start) / (end - start) in double; then max(0, c - 3).
    setKeyValueAt replaces an earlier key at the same position; missing end stops (the GUI always
    has them) repeat the nearest stop's colour.  Known answers derived by hand from these semantics:
    tests/test_frontend_files.py."""
    dedup = []
    for s in sorted(stops, key=lambda s: s[0]):     # (stable)
        if dedup and dedup[-1][0] == s[0]:
            dedup[-1] = s
        else:
            dedup.append(s)
    stops = dedup
    if stops[0][0] > 0.0:
        stops = [(0.0, stops[0][1])] + stops
    if stops[-1][0] < 1.0:
        stops = stops + [(1.0, stops[-1][1])]
    out = np.zeros((n, 4), dtype=np.uint8)
    for i in range(n):
        time = int(math.floor(i / n * 8192.0 + 0.5))
        p = _ease(easing, time / 8192.0)
        k = 0
        while k + 2 < len(stops) and p >= stops[k + 1][0]:
            k += 1
        (p0, c0), (p1, c1) = stops[k], stops[k + 1]
        lp = 0.0 if p1 == p0 else (p - p0) / (p1 - p0)
        for c in range(4):
            v = int(c0[c] + (c1[c] - c0[c]) * lp)
            v = min(255, max(0, v))
            out[i, c] = max(0, v - 3)
    return out


def opaque_ramp_tff(n=1024):
    """ERT stress table of SURVEY 8(d): alpha = clamp(4*(x - 0.25)), grey ramp colour."""
    x = (np.arange(n) + 0.5) / n
    out = np.zeros((n, 4), dtype=np.uint8)
    out[:, 3] = np.clip(np.round(255 * np.clip(4 * (x - 0.25), 0, 1)), 0, 255)
    out[:, 0] = np.round(255 * x)
    out[:, 1] = np.round(255 * (1 - x))
    out[:, 2] = 128
    return out


def haze_tff(n=1024, alpha=12):
    """Semi-transparent table (no ERT, every sample shaded or not by `alpha`): keeps rays
    alive through the whole volume -- the dense-sampling regime of SURVEY 7."""
    x = (np.arange(n) + 0.5) / n
    out = np.zeros((n, 4), dtype=np.uint8)
    out[:, 0] = np.round(255 * x)
    out[:, 1] = np.round(200 * (1 - x))
    out[:, 2] = 90
    out[:, 3] = np.where(x > 0.05, alpha, 0)
    return out


def prefix_sum(tff):
    """volumerendercl.cpp:879-884."""
    tff = np.asarray(tff, dtype=np.uint8).reshape(-1, 4)
    return np.cumsum(tff[:, 3].astype(np.uint64)).astype(np.uint32)


# ---- the reference GUI's saved files: camera state, transfer functions ----------

def read_tff_stops(path):
    """`.tff` gradient-stop file (MainWindow::readTff / saveTff, mainwindow.cpp:583-657): one
    stop per line, `position r g b a`; lines with fewer than 5 fields are skipped."""
    stops = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 5:
                continue
            stops.append((float(p[0]), tuple(int(float(v)) for v in p[1:5])))
    if not stops:
        raise ValueError("Empty transfer function file.")
    return stops


def write_tff_stops(path, stops):
    with open(path, "w") as f:
        for pos, c in stops:
            f.write("%s %d %d %d %d\n" % (repr(float(pos)), c[0], c[1], c[2], c[3]))


def read_raw_tff(path):
    """Raw transfer function text file (MainWindow::loadRawTff / saveRawTff, mainwindow.cpp:
    662-727): whitespace-separated numbers, each cast to unsigned char; RGBA8 table."""
    import numpy as np
    with open(path) as f:
        vals = [float(v) for v in f.read().split()]
    tff = np.array([int(v) & 0xFF for v in vals], dtype=np.uint8)
    if tff.size == 0 or tff.size % 4:
        raise ValueError("Invalid raw transfer function file " + path)
    return tff


def write_raw_tff(path, tff):
    import numpy as np
    with open(path, "w") as f:
        f.write("".join("%d " % int(c) for c in np.asarray(tff, dtype=np.uint8).reshape(-1)))


def read_cam_state(path):
    """JSON state file (MainWindow::loadCamState, mainwindow.cpp:374-410 and
    VolumeRenderWidget::read, volumerenderwidget.cpp:1457-1480).  Returns a dict with the keys
    present in the file: rotation (w, x, y, z), translation (x, y, z), imgResFactor, rayStepSize,
    useLerp, useAO, showContours, useAerial, showBox, useOrtho."""
    import json
    with open(path) as f:
        js = json.load(f)
    out = {}
    if "camRotation" in js:
        p = str(js["camRotation"]).split(" ")
        if len(p) >= 4:
            out["rotation"] = tuple(float(v) for v in p[:4])
    if "camTranslation" in js:
        p = str(js["camTranslation"]).split(" ")
        if len(p) >= 3:
            out["translation"] = tuple(float(v) for v in p[:3])
    for k in ("imgResFactor", "rayStepSize"):
        if isinstance(js.get(k), (int, float)) and not isinstance(js.get(k), bool):
            out[k] = float(js[k])
    for k in ("useLerp", "useAO", "showContours", "useAerial", "showBox", "useOrtho"):
        if isinstance(js.get(k), bool):
            out[k] = js[k]
    return out


def write_cam_state(path, rotation=DEFAULT_ROTATION, translation=DEFAULT_TRANSLATION, **flags):
    """MainWindow::saveCamState + VolumeRenderWidget::write (mainwindow.cpp:415-452,
    volumerenderwidget.cpp:1496-1505): floats go through double as QString::number does."""
    import json
    import numpy as np
    js = {"imgResFactor": float(flags.get("imgResFactor", 1.0)),
          "rayStepSize": float(flags.get("rayStepSize", 1.5))}
    for k, d in (("useLerp", True), ("useAO", False), ("showContours", False), ("useAerial", False),
                 ("showBox", False), ("useOrtho", False)):
        js[k] = bool(flags.get(k, d))
    js["camRotation"] = " ".join("%.6g" % float(np.float32(v)) for v in rotation)
    js["camTranslation"] = " ".join("%.6g" % float(np.float32(v)) for v in translation)
    with open(path, "w") as f:
        json.dump(js, f, indent=4)


def apply_cam_state(vr, state):
    """Apply a state read by read_cam_state to a VolumeRenderCL (what the GUI's widgets forward)."""
    vr.updateView(view_matrix(state.get("rotation", DEFAULT_ROTATION),
                              state.get("translation", DEFAULT_TRANSLATION)))
    if "rayStepSize" in state:
        vr.updateSamplingRate(state["rayStepSize"])
    if "useLerp" in state:
        vr.setLinearInterpolation(state["useLerp"])
    if "showContours" in state:
        vr.setContours(state["showContours"])
    if "useAerial" in state:
        vr.setAerial(state["useAerial"])
    if "useOrtho" in state:
        vr.setCamOrtho(state["useOrtho"])
    if "useAO" in state:
        vr.setAmbientOcclusion(state["useAO"])
    if state.get("showBox"):
        vr.setShowESS(True)
