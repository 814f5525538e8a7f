"""The reference GUI's saved files: camera state JSON, gradient-stop and raw
transfer-function files -- readers/writers of the headless front end."""
import numpy as np
import pytest

from volumerenderercl_amd import frontend


def test_cam_state_round_trip(tmp_path):
    q = frontend.quat_from_axis_angle((1, 1, 0), 30.0)
    p = tmp_path / "state.json"
    frontend.write_cam_state(str(p), q, (0.25, -0.5, 3.0), rayStepSize=2.0, useOrtho=True,
                             showContours=True)
    st = frontend.read_cam_state(str(p))
    np.testing.assert_allclose(st["rotation"], q, rtol=1e-5)
    assert st["translation"] == (0.25, -0.5, 3.0)
    assert st["rayStepSize"] == 2.0 and st["useOrtho"] is True and st["showContours"] is True
    assert st["useAO"] is False and st["useLerp"] is True


def test_cam_state_as_qt_writes_it(tmp_path):
    # QJsonDocument::toJson(): 4-space indent, keys sorted, numbers as doubles
    text = """{
    "camRotation": "0.965926 0.183013 0.183013 0",
    "camTranslation": "0 0 2",
    "imgResFactor": 1,
    "rayStepSize": 1.5,
    "showBox": false,
    "showContours": false,
    "useAO": false,
    "useAerial": true,
    "useLerp": true,
    "useOrtho": false
}
"""
    p = tmp_path / "qt.json"
    p.write_text(text)
    st = frontend.read_cam_state(str(p))
    assert st["rotation"] == (0.965926, 0.183013, 0.183013, 0.0) and st["translation"] == (0.0, 0.0, 2.0)
    assert st["useAerial"] is True and st["imgResFactor"] == 1.0
    m = frontend.view_matrix(st["rotation"], st["translation"])
    ref = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    np.testing.assert_allclose(m, ref, atol=2e-5)


def test_tff_files_round_trip(tmp_path):
    stops = [(0.0, (0, 0, 0, 0)), (0.35, (200, 40, 10, 30)), (1.0, (255, 255, 255, 255))]
    p = tmp_path / "a.tff"
    frontend.write_tff_stops(str(p), stops)
    assert frontend.read_tff_stops(str(p)) == stops
    table = frontend.tff_from_stops(stops)
    r = tmp_path / "raw.tff"
    frontend.write_raw_tff(str(r), table)
    np.testing.assert_array_equal(frontend.read_raw_tff(str(r)), np.asarray(table, np.uint8).reshape(-1))
    (tmp_path / "empty.tff").write_text("# nothing\n1 2 3\n")
    with pytest.raises(ValueError):
        frontend.read_tff_stops(str(tmp_path / "empty.tff"))


def test_view_matrix_against_an_independent_composition():
    """updateViewMatrix (volumerenderwidget.cpp:1079-1098) composes QMatrix4x4 operations:
    identity . rotate(q) . translate(t) . scale(t.z), each post-multiplying, and hands the matrix over
    transposed (Qt stores column-major, the kernel reads rows).  Restated here with scipy's
    quaternion-to-matrix conversion (Qt's scalar-first (w, x, y, z) -> scipy's (x, y, z, w)) and
    explicit 4x4 products -- an implementation that shares no code with frontend.view_matrix."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    cases = [((1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 2.0)), (frontend.quat_from_axis_angle((1, 1, 0), 30.0), (0.0, 0.0, 2.0))]
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        cases.append((tuple(q), tuple(rng.uniform(-1.5, 3.0, 3))))
    for q, t in cases:
        R = np.eye(4)
        R[:3, :3] = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        T = np.eye(4)
        T[:3, 3] = t
        S = np.diag([t[2], t[2], t[2], 1.0])
        want = (R @ T @ S).astype(np.float32).reshape(-1)      # row-major = the transposed column-major data
        got = np.array(frontend.view_matrix(q, t), dtype=np.float32)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    # the default camera of SURVEY 8(d): rot = identity, translation (0, 0, 2)
    assert frontend.view_matrix() == [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 2, 0, 0, 0, 1]
