import sys, numpy as np
sys.path.insert(0, "/root/repo")
from volumerenderercl_amd import VolumeRenderCL, frontend
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume("shells", (2048,)*3, 0)
vr.setTransferFunction(frontend.tff_from_stops())
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
for V in (1024, 2048):
    vr.setSeed(3499211612); vr.setIteration(0)
    img = vr.runRaycastNoGL(V, V)
    a = img[..., 3] > 0
    nonbg = np.abs(img[..., :3] - 1.0).max(axis=-1) > 0
    for T in (64, 32, 16, 8):
        t = a.reshape(V // T, T, V // T, T).any(axis=(1, 3))
        t2 = nonbg.reshape(V // T, T, V // T, T).any(axis=(1, 3))
        print("viewport %d tile %2d: tiles with alpha > 0: %.3f   with a non-background colour: %.3f" % (V, T, t.mean(), t2.mean()))
vr.close()
