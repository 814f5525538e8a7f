"""Host-side mirror of the reference's `VolumeRenderCL` interface over the C ABI.

The production host is the C++ class in csrc/host/ (same interface, the reference is
compiled code); this Python twin exists so tests and bench.py can drive libvrhip.so
through exactly the calls the reference's caller makes (volumerenderwidget.cpp), with
the reference's method names, argument meaning, call order and error behaviour
(/root/reference/src/core/volumerendercl.h:123-356).
"""
import ctypes as C

import numpy as np

from . import _lib, frontend
from ._lib import (CameraParams, PathtraceParams, RaycastParams, RenderingParams, Stats,
                   UCHAR, USHORT, FLOAT)

NP_DTYPE = {UCHAR: np.uint8, USHORT: np.uint16, FLOAT: np.float32}
_MT19937_DEFAULT_SEED = 5489   # std::mt19937 default constructor


def _u3(v):
    return (C.c_uint32 * 3)(*[int(x) for x in v])


class VolumeRenderCL:
    TECH_RAYCAST, TECH_PATHTRACE = 0, 1

    def __init__(self):
        self._lib = None
        self._h = C.c_void_p()
        self._vol_loaded = False
        self._model_scale = np.ones(3, dtype=np.float32)
        self._camera = CameraParams()
        self._camera.viewMat[:] = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
        self._camera.bbox_bl[:] = [-1, -1, -1, 0]
        self._camera.bbox_tr[:] = [1, 1, 1, 0]
        self._rendering = RenderingParams()
        self._rendering.backgroundColor[:] = [1, 1, 1, 1]
        self._rendering.modelScale[:] = [1, 1, 1, 0]
        self._rendering.illumType = 1
        self._rendering.useLinear = 1
        self._rendering.seed = 42
        self._raycast = RaycastParams()
        self._raycast.samplingRate = 1.5
        self._raycast.brickRes[:] = [1, 1, 1, 0]
        self._pathtrace = PathtraceParams(100.0)
        self._timestep = 0
        self._res = [0, 0, 0, 1]
        self._thickness = [1.0, 1.0, 1.0]
        self._format = None
        self._histograms = []
        # the reference's member generator is default-seeded (volumerendercl.cpp:67 shadows it)
        self._generator = frontend.Mt19937(_MT19937_DEFAULT_SEED)
        self._fixed_seed = None
        self._out_size = (0, 0)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != _lib.OK:
            msg = self._lib.vrhip_last_error(self._h)
            msg = msg.decode() if msg else "vrhip error %d" % rc
            if rc == _lib.ERR_INVALID:
                raise ValueError(msg)        # std::invalid_argument
            raise RuntimeError(msg)          # std::runtime_error

    def _push_params(self):
        ms = self._model_scale
        self._rendering.modelScale[:] = [float(ms[0]), float(ms[1]), float(ms[2]), 0.0]
        self._check(self._lib.vrhip_set_camera_params(self._h, C.byref(self._camera)))
        self._check(self._lib.vrhip_set_rendering_params(self._h, C.byref(self._rendering)))
        self._check(self._lib.vrhip_set_raycast_params(self._h, C.byref(self._raycast)))
        self._check(self._lib.vrhip_set_pathtrace_params(self._h, C.byref(self._pathtrace)))

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._lib

    # ------------------------------------------------------------------ interface
    def initialize(self, useGL=False, useCPU=False, vendor=None, deviceName="", platformId=-1,
                   device_id=0):
        """volumerendercl.cpp:95-159.  useGL / useCPU / vendor select OpenCL plumbing that
        does not exist here; useCPU is refused loudly (no CPU path in the product)."""
        if useCPU:
            raise RuntimeError("ERROR: no CPU device path; libvrhip renders on MI355X only")
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.vrhip_create(int(device_id), C.byref(h))
        if rc != _lib.OK:
            msg = self._lib.vrhip_last_error(None)
            raise RuntimeError(msg.decode() if msg else "vrhip_create failed")
        self._h = h
        self._device_id = int(device_id)

    def setRoundBudget(self, rounds):
        """Scheduling knob of the two-phase march (vrhip_set_round_budget): 10 (default) for the
        shortest single frame, ~32 when several frames are in flight.  No effect on pixels."""
        self._round_budget = int(rounds)
        self._check(self._lib.vrhip_set_round_budget(self._h, int(rounds)))

    def shareVolumes(self):
        """A second renderer on the same GPU that renders from THIS renderer's voxels and ESS
        bricks (vrhip_share_volumes) with everything else of its own -- transfer function, kernel
        arguments, frame buffer, scratch, stream: one frame each can be in flight on two streams
        over a single copy of the volume.  Starts as a copy of this renderer's settings; this
        renderer must keep its volumes (no load / clear / close) while the twin is alive."""
        if not self._vol_loaded:
            raise RuntimeError("No volume data is loaded.")
        twin = VolumeRenderCL()
        twin.initialize(device_id=getattr(self, "_device_id", 0))
        twin._check(twin._lib.vrhip_share_volumes(twin._h, self._h))
        for name in ("_res", "_thickness", "_histograms"):
            setattr(twin, name, list(getattr(self, name)))
        twin._format = self._format
        twin._model_scale = self._model_scale.copy()
        twin._timestep = self._timestep
        twin._props = getattr(self, "_props", None)
        for name in ("_camera", "_rendering", "_raycast", "_pathtrace"):
            C.memmove(C.byref(getattr(twin, name)), C.byref(getattr(self, name)),
                      C.sizeof(getattr(self, name)))
        twin._fixed_seed = self._fixed_seed
        twin._vol_loaded = True
        twin._check(twin._lib.vrhip_set_timestep(twin._h, int(self._timestep)))
        twin._check(twin._lib.vrhip_set_object_ess(twin._h, 1 if getattr(self, "_obj_ess", True) else 0))
        if getattr(self, "_tff", None) is not None:
            twin.setTransferFunction(self._tff)
            if getattr(self, "_prefix", None) is not None:
                twin.setTffPrefixSum(self._prefix)
        twin._rendering.iteration = self._rendering.iteration
        if getattr(self, "_round_budget", None) is not None:
            twin.setRoundBudget(self._round_budget)
        return twin

    def close(self):
        if self._lib is not None and self._h:
            self._lib.vrhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr, use_own=False):
        """Launch on the given hipStream_t handle (0 = legacy default stream)."""
        self._check(self._lib.vrhip_set_stream(self._h, C.c_void_p(stream_ptr),
                                               1 if use_own else 0))

    def get_stream(self):
        """The hipStream_t handle (int, 0 = legacy default stream) render calls are enqueued on."""
        p = C.c_void_p()
        self._check(self._lib.vrhip_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def getPlatformNames(self):
        """volumerendercl.cpp:1062-1079 (OpenCL platforms): there is one, the HIP runtime."""
        return ["AMD HIP (ROCm)"]

    def getDeviceNames(self, platformId=0, type="GPU"):
        """volumerendercl.cpp:1087-1108: the devices of a platform by type; no CPU device here."""
        return [] if type == "CPU" else [self.getCurrentDeviceName()]

    def getCurrentDeviceName(self):
        buf = C.create_string_buffer(256)
        self._check(self._lib.vrhip_device_name(self._h, buf, 256))
        return buf.value.decode()

    def updateView(self, viewMat):
        """volumerendercl.cpp:379-390: 16 floats, row-major; silently ignored without data."""
        if not self._vol_loaded:
            return
        self._camera.viewMat[:] = [float(x) for x in viewMat]
        self._rendering.iteration = 0

    def updateSamplingRate(self, samplingRate):
        self._raycast.samplingRate = float(samplingRate)

    def updateOutputImg(self, width, height, texId=0):
        """volumerendercl.cpp:465-499: output buffers are sized by the render call; the hit images
        of image-order ESS are re-created with their initial contents (:482-488)."""
        self._out_size = (int(width), int(height))
        if self._h:
            self._check(self._lib.vrhip_reset_image_ess(self._h))

    def getImageEss(self, width, height):
        """(hit_in, hit_out) of image-order ESS for a width x height frame: uint8
        [(height/8+1), (width/8+1)]; hit_in is what the next imgEss frame reads."""
        shape = (int(height) // 8 + 1, int(width) // 8 + 1)
        a, b = np.zeros(shape, np.uint8), np.zeros(shape, np.uint8)
        self._check(self._lib.vrhip_get_image_ess(self._h, int(width), int(height),
                                                  a.ctypes.data, b.ctypes.data))
        return a, b

    def setImageEss(self, width, height, hit_in=None, hit_out=None):
        shape = (int(height) // 8 + 1, int(width) // 8 + 1)
        ptrs = []
        keep = []
        for h in (hit_in, hit_out):
            if h is None:
                ptrs.append(None)
                continue
            h = np.ascontiguousarray(h, dtype=np.uint8)
            if h.shape != shape:
                raise ValueError("hit image must have shape %r" % (shape,))
            keep.append(h)
            ptrs.append(h.ctypes.data)
        self._check(self._lib.vrhip_set_image_ess(self._h, int(width), int(height), *ptrs))

    def _begin_frame(self):
        # setMemObjectsRaycast: fresh seed per frame (volumerendercl.cpp:212)
        if self._fixed_seed is None:
            self._rendering.seed = self._generator()
        else:
            self._rendering.seed = self._fixed_seed
        self._push_params()

    def runRaycast(self, width, height, out_dev_ptr=None):
        """volumerendercl.cpp:506-558: frame stays on the GPU; iteration advances (:540)."""
        if not self._vol_loaded:
            return
        self._begin_frame()
        self._check(self._lib.vrhip_render_frame(self._h, int(width), int(height),
                                                 C.c_void_p(out_dev_ptr), 1 if out_dev_ptr else 0))
        self._rendering.iteration += 1

    def runRaycastNoGL(self, width, height, output=None):
        """volumerendercl.cpp:568-607 with the documented contract honoured:
        returns width*height*4 float32, row-major RGBA, row 0 = top."""
        if not self._vol_loaded:
            return output
        self._begin_frame()
        out = np.empty((int(height), int(width), 4), dtype=np.float32)
        self._check(self._lib.vrhip_render_frame(self._h, int(width), int(height),
                                                 out.ctypes.data_as(C.c_void_p), 0))
        self._rendering.iteration += 1   # accumulation advances in both paths
        if output is not None:
            output[:] = out.reshape(-1).tolist()
        return out

    def render_tiles(self, width, height, tile_w, tile_h, tile_ids, out_dev_ptr):
        """Image-tile decomposition entry; no reference counterpart."""
        if not self._vol_loaded:
            return
        self._begin_frame()
        ids = np.ascontiguousarray(tile_ids, dtype=np.uint32)
        self._check(self._lib.vrhip_render_tiles(self._h, int(width), int(height), int(tile_w),
                                                 int(tile_h), ids.ctypes.data_as(C.c_void_p),
                                                 int(ids.size), C.c_void_p(out_dev_ptr)))

    def render_batch(self, width, height, seeds, out_dev_ptr, tile_w=0, tile_h=0, tile_ids=None,
                     frame_stride=0):
        """len(seeds) <= 256 independent frames (frame f jittered by seeds[f]) in one set of
        launches (vrhip_render_batch): whole frames into out[f][height][width][4], or -- with
        tile_ids -- the tile subset into out[f][n_tiles][tile_h][tile_w][4] (device memory);
        frame_stride: pixels between the frames of `out` when they are not packed."""
        if not self._vol_loaded:
            return
        self._rendering.iteration = 0
        self._push_params()
        sd = np.ascontiguousarray(seeds, dtype=np.uint32)
        ids = None if tile_ids is None else np.ascontiguousarray(tile_ids, dtype=np.uint32)
        self._check(self._lib.vrhip_render_batch(
            self._h, int(width), int(height), int(tile_w), int(tile_h),
            None if ids is None else ids.ctypes.data_as(C.c_void_p), 0 if ids is None else int(ids.size),
            sd.ctypes.data_as(C.c_void_p), int(sd.size), C.c_void_p(out_dev_ptr), int(frame_stride)))

    # ---- volume
    def loadVolumeData(self, props):
        """volumerendercl.cpp:765-805. `props` is a datraw.Properties (dat_file_name or
        raw_file_names set)."""
        from . import datraw
        self._vol_loaded = False
        try:
            reader = datraw.DatRawReader()
            reader.read_files(props)
        except ValueError as e:      # std::invalid_argument -> runtime_error (:784-787)
            raise RuntimeError(str(e))
        p = reader.properties()
        vols = reader.data()
        self._histograms = reader.histograms()
        self._props = p
        co = p.image_channel_order
        if co in ("R", "", "I", "LUMINANCE"):
            channels = 1
        elif co == "RG":
            channels = 2
        elif co == "RGBA":
            channels = 4
        elif co in ("ARGB", "BGRA"):   # uploaded by the reference, but its kernel has no branch for them
            raise RuntimeError("ARGB / BGRA volumes are not supported.")
        else:
            raise RuntimeError("Unknown or invalid volume color format.")   # :711
        return self.loadVolumeArrays(vols, p.format, p.slice_thickness, p.volume_res[:3],
                                     channels=channels)

    def loadVolumeArrays(self, volumes, fmt, thickness=(1.0, 1.0, 1.0), res=None, channels=None):
        """Upload already-decoded time steps (ndarray [z, y, x], [z, y, x, c] for CL_RG / CL_RGBA
        volumes, or flat with `res` and `channels`) -- the volDataToCLmem + calcScaling + default
        prefix-sum part of loadVolumeData."""
        self._vol_loaded = False
        self._check(self._lib.vrhip_clear_volumes(self._h))
        for t, v in enumerate(volumes):
            v = np.ascontiguousarray(v, dtype=NP_DTYPE[fmt])
            r = res if res is not None else (v.shape[2], v.shape[1], v.shape[0])
            nch = channels if channels is not None else (v.shape[3] if v.ndim == 4 else 1)
            if v.size < int(r[0]) * int(r[1]) * int(r[2]) * nch:
                raise RuntimeError("Volume size does not match size specified in dat file.")
            self._check(self._lib.vrhip_upload_volume_channels(
                self._h, v.ctypes.data_as(C.c_void_p), _u3(r), fmt, int(nch), t))
        self._res = [int(r[0]), int(r[1]), int(r[2]), len(volumes)]
        self._format = fmt
        self._thickness = [float(x) for x in thickness]
        self._calc_scaling()
        # default prefix sum of the linear ramp (volumerendercl.cpp:795-801)
        prefix = np.cumsum(np.arange(1024, dtype=np.uint64) * 4).astype(np.uint32)
        self._vol_loaded = True
        self.setTffPrefixSum(prefix)
        return len(volumes)

    def synthVolume(self, kind, res, fmt):
        """SURVEY 8(d) synthetic input generated in HBM (no host copy)."""
        self._vol_loaded = False
        self._check(self._lib.vrhip_clear_volumes(self._h))
        self._check(self._lib.vrhip_synth_volume(self._h, {"sphere": 0, "shells": 1}[kind],
                                                 _u3(res), fmt, 0))
        self._res = [int(res[0]), int(res[1]), int(res[2]), 1]
        self._format = fmt
        self._thickness = [1.0, 1.0, 1.0]
        self._calc_scaling()
        self._vol_loaded = True

    def downsampleVolume(self, t, factor):
        """Device part of volumeDownsampling (volumerendercl.cpp:238-300 + kernel
        `downsampling`): returns the low-res ndarray [z, y, x] of time step t."""
        lo = (C.c_uint32 * 3)()
        self._check(self._lib.vrhip_downsample_volume(self._h, int(t), int(factor), None, 0, lo))
        out = np.empty((lo[2], lo[1], lo[0]), dtype=NP_DTYPE[self._format])
        self._check(self._lib.vrhip_downsample_volume(self._h, int(t), int(factor),
                                                      out.ctypes.data_as(C.c_void_p), out.nbytes, lo))
        return out

    def volumeDownsampling(self, t, factor):
        """volumerendercl.cpp:238-341: down-sample time step t on the GPU and write
        <dat name>_<N>.raw / .dat next to the loaded .dat (the reference's text, quirks included:
        `Format:` carries the enum's integer, `ObjectFileName:` is cut with substr(first + 1,
        lastindex)).  Returns the path without extension."""
        props = getattr(self, "_props", None)
        if not self._vol_loaded or props is None or not props.dat_file_name:
            raise RuntimeError("No volume data is loaded.")
        if factor < 2:
            raise ValueError("Factor must be greater or equal 2.")
        low = self.downsampleVolume(t, factor)
        name = props.dat_file_name
        dot = name.rfind(".")
        rawname = (name[:dot] if dot >= 0 else name) + "_" + str(low.shape[2])
        low.tofile(rawname + ".raw")
        last = rawname.rfind(".")
        first = max(rawname.rfind("/"), rawname.rfind("\\"))
        short = rawname[first + 1:] if last < 0 else rawname[first + 1:first + 1 + last]
        with open(rawname + ".dat", "w") as f:
            f.write("ObjectFileName: \t%s.raw\n" % short)
            f.write("Resolution: \t\t%d %d %d\n" % (low.shape[2], low.shape[1], low.shape[0]))
            f.write("SliceThickness: \t%g %g %g\n" % tuple(props.slice_thickness[:3]))
            f.write("Format: \t\t\t%d\n" % int(props.format))
        return rawname

    def downloadVolume(self, t=0):
        out = np.empty((self._res[2], self._res[1], self._res[0]), dtype=NP_DTYPE[self._format])
        self._check(self._lib.vrhip_download_volume(self._h, t, out.ctypes.data_as(C.c_void_p),
                                                    out.nbytes))
        return out

    def _calc_scaling(self):
        """calcScaling, volumerendercl.cpp:347-362 (valarray<float> arithmetic)."""
        res = np.array(self._res[:3], dtype=np.float32)
        th = np.array(self._thickness, dtype=np.float32)
        s = res * (th * (np.float32(1.0) / th[0]))
        self._model_scale = (s.max() / s).astype(np.float32)

    def hasData(self):
        return self._vol_loaded

    def getResolution(self):
        if not self._vol_loaded:
            return [0, 0, 0, 1]
        return list(self._res)

    def createEnvironmentMap(self, file_name):
        """volumerendercl.cpp:1121-1150: Radiance .hdr file -> float RGBA environment map; an
        empty name installs the reference's 1x1 white map, which the kernel never samples."""
        if not file_name:
            self.setEnvironmentMap(None)
            return
        from . import datraw
        self.setEnvironmentMap(datraw.load_hdr(file_name))

    def setEnvironmentMap(self, rgba):
        """float32 [height, width, 4] texels, or None to remove the map."""
        if rgba is None:
            self._check(self._lib.vrhip_set_environment_map(self._h, None, 0, 0))
            return
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("environment map must be [height, width, 4]")
        self._check(self._lib.vrhip_set_environment_map(self._h, rgba.ctypes.data,
                                                        rgba.shape[1], rgba.shape[0]))

    def getHistogram(self, timestep=0):
        if not self._vol_loaded:
            raise ValueError("Invalid timestep for histogram data.")
        return self._histograms[timestep]

    def scaleVolume(self, scale):
        self._model_scale = (self._model_scale * np.asarray(scale, dtype=np.float32)).astype(
            np.float32)

    # ---- transfer function
    def setTransferFunction(self, tff):
        """volumerendercl.cpp:864-891: upload RGBA8 table, rebuild the ESS bricks, set the
        inclusive prefix sum of the alpha bytes, reset the iteration."""
        if not self._vol_loaded:
            return
        tff = np.ascontiguousarray(tff, dtype=np.uint8).reshape(-1)
        self._tff = tff.copy()
        self._check(self._lib.vrhip_set_transfer_function(
            self._h, tff.ctypes.data_as(C.c_void_p), tff.size // 4))
        self._generate_bricks()
        prefix = np.cumsum(tff[3::4].astype(np.uint64)).astype(np.uint32)
        self.setTffPrefixSum(prefix)
        self._rendering.iteration = 0

    def setTffPrefixSum(self, prefix):
        if not self._vol_loaded:
            return
        prefix = np.ascontiguousarray(prefix, dtype=np.uint32)
        self._prefix = prefix.copy()
        self._check(self._lib.vrhip_set_tff_prefix_sum(
            self._h, prefix.ctypes.data_as(C.c_void_p), prefix.size))

    def _generate_bricks(self):
        self._check(self._lib.vrhip_build_bricks(self._h))
        brf = (C.c_float * 3)()
        self._check(self._lib.vrhip_get_brick_info(self._h, None, brf, None))
        self._raycast.brickRes[:] = [brf[0], brf[1], brf[2], 0.0]   # :627-631

    def brickInfo(self):
        tex, brf, edge = (C.c_uint32 * 3)(), (C.c_float * 3)(), (C.c_uint32 * 3)()
        self._check(self._lib.vrhip_get_brick_info(self._h, tex, brf, edge))
        return list(tex), list(brf), list(edge)

    def downloadBricks(self, t=0):
        tex, _, _ = self.brickInfo()
        out = np.empty((tex[2], tex[1], tex[0], 2), dtype=NP_DTYPE[self._format])
        self._check(self._lib.vrhip_download_bricks(self._h, t, out.ctypes.data_as(C.c_void_p),
                                                    out.nbytes))
        return out

    def downloadCells(self, fine=False):
        """(min, max) of every cell of the cell grid (vrhip_download_cells; fine: the grid of the ray
        caster's empty bits, vrhip_download_empty_cells): ndarray [cz, cy, cx, 2], shift."""
        fn = self._lib.vrhip_download_empty_cells if fine else self._lib.vrhip_download_cells
        dims, shift = (C.c_uint32 * 3)(), C.c_uint32()
        self._check(fn(self._h, None, 0, dims, C.byref(shift)))
        out = np.empty((dims[2], dims[1], dims[0], 2), dtype=np.float32)
        self._check(fn(self._h, out.ctypes.data_as(C.c_void_p), out.size, dims, C.byref(shift)))
        return out, int(shift.value)

    def lastBricksSeconds(self):
        return float(self._lib.vrhip_last_bricks_seconds(self._h))

    # ---- setters (volumerendercl.cpp:922-1047,1157-1174)
    def setCamOrtho(self, v):
        self._camera.ortho = 1 if v else 0

    def setIllumination(self, illum):
        self._rendering.illumType = int(illum)

    def setAmbientOcclusion(self, ao):
        self._raycast.useAO = 1 if ao else 0

    def setShowESS(self, v):
        self._rendering.showEss = 1 if v else 0

    def setLinearInterpolation(self, v):
        self._rendering.useLinear = 1 if v else 0

    def setContours(self, v):
        self._raycast.contours = 1 if v else 0

    def setAerial(self, v):
        self._raycast.aerial = 1 if v else 0

    def setImgEss(self, v):
        self._rendering.imgEss = 1 if v else 0

    def setObjEss(self, v):
        """The reference rebuilds its program with/without -DESS; here a variant switch."""
        self._obj_ess = bool(v)
        self._check(self._lib.vrhip_set_object_ess(self._h, 1 if v else 0))

    def setBackground(self, color):
        # the reference converts through cl_float3: alpha becomes 0 (volumerendercl.cpp:1027)
        self._rendering.backgroundColor[:] = [float(color[0]), float(color[1]), float(color[2]),
                                              0.0]

    def setUseGradient(self, v):
        self._rendering.useGradient = 1 if v else 0

    def setTechnique(self, tech):
        self._rendering.technique = int(tech)
        self._rendering.iteration = 0

    def setExtinction(self, extinction):
        self._pathtrace.max_extinction = float(extinction)

    def setBBox(self, bl_x, bl_y, bl_z, tr_x, tr_y, tr_z):
        self._camera.bbox_bl[:] = [bl_x, bl_y, bl_z, 0]
        self._camera.bbox_tr[:] = [tr_x, tr_y, tr_z, 0]
        self._rendering.iteration = 0

    def setTimestep(self, t):
        if self._vol_loaded and t >= self._res[3]:
            return
        self._timestep = int(t)
        self._check(self._lib.vrhip_set_timestep(self._h, int(t)))
        self._rendering.iteration = 0

    def getLastExecTime(self):
        return float(self._lib.vrhip_last_kernel_seconds(self._h))

    def setPhaseTiming(self, v):
        """Record an event between the two phases of a frame (vrhip_set_phase_timing; off by default:
        the event costs GPU time)."""
        self._check(self._lib.vrhip_set_phase_timing(self._h, 1 if v else 0))

    def setFrameTiming(self, v):
        """Record the two events around a frame's launches (vrhip_set_frame_timing; on by default, as the
        reference times every frame, volumerendercl.cpp:545-551).  Off: getLastExecTime() answers 0 and frames
        that follow each other without a wait lose the events' GPU time."""
        self._check(self._lib.vrhip_set_frame_timing(self._h, 1 if v else 0))

    def getLastPhaseTimes(self):
        """(phase-1 seconds, phase-2 seconds) of the last ray-cast pass rendered with setPhaseTiming(True);
        (0.0, 0.0) when the last pass was not phase-timed."""
        a, b = C.c_double(), C.c_double()
        if self._lib.vrhip_last_phase_seconds(self._h, C.byref(a), C.byref(b)) != _lib.OK:
            return 0.0, 0.0
        return float(a.value), float(b.value)

    def lastLaunchInfo(self):
        """What the last render call launched (vrhip_last_launch_info): kernel variants, waves per workgroup,
        round budget, lookahead, frames of the set -- as a dict."""
        li = _lib.LaunchInfo()
        self._check(self._lib.vrhip_last_launch_info(self._h, C.byref(li)))
        return li.as_dict()

    # ---- test / bench conveniences (not in the reference)
    def setSeed(self, seed):
        """Pin the per-frame jitter seed (None restores the mt19937 sequence).  params() shows a pinned seed at
        once, not only after the next frame."""
        self._fixed_seed = None if seed is None else int(seed) & 0xFFFFFFFF
        if self._fixed_seed is not None:
            self._rendering.seed = self._fixed_seed

    def setIteration(self, it):
        self._rendering.iteration = int(it)

    def setStatsEnabled(self, v):
        self._check(self._lib.vrhip_set_stats_enabled(self._h, 1 if v else 0))

    def getStats(self):
        st = Stats()
        self._check(self._lib.vrhip_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def countTouched(self, width, height, want_bitmap=False):
        n = C.c_uint64()
        bm = None
        if want_bitmap:
            mb = [(r + 3) // 4 for r in self._res[:3]]
            bm = np.zeros((mb[0] * mb[1] * mb[2] + 7) // 8, dtype=np.uint8)
        self._push_params()
        self._check(self._lib.vrhip_count_touched(
            self._h, int(width), int(height), C.byref(n),
            bm.ctypes.data_as(C.c_void_p) if bm is not None else None,
            bm.nbytes if bm is not None else 0))
        return int(n.value), bm

    def countFetched(self, width, height):
        """technique 1: micro-bricks the product's own fetches touch (vrhip_count_fetched)."""
        n = C.c_uint64()
        self._push_params()
        self._check(self._lib.vrhip_count_fetched(self._h, int(width), int(height), C.byref(n)))
        return int(n.value)

    def countTouchedTiles(self, width, height, tile_w, tile_h, tile_ids):
        n = C.c_uint64()
        ids = np.ascontiguousarray(tile_ids, dtype=np.uint32)
        self._push_params()
        self._check(self._lib.vrhip_count_touched_tiles(
            self._h, int(width), int(height), int(tile_w), int(tile_h),
            ids.ctypes.data_as(C.c_void_p), int(ids.size), C.byref(n)))
        return int(n.value)

    def params(self):
        """Copies of the four kernel-argument structs as they will be pushed."""
        self._rendering.modelScale[:] = [float(self._model_scale[0]), float(self._model_scale[1]),
                                         float(self._model_scale[2]), 0.0]
        return self._camera, self._rendering, self._raycast, self._pathtrace
