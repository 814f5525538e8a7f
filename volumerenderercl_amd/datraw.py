"""Python access to the C++ host loader (libvrhost.so, include/vrhost.h): the counterpart of the
reference's DatRawReader (/root/reference/src/io/datrawreader.h:38-184).  Pure host code."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvrhost.so")

UCHAR, USHORT, FLOAT, DOUBLE, UNKNOWN_FORMAT = range(5)
_NP = {UCHAR: np.uint8, USHORT: np.uint16, FLOAT: np.float32}


class _Info(C.Structure):
    _fields_ = [("res", C.c_uint32 * 4), ("thickness", C.c_double * 3), ("format", C.c_int32),
                ("endianness", C.c_int32), ("min_value", C.c_float), ("max_value", C.c_float),
                ("n_timesteps", C.c_uint64), ("bytes_per_timestep", C.c_uint64),
                ("channel_order", C.c_char * 16)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvrhost.so is not built: make -C volumerenderercl_amd/csrc/host")
        # libvrhost links libvrhip (the VolumeRenderCL class lives in it too); make sure the
        # sibling library resolves when the loader is used on its own
        C.CDLL(os.path.join(_HERE, "libvrhip.so"), mode=C.RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        lib.vrdr_load.restype = C.c_int
        lib.vrdr_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.vrdr_error.restype = C.c_char_p
        lib.vrdr_free.argtypes = [C.c_void_p]
        lib.vrdr_info.argtypes = [C.c_void_p, C.POINTER(_Info)]
        lib.vrdr_data.restype = C.c_void_p
        lib.vrdr_data.argtypes = [C.c_void_p, C.c_uint64]
        lib.vrdr_histogram.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        lib.vrhost_load_hdr.restype = C.c_int
        lib.vrhost_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.vrhost_free_pixels.argtypes = [C.POINTER(C.c_float)]
        _lib = lib
    return _lib


class Properties:
    """DatRawReader::Properties (datrawreader.h:59-103)."""

    def __init__(self, dat_file_name="", raw_file_names=None):
        self.dat_file_name = dat_file_name
        self.raw_file_names = list(raw_file_names or [])
        self.volume_res = [0, 0, 0, 1]
        self.slice_thickness = [1.0, 1.0, 1.0]
        self.format = UNKNOWN_FORMAT
        self.endianness = 0
        self.image_channel_order = "R"
        self.min_value = 0.0
        self.max_value = 0.0


class DatRawReader:
    def __init__(self):
        self._h = C.c_void_p()
        self._prop = None
        self._data = []
        self._hist = []

    def read_files(self, props):
        """Raises ValueError (std::invalid_argument) / RuntimeError (std::runtime_error)."""
        lib = _load()
        h = C.c_void_p()
        raw = props.raw_file_names[0] if props.raw_file_names else None
        rc = lib.vrdr_load(props.dat_file_name.encode(), raw.encode() if raw else None, C.byref(h))
        if rc != 0:
            msg = lib.vrdr_error().decode()
            raise ValueError(msg) if rc == 1 else RuntimeError(msg)
        info = _Info()
        lib.vrdr_info(h, C.byref(info))
        p = Properties(props.dat_file_name, props.raw_file_names)
        p.volume_res = list(info.res)
        p.slice_thickness = list(info.thickness)
        p.format = info.format
        p.endianness = info.endianness
        p.image_channel_order = info.channel_order.decode()
        p.min_value, p.max_value = float(info.min_value), float(info.max_value)
        self._data, self._hist = [], []
        for t in range(info.n_timesteps):
            buf = C.string_at(lib.vrdr_data(h, t), info.bytes_per_timestep)
            if info.format in _NP:
                self._data.append(np.frombuffer(buf, dtype=_NP[info.format]).copy())
            else:
                self._data.append(np.frombuffer(buf, dtype=np.uint8).copy())
            hist = (C.c_double * 256)()
            lib.vrdr_histogram(h, t, hist)
            self._hist.append(np.array(hist))
        lib.vrdr_free(h)
        self._prop = p

    def has_data(self):
        return bool(self._data)

    def properties(self):
        if not self._data:
            raise RuntimeError("No properties of volume data set available.")
        return self._prop

    def data(self):
        if not self._data:
            raise RuntimeError("No data available.")
        return self._data

    def histograms(self):
        return self._hist


def load_hdr(file_name):
    """Radiance .hdr -> float32 [height, width, 4] (RGB, alpha 0), through the host layer's decoder
    (reference: inc/hdr_loader.h load_hdr_float4).  Raises RuntimeError like createEnvironmentMap
    (volumerendercl.cpp:1137-1138)."""
    lib = _load()
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(0), C.c_uint32(0)
    if lib.vrhost_load_hdr(os.fsencode(file_name), C.byref(p), C.byref(w), C.byref(h)) != 0:
        raise RuntimeError("Error loading environment map file.")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.vrhost_free_pixels(p)
