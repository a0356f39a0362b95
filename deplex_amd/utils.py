"""Python mirror of the reference's `deplex.utils` (cpp/pybind/utils/utils.cpp:29-36):
DepthImage(image_path), .height, .width, .transform_to_pcd(intrinsics), .reset(image_path).

PNG decoding is file IO (the reference uses stb_image, utils/depth_image.cpp:30-40): the library's own host-side reader
(csrc/dpx_png.h, zlib for the inflation) with stb's conversion rules; the back-projection arithmetic (depth_image.cpp:55-78) runs in the HIP kernel behind dpx_backproject.
"""
from __future__ import annotations

import numpy as np

from . import _capi


def read_intrinsics(path) -> np.ndarray:
    """utils::readIntrinsics (utils/eigen_io.cpp:41-55): nine whitespace separated floats -> 3x3 float32."""
    try:
        with open(path) as fh:
            vals = fh.read().split()
    except OSError:
        raise _capi.DeplexError(_capi.DPX_ERR_IO, f"Error: Couldn't open intrinsics file {path}")
    return np.array(vals[:9], dtype=np.float32).reshape(3, 3)


def _load_u16(path) -> np.ndarray:
    """PNG file -> (H, W) uint16 through the library's reader, which applies the conversion rules of the reference's
    stbi_load_16(path, ..., STBI_grey) (utils/depth_image.cpp:30-40): colour -> luma, alpha dropped, 8-bit data * 257."""
    import ctypes as C
    h, w = C.c_int32(), C.c_int32()
    data = C.POINTER(C.c_uint16)()
    _capi.check(_capi.lib().dpx_read_depth_png(str(path).encode(), C.byref(h), C.byref(w), C.byref(data)))
    try:
        arr = np.ctypeslib.as_array(data, shape=(h.value, w.value)).copy()
    finally:
        _capi.lib().dpx_free(data)
    return arr


class DepthImage:
    def __init__(self, image_path, *, device: int = 0):
        self._device = device
        self._img = _load_u16(image_path)

    @classmethod
    def from_array(cls, depth, *, device: int = 0) -> "DepthImage":
        self = cls.__new__(cls)
        self._device = device
        self._img = np.ascontiguousarray(depth, dtype=np.uint16)
        return self

    @property
    def height(self) -> int:
        return int(self._img.shape[0])

    @property
    def width(self) -> int:
        return int(self._img.shape[1])

    @property
    def data(self) -> np.ndarray:
        return self._img

    def reset(self, image_path) -> None:
        self._img = _load_u16(image_path)

    def transform_to_pcd(self, intrinsics) -> np.ndarray:
        """(H*W, 3) float32 organised cloud; z = raw depth units (no scaling), like the reference."""
        K = np.asarray(intrinsics, dtype=np.float32).reshape(3, 3)
        h, w = self._img.shape
        xyz = np.empty((h * w, 3), dtype=np.float32)
        _capi.check(_capi.lib().dpx_backproject(self._img.ctypes.data, h, w, float(K[0, 0]), float(K[1, 1]),
                                               float(K[0, 2]), float(K[1, 2]), xyz.ctypes.data, self._device))
        return xyz
