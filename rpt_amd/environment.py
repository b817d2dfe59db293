"""`Environment` / `Hdri` (reference src/environment.rs:5-78).  Lookup runs on the device."""
import ctypes as C

import numpy as np

from . import _abi


class Hdri:
    def __init__(self, width, height, buf):  # Hdri::new, environment.rs:18-22
        buf = np.ascontiguousarray(buf, dtype=np.float64).reshape(-1, 3)
        assert len(buf) == width * height and width > 0 and height > 0
        self.width, self.height, self.buf = int(width), int(height), buf


class Environment:
    def __init__(self, color=(0.0, 0.0, 0.0), hdri=None):  # Default: black, environment.rs:64-68
        self.color = tuple(float(c) for c in color)
        self.hdri = hdri

    @staticmethod
    def Color(color):
        return Environment(color=color)

    @staticmethod
    def Hdri(hdri):
        return Environment(hdri=hdri)

    def lower_into(self, out, keep):
        out.color[:] = self.color
        if self.hdri is None:
            out.kind = _abi.RPT_ENV_COLOR
        else:
            out.kind = _abi.RPT_ENV_HDRI
            out.width, out.height = self.hdri.width, self.hdri.height
            keep.append(self.hdri.buf)
            out.texels = self.hdri.buf.ctypes.data_as(C.POINTER(C.c_double))
