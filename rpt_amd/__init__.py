"""rpt_amd — the MI355X-native back-end for rpt's hot path, behind rpt's own builder API.

`from rpt_amd import *` gives the names a user of the reference crate has after
`use rpt::*` (reference src/lib.rs:9-21) for everything on the path:
Scene, Object, Light, Material, Camera, Renderer, Buffer, Filter, Environment, Hdri,
sphere, plane, cube, polygon, Mesh, KdTree, Triangle, Transformed, hex_color, color_bytes.
The path tracer itself is HIP (rpt_amd/csrc) behind the C ABI in include/rpt_gpu.h.
"""
from . import glm  # noqa: F401
from ._abi import RptGpuError  # noqa: F401
from .buffer import Buffer, Filter  # noqa: F401
from .camera import Camera  # noqa: F401
from .color import color_bytes, hex_color  # noqa: F401
from .device import DeviceBuffer, GpuScene, device_count, make_params  # noqa: F401
from .environment import Environment, Hdri  # noqa: F401
from .io import load_mtl, load_obj, load_obj_with_mtl, load_stl  # noqa: F401
from .light import Light  # noqa: F401
from .material import Material  # noqa: F401
from .object import Object  # noqa: F401
from .renderer import Renderer  # noqa: F401
from .scene import Scene  # noqa: F401
from . import scenes  # noqa: F401
from .shape import (Cube, KdTree, Mesh, MonomialSurface, Plane, Shape, Sphere, Transformed, Triangle,  # noqa: F401
                    cube, monomial_surface, plane, polygon, sphere)
