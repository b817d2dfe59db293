"""The scenes of rust/rpt_additions/dump_golden.rs (the patched reference's golden-vector dump), built with the host
mirror: what scripts/compare_rust_golden.py and tests/test_rust_golden.py render to compare with the Rust program's
frames.  Keep the two files in step — same objects, same order, same constants."""
import numpy as np

from . import scenes
from .camera import Camera
from .environment import Environment
from .color import hex_color
from .light import Light
from .material import Material
from .object import Object
from .scene import Scene
from .shape import Mesh, plane, sphere

NAMES = ("sphere", "cornell", "teapot")


def teapot_golden(triangles):
    """dump_golden.rs `teapot`: examples/teapot.rs's mesh and placement, a glass sphere, a plane, ambient + point light,
    a constant sky.  `triangles`: (n, 18) rows of the crate's examples/teapot.obj as load_obj parses it."""
    scene = Scene()
    scene.add(Object(Mesh(np.asarray(triangles, dtype=np.float64)).scale((0.5, 0.5, 0.5)).translate((0.0, -1.0, 0.0)))
              .material(Material.metallic_(hex_color(0xFF0000), 0.4)))
    scene.add(Object(sphere().scale((0.4, 0.4, 0.4)).translate((1.3, -0.6, 0.8))).material(Material.clear(1.5, 0.02)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((60.0, 60.0, 60.0), (0.0, 5.0, 5.0)))
    scene.environment = Environment.Color((0.3, 0.4, 0.6))
    return scene, Camera()


def build(name, triangles=None):
    if name == "sphere":
        s, c, _ = scenes.sphere_scene()
        return s, c
    if name == "cornell":
        s, c, _ = scenes.cornell()
        return s, c
    if name == "teapot":
        if triangles is None:
            raise ValueError("the teapot golden scene needs the parsed triangles of examples/teapot.obj")
        return teapot_golden(triangles)
    raise KeyError(name)
