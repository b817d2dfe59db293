"""`Object` (reference src/object.rs:10-31): one shape, one material, builder style."""
from . import _abi
from .material import Material


class Object:
    def __init__(self, shape):  # Object::new, object.rs:20-25 (default material)
        self.shape = shape
        self._material = Material()

    def material(self, material):  # object.rs:28-31
        self._material = material
        return self

    def lower_into(self, out, keep):
        self.shape._fill(out.shape, keep)
        out.material = self._material.lower()
