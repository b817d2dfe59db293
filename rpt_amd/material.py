"""`Material` (reference src/material.rs:8-105): the six fields and the constructors.
`bsdf` / `sample_f` (material.rs:125-313) run only on the device (shade kernel)."""
from . import _abi
from .color import hex_color


class Material:
    def __init__(self, color=None, index=1.5, roughness=0.5, metallic=0.0, emittance=0.0,
                 transparent=False):
        # Default: specular(hex_color(0xff0000), 0.5), material.rs:28-32
        self.color = tuple(float(c) for c in (color if color is not None else hex_color(0xFF0000)))
        self.index = float(index)
        self.roughness = float(roughness)
        self.metallic = float(metallic)
        self.emittance = float(emittance)
        self.transparent = bool(transparent)

    @staticmethod
    def diffuse(color):  # material.rs:36-45
        return Material(color, 1.5, 1.0, 0.0, 0.0, False)

    @staticmethod
    def specular(color, roughness):  # material.rs:48-57
        return Material(color, 1.5, roughness, 0.0, 0.0, False)

    @staticmethod
    def clear(index, roughness):  # material.rs:60-69
        return Material((1.0, 1.0, 1.0), index, roughness, 0.0, 0.0, True)

    @staticmethod
    def transparent_(color, index, roughness):  # material.rs:72-81 (`transparent`)
        return Material(color, index, roughness, 0.0, 0.0, True)

    @staticmethod
    def metallic_(color, roughness):  # material.rs:84-93 (`metallic`)
        return Material(color, 1.5, roughness, 1.0, 0.0, False)

    @staticmethod
    def light(color, emittance):  # material.rs:96-105
        return Material(color, 1.0, 1.0, 0.0, emittance, False)

    def lower(self):
        m = _abi.RptMaterial()
        m.color[:] = self.color
        m.index, m.roughness, m.metallic = self.index, self.roughness, self.metallic
        m.emittance = self.emittance
        m.transparent = 1 if self.transparent else 0
        return m
