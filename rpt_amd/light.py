"""`Light` (reference src/light.rs:7-19).  `illuminate` (light.rs:23-47) runs on the device."""
from . import _abi, glm


class Light:
    def __init__(self, kind, color=(0.0, 0.0, 0.0), vec=(0.0, 0.0, 0.0), obj=None):
        self.kind, self.color, self.vec, self.object = kind, glm.vec3(*color), glm.vec3(*vec), obj

    @staticmethod
    def Point(color, location):
        return Light(_abi.RPT_LIGHT_POINT, color, location)

    @staticmethod
    def Ambient(color):
        return Light(_abi.RPT_LIGHT_AMBIENT, color)

    @staticmethod
    def Directional(color, direction):
        return Light(_abi.RPT_LIGHT_DIRECTIONAL, color, direction)

    @staticmethod
    def Object(obj):
        return Light(_abi.RPT_LIGHT_OBJECT, obj=obj)

    def lower_into(self, out, keep):
        out.kind = self.kind
        out.color[:] = self.color
        out.vec[:] = self.vec
        if self.kind == _abi.RPT_LIGHT_OBJECT:
            self.object.lower_into(out.object, keep)
