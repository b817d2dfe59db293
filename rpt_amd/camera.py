"""`Camera` (reference src/camera.rs:8-62).  `cast_ray` (camera.rs:64-81) runs on the device
(ray-generation kernel)."""
import math

from . import _abi, glm


class Camera:
    def __init__(self, eye=(0.0, 0.0, 10.0), direction=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0),
                 fov=math.pi / 6.0, aperture=0.0, focal_distance=0.0):  # Default camera.rs:28-39
        self.eye, self.direction, self.up = glm.vec3(*eye), glm.vec3(*direction), glm.vec3(*up)
        self.fov, self.aperture, self.focal_distance = float(fov), float(aperture), float(focal_distance)

    @staticmethod
    def look_at(eye, center, up, fov):  # camera.rs:43-54
        eye, center, up = glm.vec3(*eye), glm.vec3(*center), glm.vec3(*up)
        direction = glm.normalize(glm.sub(center, eye))
        up = glm.normalize(glm.sub(up, glm.scale(direction, glm.dot(up, direction))))
        return Camera(eye, direction, up, fov, 0.0, 0.0)

    def focus(self, focal_point, aperture):  # camera.rs:57-61
        self.focal_distance = glm.dot(glm.sub(glm.vec3(*focal_point), self.eye), self.direction)
        self.aperture = float(aperture)
        return self

    def lower(self):
        c = _abi.RptCamera()
        c.eye[:], c.direction[:], c.up[:] = self.eye, self.direction, self.up
        c.fov, c.aperture, c.focal_distance = self.fov, self.aperture, self.focal_distance
        return c
