"""`Scene` / `SceneAdd` (reference src/scene.rs:7-41): two lists and an environment."""
from . import _abi
from .environment import Environment
from .light import Light
from .object import Object


class Scene:
    def __init__(self):  # Scene::new, scene.rs:20-23
        self.objects = []
        self.lights = []
        self.environment = Environment()

    def add(self, node):  # SceneAdd<Object> / SceneAdd<Light>, scene.rs:31-41
        if isinstance(node, Object):
            self.objects.append(node)
        elif isinstance(node, Light):
            self.lights.append(node)
        else:
            raise TypeError("Scene.add takes an Object or a Light")

    def lower(self):
        """-> (RptScene, keepalive list)"""
        keep = []
        s = _abi.RptScene()
        objs = (_abi.RptObject * max(1, len(self.objects)))()
        for i, o in enumerate(self.objects):
            o.lower_into(objs[i], keep)
        lights = (_abi.RptLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            l.lower_into(lights[i], keep)
        keep += [objs, lights]
        s.objects, s.num_objects = objs, len(self.objects)
        s.lights, s.num_lights = lights, len(self.lights)
        self.environment.lower_into(s.environment, keep)
        return s, keep
