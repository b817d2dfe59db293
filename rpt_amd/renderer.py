"""`Renderer` (reference src/renderer.rs:18-115): the builder and the two public entry
points.  `sample()` — the seam of SURVEY §8b — is one call into the C ABI
(`rptgpu_render_batch`); nothing of the path tracer is computed in Python."""
import math

from . import _abi
from .buffer import Buffer, Filter
from .device import DeviceBuffer, GpuScene, make_params


class Renderer:
    def __init__(self, scene, camera):  # Renderer::new, renderer.rs:46-57
        self.scene = scene
        self.camera = camera
        self._width = 800
        self._height = 600
        self._exposure_value = 0.0
        self._filter = Filter()
        self._max_bounces = 0
        self._num_samples = 1
        # additions (the reference has no seed and no device): see include/rpt_gpu.h
        self._seed = 0x52505447
        self._device = 0
        self._precision = _abi.RPT_PRECISION_F64_STRICT
        self._samples_done = 0
        self._gpu = None

    def width(self, width):  # renderer.rs:60-63
        self._width = int(width)
        return self

    def height(self, height):  # renderer.rs:66-69
        self._height = int(height)
        return self

    def exposure_value(self, ev):  # renderer.rs:72-75
        self._exposure_value = float(ev)
        return self

    def filter(self, filter):  # renderer.rs:78-81
        self._filter = filter
        return self

    def max_bounces(self, n):  # renderer.rs:84-87
        self._max_bounces = int(n)
        return self

    def num_samples(self, n):  # renderer.rs:90-93
        self._num_samples = int(n)
        return self

    def seed(self, seed):
        self._seed = int(seed)
        return self

    def device(self, device):
        self._device = int(device)
        return self

    def precision(self, mode):
        self._precision = int(mode)
        return self

    def gpu_scene(self):
        if self._gpu is None:
            self._gpu = GpuScene(self.scene, self._device)
        return self._gpu

    def render(self):  # renderer.rs:96-100 -> (H, W, 3) uint8
        buffer = Buffer(self._width, self._height, self._filter)
        self._samples_done = 0
        self.sample(self._num_samples, buffer)
        return buffer.image()

    def iterative_render(self, callback_interval, callback, on_device=False):  # renderer.rs:103-115
        """`on_device=True` keeps the Buffer on the GPU (rptgpu_buffer_*): the callback receives a
        DeviceBuffer with the same image() / variance() and no batch ever travels to the host."""
        if on_device:
            buffer = DeviceBuffer(self.gpu_scene(), self._width, self._height, self._filter)
            iteration = 0
            self._samples_done = 0
            while iteration < self._num_samples:
                steps = min(self._num_samples - iteration, int(callback_interval))
                params = make_params(self._width, self._height, self._max_bounces, steps, self._exposure_value,
                                     self._seed, self._samples_done, precision=self._precision)
                buffer.sample(self.camera, params)
                self._samples_done += steps
                iteration += steps
                callback(iteration, buffer)
            buffer.close()
            return
        buffer = Buffer(self._width, self._height, self._filter)
        iteration = 0
        self._samples_done = 0
        while iteration < self._num_samples:
            steps = min(self._num_samples - iteration, int(callback_interval))
            self.sample(steps, buffer)
            iteration += steps
            callback(iteration, buffer)

    def sample(self, iterations, buffer):  # renderer.rs:117-129 — THE hot path, on the GPU
        params = make_params(self._width, self._height, self._max_bounces, iterations,
                             self._exposure_value, self._seed, self._samples_done,
                             precision=self._precision)
        colors = self.gpu_scene().render_batch(self.camera, params)
        self._samples_done += iterations
        buffer.add_samples(colors)
