"""Thin object wrapper over the C ABI (include/rpt_gpu.h): scene upload, the render-batch
hot path, the closest-hit kernel and the accounting.  Everything here calls into
`librptgpu.so`; nothing is computed in Python."""
import ctypes as C

import numpy as np

from . import _abi


def make_params(width, height, max_bounces, iterations, exposure_value=0.0, seed=0x52505447,
                sample_index_base=0, tile=(32, 8), part=(0, 1),
                precision=_abi.RPT_PRECISION_F64_STRICT, flags=0, collective=_abi.RPT_COLLECTIVE_DEFAULT):
    p = _abi.RptRenderParams()
    p.collective = int(collective)
    p.width, p.height, p.max_bounces, p.iterations = int(width), int(height), int(max_bounces), int(iterations)
    p.exposure_value = float(exposure_value)
    p.seed, p.sample_index_base = int(seed), int(sample_index_base)
    p.tile_width, p.tile_height = int(tile[0]), int(tile[1])
    p.part_index, p.part_count = int(part[0]), int(part[1])
    p.precision_mode, p.flags = int(precision), int(flags)
    return p


def scene_options(**fields):
    """RptSceneOptions with the library's defaults (rptgpu_scene_options_default) and the given fields set."""
    o = _abi.RptSceneOptions()
    _abi.load_library().rptgpu_scene_options_default(C.byref(o))
    known = {name for name, _ in _abi.RptSceneOptions._fields_}
    for k, v in fields.items():
        if k not in known or k in ("struct_size", "_reserved0"):
            raise TypeError("RptSceneOptions has no field %r" % k)
        setattr(o, k, v)
    return o


def device_count():
    lib = _abi.load_library()
    n = C.c_int(0)
    _abi.check(lib.rptgpu_device_count(C.byref(n)))
    return n.value


class GpuScene:
    """Owns one `rptgpu_scene*` (device-resident flattened scene + kd-trees)."""

    def __init__(self, scene, device=0, **options):
        """options: fields of RptSceneOptions (include/rpt_gpu.h) by name, e.g. GpuScene(scene, 0, sort_rays=0,
        deep_depth=1); the rest keep their defaults.  RPTGPU_* environment variables still override."""
        self.lib = _abi.load_library()
        desc, keep = scene.lower()
        h = C.c_void_p()
        if options:
            o = scene_options(**options)
            _abi.check(self.lib.rptgpu_scene_create_opts(C.byref(desc), int(device), C.byref(o), C.byref(h)))
        else:
            _abi.check(self.lib.rptgpu_scene_create(C.byref(desc), int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def options(self):
        """The options the handle runs with (defaults, the caller's, environment overrides) as a dict."""
        o = _abi.RptSceneOptions()
        o.struct_size = C.sizeof(_abi.RptSceneOptions)  # ABI v7: the caller says how large ITS struct is
        _abi.check(self.lib.rptgpu_scene_get_options(self.handle, C.byref(o)), self.handle)
        return {name: getattr(o, name) for name, _ in _abi.RptSceneOptions._fields_ if not name.startswith("_")}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rptgpu_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_batch(self, camera, params):
        """Renderer::sample's output: (H*W, 3) float64 means, row-major, top row first."""
        out = np.empty((params.height * params.width, 3), dtype=np.float64)
        cam = camera.lower() if hasattr(camera, "lower") else camera
        code = self.lib.rptgpu_render_batch(self.handle, C.byref(cam), C.byref(params),
                                            out.ctypes.data_as(C.POINTER(C.c_double)))
        _abi.check(code, self.handle)
        return out

    def render_batch_device(self, camera, params, d_ptr, out_is_f32=False, stream=None):
        cam = camera.lower() if hasattr(camera, "lower") else camera
        code = self.lib.rptgpu_render_batch_device(self.handle, C.byref(cam), C.byref(params),
                                                   C.c_void_p(d_ptr), 1 if out_is_f32 else 0,
                                                   C.c_void_p(stream or 0))
        _abi.check(code, self.handle)

    # ---- multi-GPU: the library's own RCCL communicator (include/rpt_gpu.h) ----
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte id every rank passes to comm_init (hand it over any side channel)"""
        lib = _abi.load_library()
        buf = (C.c_uint8 * _abi.RPTGPU_UNIQUE_ID_BYTES)()
        _abi.check(lib.rptgpu_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * _abi.RPTGPU_UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _abi.check(self.lib.rptgpu_comm_init(self.handle, int(rank), int(world), buf), self.handle)

    def comm_destroy(self):
        _abi.check(self.lib.rptgpu_comm_destroy(self.handle), self.handle)

    def render_batch_reduce(self, camera, params, root=0, out=None):
        """Renderer::sample on every rank: this rank's tiles, the owned pixels gathered on `root` over RCCL
        (params.collective = RPT_COLLECTIVE_REDUCE: ncclReduce(sum) of zero-filled frames), result in host memory on root.  `out`: float32 array of width*height*3 (allocated if None on root)."""
        cam = camera.lower() if hasattr(camera, "lower") else camera
        if out is None:
            out = np.empty(params.height * params.width * 3, dtype=np.float32)
        code = self.lib.rptgpu_render_batch_reduce(self.handle, C.byref(cam), C.byref(params), int(root),
                                                   out.ctypes.data_as(C.POINTER(C.c_float)))
        _abi.check(code, self.handle)
        return out

    def render_batch_emulate_ranks(self, camera, params, world, out=None):
        """Diagnostics: the frame as `world` ranks' gather would assemble it, on this one GPU (include/rpt_gpu.h)."""
        cam = camera.lower() if hasattr(camera, "lower") else camera
        if out is None:
            out = np.empty(params.height * params.width * 3, dtype=np.float32)
        code = self.lib.rptgpu_render_batch_emulate_ranks(self.handle, C.byref(cam), C.byref(params), int(world),
                                                          out.ctypes.data_as(C.POINTER(C.c_float)))
        _abi.check(code, self.handle)
        return out

    def closest_hit(self, origins, dirs, precision=_abi.RPT_PRECISION_F64_STRICT):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, dtype=np.float64)
        nrm = np.empty((n, 3), dtype=np.float64)
        obj = np.empty(n, dtype=np.int32)
        PD = C.POINTER(C.c_double)
        code = self.lib.rptgpu_closest_hit(self.handle, n, o.ctypes.data_as(PD), d.ctypes.data_as(PD),
                                           int(precision), t.ctypes.data_as(PD), nrm.ctypes.data_as(PD),
                                           obj.ctypes.data_as(C.POINTER(C.c_int32)))
        _abi.check(code, self.handle)
        return t, nrm, obj

    def eval_math(self, fn, x, y=None):
        """include/rpt_math.h evaluated on the device (diagnostics)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        yy = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
        out = np.empty_like(x)
        PD = C.POINTER(C.c_double)
        code = self.lib.rptgpu_eval_math(self.handle, int(fn), x.size, x.ctypes.data_as(PD),
                                         yy.ctypes.data_as(PD) if yy is not None else None, out.ctypes.data_as(PD))
        _abi.check(code, self.handle)
        return out

    def stats(self):
        s = _abi.RptStats()
        _abi.check(self.lib.rptgpu_get_stats(self.handle, C.byref(s)), self.handle)
        return s

    def reset_stats(self):
        _abi.check(self.lib.rptgpu_reset_stats(self.handle), self.handle)


def kdtree_build(boxes, lib=None, prefix="rptgpu", device=None):
    """KdTree::new over (n, 6) boxes through the C ABI -> dict of numpy arrays.  device: build on that HIP device
    (rptgpu_kdtree_build_device) instead of the host."""
    lib = lib or _abi.load_library()
    b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    t = _abi.RptKdTree()
    free = getattr(lib, prefix + "_kdtree_free")
    if device is None:
        code = getattr(lib, prefix + "_kdtree_build")(b.ctypes.data_as(C.POINTER(C.c_double)), len(b), C.byref(t))
    else:
        code = lib.rptgpu_kdtree_build_device(b.ctypes.data_as(C.POINTER(C.c_double)), len(b), int(device), C.byref(t))
    if code != 0:
        detail = lib.rptgpu_last_error_detail(None) if prefix == "rptgpu" else b""
        raise _abi.RptGpuError(code, "kdtree_build: " + (detail.decode() if detail else ""))
    n, r = t.num_nodes, t.num_refs
    out = {
        "split": np.ctypeslib.as_array(t.split, (n,)).copy(),
        "info": np.ctypeslib.as_array(t.info, (n,)).copy(),
        "a": np.ctypeslib.as_array(t.a, (n,)).copy(),
        "b": np.ctypeslib.as_array(t.b, (n,)).copy(),
        "refs": np.ctypeslib.as_array(t.refs, (max(r, 1),)).copy()[:r],
        "max_depth": t.max_depth,
        "regular": t.regular,
    }
    free(C.byref(t))
    return out


class DeviceBuffer:
    """`Buffer` (reference src/buffer.rs) kept on the GPU: `rptgpu_buffer_*` of include/rpt_gpu.h.
    Same results as the host `rpt_amd.Buffer`, without moving any batch to the host."""

    def __init__(self, gpu_scene, width, height, filter=None):
        self.gpu = gpu_scene
        self.width, self.height = int(width), int(height)
        radius = filter.radius if filter is not None else 0
        h = C.c_void_p()
        _abi.check(gpu_scene.lib.rptgpu_buffer_create(gpu_scene.handle, self.width, self.height, int(radius), C.byref(h)),
                   gpu_scene.handle)
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.gpu.lib.rptgpu_buffer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sample(self, camera, params):
        """Renderer::sample + Buffer::add_samples on the device."""
        cam = camera.lower() if hasattr(camera, "lower") else camera
        _abi.check(self.gpu.lib.rptgpu_buffer_sample(self.handle, C.byref(cam), C.byref(params)), self.gpu.handle)

    def image(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        _abi.check(self.gpu.lib.rptgpu_buffer_image(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))), self.gpu.handle)
        return out

    def variance(self):
        v = C.c_double(0.0)
        _abi.check(self.gpu.lib.rptgpu_buffer_variance(self.handle, C.byref(v)), self.gpu.handle)
        return v.value

    def num_batches(self):
        n = C.c_uint32(0)
        _abi.check(self.gpu.lib.rptgpu_buffer_num_batches(self.handle, C.byref(n)), self.gpu.handle)
        return n.value
