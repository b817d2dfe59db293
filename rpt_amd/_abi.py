"""ctypes mirror of include/rpt_gpu.h and the loader of the product library.

The product library (`rpt_amd/lib/librptgpu.so`, built by `__graft_entry__.build()` with
hipcc for gfx950) is the ONLY compute path: there is no CPU fallback here.  If the library
is missing, loading raises; if it loads but no GPU is present, every compute entry point
returns RPTGPU_E_NO_DEVICE which is raised as `RptGpuError`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RPTGPU_LIB") or os.path.join(HERE, "lib", "librptgpu.so")  # RPTGPU_LIB: dev A/B builds

ABI_VERSION = 7

RPTGPU_OK = 0
RPTGPU_E_INVALID_ARGUMENT = -1
RPTGPU_E_UNSUPPORTED_SHAPE = -2
RPTGPU_E_NO_DEVICE = -3
RPTGPU_E_HIP = -4
RPTGPU_E_OUT_OF_MEMORY = -5
RPTGPU_E_TREE_TOO_DEEP = -6
RPTGPU_E_UNIMPLEMENTED_SAMPLE = -7
RPTGPU_E_COMM = -8
RPTGPU_UNIQUE_ID_BYTES = 128

RPT_SHAPE_SPHERE, RPT_SHAPE_PLANE, RPT_SHAPE_CUBE, RPT_SHAPE_MESH, RPT_SHAPE_GROUP, RPT_SHAPE_MONOMIAL = range(6)
RPT_LIGHT_POINT, RPT_LIGHT_AMBIENT, RPT_LIGHT_DIRECTIONAL, RPT_LIGHT_OBJECT = range(4)
RPT_ENV_COLOR, RPT_ENV_HDRI = range(2)
RPT_PRECISION_F64_STRICT = 0  # the only arithmetic mode (ABI v4 removed F64_FAST)
RPT_FLAG_PROFILE_KERNELS = 1
RPT_FLAG_WAVEFRONT = 2
RPT_FLAG_GENERAL_TRAVERSAL = 4
RPT_FLAG_PERSISTENT = 8
RPT_K_RAYGEN, RPT_K_EXTEND, RPT_K_SHADE, RPT_K_SHADOW, RPT_K_RESOLVE, RPT_K_PATHS, RPT_K_TREE_TRACE, RPT_K_TREE_SORT = range(8)
RPT_K_COUNT = 8

f64 = C.c_double
V3 = f64 * 3


class RptMaterial(C.Structure):
    _fields_ = [("color", V3), ("index", f64), ("roughness", f64), ("metallic", f64),
                ("emittance", f64), ("transparent", C.c_int32), ("_pad", C.c_int32)]


class RptTriangle(C.Structure):
    _fields_ = [("v1", V3), ("v2", V3), ("v3", V3), ("n1", V3), ("n2", V3), ("n3", V3)]


class RptTransform(C.Structure):
    _fields_ = [("transform", f64 * 16), ("linear", f64 * 9), ("inverse_transform", f64 * 16),
                ("normal_transform", f64 * 9), ("scale", f64)]


class RptShape(C.Structure):
    pass


RptShape._fields_ = [("kind", C.c_int32), ("transformed", C.c_int32), ("xf", RptTransform),
                     ("plane_normal", V3), ("plane_value", f64),
                     ("monomial_height", f64), ("monomial_exp", f64),
                     ("triangles", C.POINTER(RptTriangle)), ("num_triangles", C.c_uint64),
                     ("children", C.POINTER(RptShape)), ("num_children", C.c_uint64)]


class RptObject(C.Structure):
    _fields_ = [("shape", RptShape), ("material", RptMaterial)]


class RptLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("color", V3), ("vec", V3),
                ("object", RptObject)]


class RptEnvironment(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("color", V3), ("width", C.c_uint32),
                ("height", C.c_uint32), ("texels", C.POINTER(f64))]


class RptScene(C.Structure):
    _fields_ = [("objects", C.POINTER(RptObject)), ("num_objects", C.c_uint64),
                ("lights", C.POINTER(RptLight)), ("num_lights", C.c_uint64),
                ("environment", RptEnvironment)]


class RptCamera(C.Structure):
    _fields_ = [("eye", V3), ("direction", V3), ("up", V3), ("fov", f64), ("aperture", f64),
                ("focal_distance", f64)]


class RptRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_bounces", C.c_uint32),
                ("iterations", C.c_uint32), ("exposure_value", f64), ("seed", C.c_uint64),
                ("sample_index_base", C.c_uint64), ("tile_width", C.c_uint32),
                ("tile_height", C.c_uint32), ("part_index", C.c_uint32),
                ("part_count", C.c_uint32), ("precision_mode", C.c_uint32),
                ("flags", C.c_uint32), ("collective", C.c_uint32), ("_reserved0", C.c_uint32)]


RPT_COLLECTIVE_DEFAULT, RPT_COLLECTIVE_GATHER, RPT_COLLECTIVE_REDUCE = 0, 1, 2


class RptSceneOptions(C.Structure):
    """include/rpt_gpu.h RptSceneOptions (ABI v6, sized access v7): the knobs of a scene handle; rptgpu_scene_options_default fills it."""
    _fields_ = [("struct_size", C.c_uint32), ("_reserved0", C.c_uint32),
                ("deep_depth", C.c_uint32), ("fast_max_depth", C.c_uint32),
                ("sort_rays", C.c_int32), ("rays_in_kernel", C.c_int32),
                ("sort_min_bytes", C.c_uint64), ("sort_shadow_min_bytes", C.c_uint64),
                ("sort_min_rays", C.c_uint32), ("nest_trace", C.c_int32),
                ("leaf_boxes", C.c_int32), ("object_filter_min", C.c_int32),
                ("device_build_min", C.c_uint64),
                ("build_threads", C.c_uint32), ("paths_chunk", C.c_uint32),
                ("workspace_bytes", C.c_uint64), ("lbuf_bytes", C.c_uint64), ("target_paths", C.c_uint64),
                ("comm_timeout_s", f64),
                ("env_park", C.c_int32), ("paths_batch", C.c_uint32)]


class RptStats(C.Structure):
    _fields_ = [("kernel_ms", f64 * RPT_K_COUNT), ("kernel_launches", C.c_uint64 * RPT_K_COUNT),
                ("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shadow_rays_traced", C.c_uint64),
                ("samples", C.c_uint64), ("total_ms", f64),
                ("reduce_calls", C.c_uint64), ("reduce_render_ms", f64), ("reduce_collective_ms", f64),
                ("reduce_copy_ms", f64)]


class RptKdTree(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("num_refs", C.c_uint64), ("max_depth", C.c_uint32),
                ("regular", C.c_uint32), ("split", C.POINTER(f64)), ("info", C.POINTER(C.c_uint32)),
                ("a", C.POINTER(C.c_uint32)), ("b", C.POINTER(C.c_uint32)),
                ("refs", C.POINTER(C.c_uint32))]


# every symbol include/rpt_gpu.h declares: (name, restype, argtypes)
_VP = C.c_void_p
_PD = C.POINTER(f64)
SYMBOLS = [
    ("rptgpu_abi_version", C.c_int, []),
    ("rptgpu_strerror", C.c_char_p, [C.c_int]),
    ("rptgpu_last_error_detail", C.c_char_p, [_VP]),
    ("rptgpu_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("rptgpu_scene_create", C.c_int, [C.POINTER(RptScene), C.c_int, C.POINTER(_VP)]),
    ("rptgpu_scene_destroy", None, [_VP]),
    ("rptgpu_scene_options_default", None, [C.POINTER(RptSceneOptions)]),
    ("rptgpu_scene_options_default_sized", C.c_int, [C.POINTER(RptSceneOptions), C.c_uint32]),
    ("rptgpu_scene_create_opts", C.c_int, [C.POINTER(RptScene), C.c_int, C.POINTER(RptSceneOptions), C.POINTER(_VP)]),
    ("rptgpu_scene_get_options", C.c_int, [_VP, C.POINTER(RptSceneOptions)]),
    ("rptgpu_render_batch", C.c_int, [_VP, C.POINTER(RptCamera), C.POINTER(RptRenderParams), _PD]),
    ("rptgpu_render_batch_device", C.c_int,
     [_VP, C.POINTER(RptCamera), C.POINTER(RptRenderParams), _VP, C.c_int, _VP]),
    ("rptgpu_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("rptgpu_comm_init", C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    ("rptgpu_comm_destroy", C.c_int, [_VP]),
    ("rptgpu_render_batch_reduce", C.c_int,
     [_VP, C.POINTER(RptCamera), C.POINTER(RptRenderParams), C.c_int, C.POINTER(C.c_float)]),
    ("rptgpu_render_batch_emulate_ranks", C.c_int,
     [_VP, C.POINTER(RptCamera), C.POINTER(RptRenderParams), C.c_int, C.POINTER(C.c_float)]),
    ("rptgpu_closest_hit", C.c_int,
     [_VP, C.c_uint64, _PD, _PD, C.c_uint32, _PD, _PD, C.POINTER(C.c_int32)]),
    ("rptgpu_eval_math", C.c_int, [_VP, C.c_int, C.c_uint64, _PD, _PD, _PD]),
    ("rptgpu_kdtree_build", C.c_int, [_PD, C.c_uint64, C.POINTER(RptKdTree)]),
    ("rptgpu_kdtree_build_device", C.c_int, [_PD, C.c_uint64, C.c_int, C.POINTER(RptKdTree)]),
    ("rptgpu_kdtree_free", None, [C.POINTER(RptKdTree)]),
    ("rptgpu_buffer_create", C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    ("rptgpu_buffer_destroy", None, [_VP]),
    ("rptgpu_buffer_sample", C.c_int, [_VP, C.POINTER(RptCamera), C.POINTER(RptRenderParams)]),
    ("rptgpu_buffer_image", C.c_int, [_VP, C.POINTER(C.c_uint8)]),
    ("rptgpu_buffer_variance", C.c_int, [_VP, _PD]),
    ("rptgpu_buffer_num_batches", C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    ("rptgpu_get_stats", C.c_int, [_VP, C.POINTER(RptStats)]),
    ("rptgpu_reset_stats", C.c_int, [_VP]),
    ("rptgpu_kernel_name", C.c_char_p, [C.c_int]),
]


class RptGpuError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        self.detail = detail
        super().__init__("rptgpu error %d: %s" % (code, detail))


_lib = None


def load_library(path=None):
    """Load librptgpu.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            "rpt_amd: %s not found — the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rptgpu_abi_version() != ABI_VERSION:
        raise ImportError("rpt_amd: ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(code, handle=None):
    if code != RPTGPU_OK:
        lib = load_library()
        msg = lib.rptgpu_strerror(code).decode()
        det = lib.rptgpu_last_error_detail(handle)
        if det:
            msg += " — " + det.decode()
        raise RptGpuError(code, msg)
