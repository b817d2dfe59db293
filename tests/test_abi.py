"""The C-ABI library loads on a CPU-only box and exports every symbol include/rpt_gpu.h declares;
host-side validation works without a GPU; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rpt_amd
from rpt_amd import _abi
from rpt_amd.shape import Triangle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rpt_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rptgpu_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == {s[0] for s in _abi.SYMBOLS}
    assert lib.rptgpu_abi_version() == _abi.ABI_VERSION == 7


def test_struct_sizes_match_header(tmp_path):
    # compile the header with gcc (as C) and compare sizeof of every struct with the ctypes mirror
    names = ["RptMaterial", "RptTriangle", "RptTransform", "RptShape", "RptObject", "RptLight",
             "RptEnvironment", "RptScene", "RptCamera", "RptRenderParams", "RptSceneOptions", "RptStats", "RptKdTree"]
    src = '#include <stdio.h>\n#include "rpt_gpu.h"\nint main(void){' + "".join(
        'printf("%%zu\\n", sizeof(%s));' % n for n in names) + "return 0;}"
    c = tmp_path / "sz.c"
    c.write_text(src)
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    for n, sz in zip(names, sizes):
        assert C.sizeof(getattr(_abi, n)) == sz, n


def test_scene_options_defaults_validation_and_environment_overrides(monkeypatch):
    """RptSceneOptions (ABI v6 / v7): the library fills the defaults, rejects a struct of a size it does not know and fields out
    of range BEFORE anything else is looked at (no GPU needed), and takes a smaller struct of an older header."""
    lib = _abi.load_library()
    o = _abi.RptSceneOptions()
    lib.rptgpu_scene_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_abi.RptSceneOptions)
    assert (o.deep_depth, o.fast_max_depth, o.sort_rays, o.rays_in_kernel) == (8, 32, -1, 0)
    assert (o.sort_min_bytes, o.sort_shadow_min_bytes, o.sort_min_rays) == (8 << 20, 8 << 20, 1 << 19)
    assert (o.nest_trace, o.leaf_boxes, o.object_filter_min, o.device_build_min) == (1, 1, 5, 32768)
    assert (o.build_threads, o.paths_chunk, o.workspace_bytes, o.lbuf_bytes, o.target_paths) == (0, 0, 240 << 30, 32 << 30, 0)
    assert o.comm_timeout_s == 300.0 and o.env_park == 1 and o.paths_batch == 0
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Object(rpt_amd.sphere()))
    desc, keep = scene.lower()
    h = C.c_void_p()

    def create(opts):
        return lib.rptgpu_scene_create_opts(C.byref(desc), 0, C.byref(opts) if opts is not None else None, C.byref(h))
    assert create(None) == _abi.RPTGPU_E_NO_DEVICE  # valid options: the flattening runs, then the device is missing
    assert create(o) == _abi.RPTGPU_E_NO_DEVICE
    bad = rpt_amd.device.scene_options()
    bad.struct_size = C.sizeof(_abi.RptSceneOptions) + 8
    assert create(bad) == _abi.RPTGPU_E_INVALID_ARGUMENT and b"struct_size" in lib.rptgpu_last_error_detail(None)
    for field, value in (("sort_rays", 2), ("deep_depth", 0), ("lbuf_bytes", 8), ("workspace_bytes", 1000),
                         ("comm_timeout_s", 0.0), ("target_paths", 5), ("paths_batch", 1025)):
        bad = rpt_amd.device.scene_options(**{field: value})
        assert create(bad) == _abi.RPTGPU_E_INVALID_ARGUMENT, field
    for cut in (8, 24, 100, 108):  # only WHOLE structs of a header that existed: a size in between cuts a field in half
        old = rpt_amd.device.scene_options(sort_rays=0)
        old.struct_size = cut
        assert create(old) == _abi.RPTGPU_E_INVALID_ARGUMENT, cut
    old = rpt_amd.device.scene_options(paths_chunk=4)
    old.struct_size = 104  # the first v6 header, before env_park
    assert create(old) == _abi.RPTGPU_E_NO_DEVICE
    with pytest.raises(TypeError):
        rpt_amd.device.scene_options(no_such_field=1)
    assert lib.rptgpu_scene_get_options(None, C.byref(o)) == _abi.RPTGPU_E_INVALID_ARGUMENT
    # the environment's overrides are held to the fields' ranges too; RPTGPU_SORT_RAYS=-1 IS the documented default
    monkeypatch.setenv("RPTGPU_DEEP_DEPTH", "3")
    monkeypatch.setenv("RPTGPU_SORT_RAYS", "-1")
    assert create(None) == _abi.RPTGPU_E_NO_DEVICE
    monkeypatch.setenv("RPTGPU_LBUF_BYTES", "24")
    monkeypatch.setenv("RPTGPU_COMM_TIMEOUT_S", "-5")  # (ignored: not a positive number)
    assert create(None) == _abi.RPTGPU_E_NO_DEVICE


def test_sized_option_access_never_writes_past_the_callers_struct():
    """a caller built against the 104-byte first v6 header must not get 112 bytes written into its
    struct.  rptgpu_scene_options_default_sized writes exactly the size it is told and knows only the sizes the struct
    has had; (rptgpu_scene_get_options does the same with out->struct_size — needs a handle: tests/test_gpu_parity.py)."""
    lib = _abi.load_library()
    full = C.sizeof(_abi.RptSceneOptions)
    assert full == 112
    for size in (104, 112):
        raw = (C.c_uint8 * (full + 16))(*([0xAB] * (full + 16)))
        o = C.cast(raw, C.POINTER(_abi.RptSceneOptions))
        assert lib.rptgpu_scene_options_default_sized(o, size) == _abi.RPTGPU_OK
        assert o.contents.struct_size == size and o.contents.deep_depth == 8 and o.contents.comm_timeout_s == 300.0
        assert all(b == 0xAB for b in raw[size:]), size  # nothing behind the caller's struct was touched
        if size == 112:
            assert o.contents.env_park == 1 and o.contents.paths_batch == 0
    for size in (0, 8, 100, 108, 120):
        raw = (C.c_uint8 * (full + 16))(*([0xAB] * (full + 16)))
        assert lib.rptgpu_scene_options_default_sized(C.cast(raw, C.POINTER(_abi.RptSceneOptions)), size) == _abi.RPTGPU_E_INVALID_ARGUMENT
        assert all(b == 0xAB for b in raw)
    assert lib.rptgpu_scene_options_default_sized(None, 112) == _abi.RPTGPU_E_INVALID_ARGUMENT


def test_strerror_and_kernel_names():
    lib = _abi.load_library()
    assert lib.rptgpu_strerror(0) == b"ok"
    assert b"device" in lib.rptgpu_strerror(_abi.RPTGPU_E_NO_DEVICE)
    assert lib.rptgpu_kernel_name(_abi.RPT_K_EXTEND) == b"rpt_extend"
    assert lib.rptgpu_kernel_name(_abi.RPT_K_SHADE) == b"rpt_shade"


def test_unsupported_shapes_rejected_at_scene_create_without_gpu():
    # validation happens on the host before the GPU is touched (SURVEY §8b error convention)
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Light.Object(rpt_amd.Object(rpt_amd.plane((0, 1, 0), 0.0))))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_UNIMPLEMENTED_SAMPLE  # plane.rs:34-36 unimplemented!()

    # groups inside groups to ANY depth, as in the reference (kdtree.rs:14-24): flattening succeeds, only the device is missing
    inner = rpt_amd.KdTree([rpt_amd.sphere().translate((0, 0, 0))])
    g = inner
    for _ in range(12):
        g = rpt_amd.KdTree([g, rpt_amd.sphere(), rpt_amd.cube()])
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Object(g))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_NO_DEVICE
    # ... except as the shape of a Light::Object: Shape::sample keeps its chain of groups in registers, eight levels
    lamp = inner
    for _ in range(7):
        lamp = rpt_amd.KdTree([lamp, rpt_amd.sphere()])
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Light.Object(rpt_amd.Object(lamp)))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_NO_DEVICE
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Light.Object(rpt_amd.Object(rpt_amd.KdTree([lamp, rpt_amd.cube()]))))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_UNSUPPORTED_SHAPE and "Light::Object" in str(e.value)

    scene = rpt_amd.Scene()  # MonomialSurface: only exp = 4 (monomial_surface.rs:10), at top level or as a tree child
    scene.add(rpt_amd.Object(rpt_amd.monomial_surface(1.0, 3.0)))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_UNSUPPORTED_SHAPE
    scene = rpt_amd.Scene()
    scene.add(rpt_amd.Object(rpt_amd.KdTree([rpt_amd.monomial_surface(1.0, 4.0), rpt_amd.sphere()])))
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_NO_DEVICE
    with pytest.raises(rpt_amd.RptGpuError):
        rpt_amd.KdTree([rpt_amd.plane((0, 1, 0), 0.0), rpt_amd.sphere()]).lower([])


def test_no_cpu_fallback(gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    scene, camera, _ = rpt_amd.scenes.sphere_scene()
    with pytest.raises(rpt_amd.RptGpuError) as e:
        rpt_amd.GpuScene(scene)
    assert e.value.code == _abi.RPTGPU_E_NO_DEVICE
    with pytest.raises(rpt_amd.RptGpuError):
        rpt_amd.Renderer(scene, camera).width(8).height(8).render()


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(ImportError):
        _abi.load_library(str(tmp_path / "nope.so"))


def test_null_arguments_return_error_codes():
    lib = _abi.load_library()
    assert lib.rptgpu_scene_create(None, 0, None) == _abi.RPTGPU_E_INVALID_ARGUMENT
    assert lib.rptgpu_render_batch(None, None, None, None) == _abi.RPTGPU_E_INVALID_ARGUMENT
    assert lib.rptgpu_get_stats(None, None) == _abi.RPTGPU_E_INVALID_ARGUMENT
    assert lib.rptgpu_kdtree_build(None, 5, None) == _abi.RPTGPU_E_INVALID_ARGUMENT
    lib.rptgpu_scene_destroy(None)  # no-op
