"""The host mirror keeps rpt's builder API (reference src/renderer.rs:46-93, shape.rs:179-313,
material.rs:28-105, camera.rs:28-62, scene.rs:26-41): names, defaults, argument meaning."""
import math

import numpy as np

import rpt_amd
from rpt_amd import (Camera, Filter, Light, Material, Object, Renderer, Scene, cube, glm, hex_color,
                     plane, polygon, sphere)


def test_renderer_defaults_and_builder():
    scene, cam = Scene(), Camera()
    r = Renderer(scene, cam)  # renderer.rs:46-57
    assert (r._width, r._height, r._exposure_value, r._max_bounces, r._num_samples) == (800, 600, 0.0, 0, 1)
    assert r._filter.radius == 0
    r2 = r.width(64).height(32).exposure_value(1.5).filter(Filter.Box(1)).max_bounces(3).num_samples(7)
    assert r2 is r and (r._width, r._height, r._max_bounces, r._num_samples) == (64, 32, 3, 7)


def test_material_constructors():
    d = Material()  # default = specular(red, 0.5), material.rs:28-32
    assert d.color == hex_color(0xFF0000) and d.roughness == 0.5 and d.index == 1.5 and not d.transparent
    m = Material.diffuse((1, 2, 3))
    assert (m.index, m.roughness, m.metallic, m.emittance, m.transparent) == (1.5, 1.0, 0.0, 0.0, False)
    m = Material.clear(1.33, 0.01)
    assert m.color == (1.0, 1.0, 1.0) and m.index == 1.33 and m.transparent
    m = Material.metallic_((0.5, 0.5, 0.5), 0.2)
    assert m.metallic == 1.0 and m.roughness == 0.2
    m = Material.light((1, 1, 1), 40.0)
    assert (m.index, m.roughness, m.emittance) == (1.0, 1.0, 40.0)
    m = Material.transparent_((0.2, 0.3, 0.4), 1.4, 0.1)
    assert m.transparent and m.color == (0.2, 0.3, 0.4)


def test_scene_add_and_object_default_material():
    s = Scene()
    s.add(Object(sphere()))
    s.add(Light.Ambient((0.1, 0.1, 0.1)))
    assert len(s.objects) == 1 and len(s.lights) == 1
    assert s.objects[0]._material.color == hex_color(0xFF0000)
    assert s.environment.color == (0.0, 0.0, 0.0)
    try:
        s.add(42)
        assert False
    except TypeError:
        pass


def test_chained_transforms_compose_as_T_R_S():
    # shape.rs:234-284: cube().scale(s).rotate_y(a).translate(t) = T * R * S, no nesting
    s, a, t = (2.0, 3.0, 4.0), 0.7, (5.0, 6.0, 7.0)
    x = cube().scale(s).rotate_y(a).translate(t)
    assert isinstance(x.shape, rpt_amd.Cube)
    M = np.array(x.transform_m).reshape(4, 4).T
    S = np.diag([2.0, 3.0, 4.0, 1.0])
    R = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]])
    T = np.eye(4)
    T[:3, 3] = t
    assert np.allclose(M, T @ R @ S, atol=1e-15)
    Minv = np.array(x.inverse_transform).reshape(4, 4).T
    assert np.allclose(Minv @ M, np.eye(4), atol=1e-14)
    N = np.array(x.normal_transform).reshape(3, 3).T
    assert np.allclose(N, np.linalg.inv(M[:3, :3]).T, atol=1e-14)
    assert abs(x.scale_det - 24.0) < 1e-13
    assert np.allclose(np.array(glm.rotation(0.3, (0, 0, 2))).reshape(4, 4).T[:3, :3],
                       [[math.cos(0.3), -math.sin(0.3), 0], [math.sin(0.3), math.cos(0.3), 0], [0, 0, 1]])


def test_polygon_is_a_triangle_fan_with_face_normals():
    q = polygon([(0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 0)])  # shape.rs:307-313
    assert len(q) == 2
    assert np.allclose(q.triangles[0, 9:12], (0, 1, 0)) and np.allclose(q.triangles[1, :3], (0, 0, 0))


def test_lowering_roundtrip():
    scene, cam, cfg = rpt_amd.scenes.cornell()
    desc, keep = scene.lower()
    assert desc.num_objects == 7 and desc.num_lights == 1
    assert desc.objects[0].shape.kind == rpt_amd._abi.RPT_SHAPE_MESH and desc.objects[0].shape.num_triangles == 2
    assert desc.objects[5].shape.kind == rpt_amd._abi.RPT_SHAPE_CUBE and desc.objects[5].shape.transformed == 1
    assert desc.lights[0].kind == rpt_amd._abi.RPT_LIGHT_OBJECT and desc.lights[0].object.material.emittance == 100.0
    scene, cam, cfg = rpt_amd.scenes.fractal_spheres()
    desc, keep = scene.lower()
    assert [desc.objects[i].shape.num_children for i in range(5)] == [1, 6, 30, 150, 750]
    assert desc.objects[5].shape.kind == rpt_amd._abi.RPT_SHAPE_PLANE
    c = cam.lower()
    assert abs(sum(x * x for x in c.direction) - 1.0) < 1e-15


def test_scene_generators_are_deterministic():
    a = rpt_amd.scenes.knot_mesh(32, 8)
    b = rpt_amd.scenes.knot_mesh(32, 8)
    assert a.shape == (512, 18) and (a == b).all()
    assert rpt_amd.scenes.knot_mesh().shape == (100352, 18)
    g = rpt_amd.scenes.lathe_glass_mesh(128)
    assert 12000 < len(g) < 20000 and np.isfinite(g).all()
    h1, h2 = rpt_amd.scenes.synthetic_hdri(64, 32), rpt_amd.scenes.synthetic_hdri(64, 32)
    assert (h1.buf == h2.buf).all() and h1.buf.max() <= 60.0 and h1.buf.min() >= 0.0


def test_shape_nesting_limits_are_reported_at_scene_create():
    """KdTree<Box<dyn Bounded>> children: every Bounded shape incl. another group, to ANY nesting depth (kdtree.rs:14-24),
    is accepted by the flattener — which runs before the device is touched, so without a GPU the error is NO_DEVICE, not
    UNSUPPORTED; a Plane (not Bounded, kdtree.rs:9-12) as a child is UNSUPPORTED_SHAPE."""
    import rpt_amd
    from rpt_amd import GpuScene, KdTree, Object, Scene, _abi, cube, monomial_surface, plane, sphere

    def code_of(shape):
        s = Scene()
        s.add(Object(shape))
        try:
            GpuScene(s, 0).close()
            return 0
        except rpt_amd.RptGpuError as e:
            return e.code

    ok = (0, _abi.RPTGPU_E_NO_DEVICE)
    two = KdTree([KdTree([sphere(), cube().translate((2.0, 0.0, 0.0))]).translate((0.0, 1.0, 0.0)), monomial_surface(1.0, 4.0), sphere()])
    assert code_of(two) in ok
    three = KdTree([KdTree([KdTree([sphere()]), cube()]), sphere()])
    assert code_of(three) in ok
    four = KdTree([three.translate((0.0, 1.0, 0.0)), cube()])
    assert code_of(four) in ok
    deep = four
    for _ in range(20):
        deep = KdTree([deep.translate((0.1, 0.0, 0.0)), sphere()])
    assert code_of(deep) in ok
    with __import__("pytest").raises(rpt_amd.RptGpuError) as e:
        KdTree([sphere(), plane((0.0, 1.0, 0.0), 0.0)]).lower
        s = Scene()
        s.add(Object(KdTree([sphere(), plane((0.0, 1.0, 0.0), 0.0)])))
        GpuScene(s, 0)
    assert e.value.code == _abi.RPTGPU_E_UNSUPPORTED_SHAPE
