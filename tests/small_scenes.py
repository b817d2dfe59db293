"""Small variants of the BASELINE configs (same scene code, smaller meshes / HDRIs / frames) used by
the golden fixtures and the GPU parity tests, plus a `coverage` scene that exercises every shape,
light kind and camera feature the device supports."""
import math

import numpy as np

import rpt_amd
from rpt_amd import (Camera, Environment, KdTree, Light, Material, Mesh, Object, Scene, cube, hex_color,
                     make_params, monomial_surface, plane, polygon, scenes, sphere)


def coverage():
    scene = Scene()
    scene.environment = Environment.Hdri(scenes.synthetic_hdri(64, 32, seed=5))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0x999999))))
    scene.add(Object(sphere().scale((0.7, 0.9, 0.7)).translate((-1.6, -0.1, 0.3)))
              .material(Material.metallic_(hex_color(0xE0B060), 0.15)))
    scene.add(Object(sphere().translate((1.7, 0.0, -0.4))).material(Material.clear(1.5, 0.05)))
    scene.add(Object(cube().scale((0.8, 1.6, 0.8)).rotate_y(0.6).rotate_x(0.2).translate((0.2, -0.2, -1.5)))
              .material(Material.specular(hex_color(0x4060C0), 0.3)))
    scene.add(Object(cube()).material(Material.transparent_((0.7, 1.0, 0.8), 1.3, 0.2)))
    scene.add(Object(Mesh(scenes.knot_mesh(24, 6)).scale((1.5, 1.5, 1.5)).translate((0.0, 1.2, 0.5)))
              .material(Material.specular(hex_color(0xB7CA79), 0.2)))
    kids = [sphere().scale((0.2, 0.2, 0.2)).translate((math.cos(a) * 2.6, -0.8, math.sin(a) * 2.6))
            for a in np.linspace(0, 2 * math.pi, 20, endpoint=False)]
    kids += [cube().scale((0.3, 0.3, 0.3)).rotate_z(0.3 * i).translate((-2.0 + i, 2.2, -1.0)) for i in range(5)]
    kids += [sphere(), cube()]  # untransformed children too
    scene.add(Object(KdTree(kids).translate((0.0, 0.0, -0.2))).material(Material.diffuse(hex_color(0xCC6655))))
    scene.add(Light.Ambient((0.03, 0.03, 0.04)))
    scene.add(Light.Point((30.0, 28.0, 25.0), (3.0, 4.0, 3.0)))
    scene.add(Light.Directional((0.4, 0.4, 0.5), (0.3, -1.0, -0.2)))
    scene.add(Light.Object(Object(sphere().scale((0.5, 0.5, 0.5)).translate((-3.0, 3.5, 1.0)))
                           .material(Material.light((1.0, 0.9, 0.8), 60.0))))
    scene.add(Light.Object(Object(cube().scale((0.6, 0.1, 0.6)).translate((2.0, 3.0, 2.0)))
                           .material(Material.light((0.8, 0.9, 1.0), 40.0))))
    scene.add(Light.Object(Object(polygon([(-0.5, 3.8, -0.5), (-0.5, 3.8, 0.5), (0.5, 3.8, 0.5), (0.5, 3.8, -0.5)])
                                  .rotate_z(0.1)).material(Material.light((1.0, 1.0, 1.0), 50.0))))
    scene.add(Light.Object(Object(KdTree([sphere().scale((0.1, 0.1, 0.1)).translate((0.0, 2.9, 2.0 + 0.3 * i))
                                          for i in range(3)])).material(Material.light((1.0, 0.6, 0.6), 80.0))))
    camera = Camera.look_at((0.5, 2.2, 7.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), 0.7).focus((0.0, 0.0, 0.0), 0.05)
    return scene, camera


def monomial():
    """MonomialSurface seen from above, from below (the Newton branch) and edge-on, as a glass bowl, a
    mirror bowl and an area light (MonomialSurface::sample), with every light kind."""
    scene = Scene()
    scene.environment = Environment.Hdri(scenes.synthetic_hdri(64, 32, seed=9))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0x998877))))
    scene.add(Object(monomial_surface(2.0, 4.0).translate((0.0, -1.0, 0.0)))
              .material(Material.metallic_(hex_color(0xFFFFFF), 0.05)))
    scene.add(Object(monomial_surface(1.0, 4.0).scale((0.8, 1.2, 0.8)).rotate_z(0.4).translate((-2.2, -0.4, 0.5)))
              .material(Material.clear(1.5, 0.02)))
    scene.add(Object(monomial_surface(0.5, 4.0).rotate_x(math.pi).translate((2.2, 1.2, 0.0)))
              .material(Material.specular(hex_color(0x5577CC), 0.3)))
    scene.add(Object(sphere().scale((0.3, 0.3, 0.3)).translate((0.0, -0.6, 0.0)))
              .material(Material.diffuse(hex_color(0xCC4444))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((40.0, 40.0, 40.0), (0.0, 5.0, 5.0)))
    scene.add(Light.Object(Object(monomial_surface(0.3, 4.0).rotate_x(math.pi).scale((0.7, 0.7, 0.7))
                                  .translate((0.0, 3.5, 0.5))).material(Material.light((1.0, 0.95, 0.9), 30.0))))
    camera = Camera.look_at((0.0, 2.6, 6.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.75)
    return scene, camera


def nested_groups():
    """KdTree<Box<dyn Bounded>> inside KdTree<Box<dyn Bounded>> (Bounded is forwarded through Box, kdtree.rs:14-24) with
    every Bounded shape as a tree child: spheres, cubes, a mesh, monomial surfaces (monomial_surface.rs:183-190) —
    transformed and not — plus a lamp that is itself a group of groups (KdTree::sample through two levels)."""
    scene = Scene()
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0x999999))))
    mesh = Mesh(scenes.knot_mesh(24, 6))
    ring = [sphere().scale((0.18, 0.18, 0.18)).translate((math.cos(a) * 0.9, 0.0, math.sin(a) * 0.9))
            for a in np.linspace(0, 2 * math.pi, 18, endpoint=False)]
    inner = KdTree(ring + [cube().scale((0.4, 0.4, 0.4)).rotate_y(0.5), monomial_surface(0.6, 4.0).scale((0.5, 0.5, 0.5)).translate((0.0, 0.3, 0.0))])
    kids = [inner.translate((-1.6, -0.2, 0.0)), inner.scale((0.7, 1.3, 0.7)).rotate_z(0.4).translate((1.7, 0.3, -0.5)),
            mesh.scale((1.2, 1.2, 1.2)).translate((0.0, 0.9, -1.2)), monomial_surface(1.5, 4.0).translate((0.0, -1.0, 1.2)),
            monomial_surface(0.8, 4.0).rotate_x(math.pi).scale((0.6, 0.6, 0.6)).translate((0.2, 2.4, 0.4)), sphere(), cube().translate((0.0, 0.0, -3.0))]
    kids += [sphere().scale((0.12, 0.12, 0.12)).translate((-2.5 + 0.3 * i, -0.85, 2.0)) for i in range(16)]
    scene.add(Object(KdTree(kids)).material(Material.specular(hex_color(0x6688CC), 0.3)))
    scene.add(Object(KdTree([inner.translate((0.0, 1.6, 1.5)), cube().scale((0.3, 0.3, 0.3)).translate((0.9, 1.6, 1.5))]).rotate_y(0.3))
              .material(Material.clear(1.5, 0.05)))
    scene.add(Light.Ambient((0.02, 0.02, 0.03)))
    scene.add(Light.Point((25.0, 25.0, 22.0), (2.0, 4.0, 4.0)))
    lamp = KdTree([KdTree([sphere().scale((0.15, 0.15, 0.15)).translate((0.3 * i, 3.2, 0.0)) for i in range(3)]).translate((-0.5, 0.0, 1.0)),
                   cube().scale((0.3, 0.05, 0.3)).translate((1.5, 3.0, 1.0)),
                   monomial_surface(0.3, 4.0).rotate_x(math.pi).scale((0.4, 0.4, 0.4)).translate((-1.5, 3.4, 0.5))])
    scene.add(Light.Object(Object(lamp).material(Material.light((1.0, 0.95, 0.85), 45.0))))
    camera = Camera.look_at((0.4, 2.4, 6.5), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0), 0.75)
    return scene, camera


def deep_nest():
    """Groups four levels deep (what rounds 2-3 could take at most; `seven_nest` goes past any fixed limit, as the
    reference's recursion does, kdtree.rs:14-24): every level placed by its own transform, a mesh, spheres, cubes and a monomial
    surface at the bottom, siblings at every level, and a lamp that is itself nested four deep (KdTree::sample at every
    level, kdtree.rs:138-143)."""
    scene = Scene()
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    mesh = Mesh(scenes.knot_mesh(24, 6, seed=0xDEE9))
    l3 = KdTree([mesh.scale((0.5, 0.5, 0.5)), sphere().scale((0.2, 0.2, 0.2)).translate((0.8, 0.0, 0.0)),
                 cube().scale((0.3, 0.3, 0.3)).rotate_y(0.4).translate((-0.8, 0.1, 0.0)),
                 monomial_surface(0.5, 4.0).scale((0.3, 0.3, 0.3)).translate((0.0, 0.6, 0.0))] +
                [sphere().scale((0.07, 0.07, 0.07)).translate((-0.9 + 0.12 * i, -0.5, 0.3)) for i in range(16)])
    l2 = KdTree([l3.rotate_y(0.3).translate((0.0, 0.0, 0.0)), l3.scale((0.6, 0.6, 0.6)).translate((1.5, 0.4, 0.2)),
                 sphere().scale((0.25, 0.25, 0.25)).translate((-1.3, 0.0, 0.4))])
    l1 = KdTree([l2.translate((-0.4, 0.0, 0.0)), l2.scale((0.5, 0.7, 0.5)).rotate_z(0.3).translate((0.3, 1.6, -0.5)),
                 cube().scale((0.4, 0.4, 0.4)).translate((2.6, -0.6, 0.5))])
    l0 = KdTree([l1.rotate_y(-0.2), l1.scale((0.4, 0.4, 0.4)).translate((-2.4, -0.3, 1.2)), sphere().translate((0.0, 0.0, -3.5))])
    scene.add(Object(l0.translate((0.0, 0.2, 0.0))).material(Material.specular(hex_color(0x88AA66), 0.3)))
    scene.add(Object(KdTree([l2.scale((0.5, 0.5, 0.5)).translate((2.2, 1.8, 1.0))])).material(Material.clear(1.5, 0.05)))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((30.0, 30.0, 28.0), (2.0, 4.0, 4.5)))
    b3 = KdTree([sphere().scale((0.12, 0.12, 0.12)).translate((0.3 * i, 0.0, 0.0)) for i in range(3)] +
                [mesh.scale((0.15, 0.15, 0.15)).translate((0.45, 0.2, 0.0))])
    b2 = KdTree([b3.translate((0.0, 0.0, 0.2)), cube().scale((0.2, 0.05, 0.2)).translate((1.2, 0.0, 0.0))])
    b1 = KdTree([b2.rotate_y(0.5), b3.scale((0.8, 0.8, 0.8)).translate((-1.0, 0.1, 0.0))])
    lamp = KdTree([b1.translate((-0.3, 3.2, 0.8)), sphere().scale((0.1, 0.1, 0.1)).translate((1.4, 3.1, 0.6))])
    scene.add(Light.Object(Object(lamp).material(Material.light((1.0, 0.95, 0.85), 50.0))))
    camera = Camera.look_at((0.6, 2.2, 6.8), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0), 0.8)
    return scene, camera


def seven_nest():
    """Groups SEVEN levels deep — past anything a fixed number of traversal instantiations would cover; the reference
    recurses without a limit (kdtree.rs:14-24 forwards Bounded through Box) and the device walks such an object with
    rpt_tree_generic.  Every level has its own transform and siblings of every kind (a mesh two, four and seven levels
    down, spheres, cubes, a monomial surface), two of the groups are large enough to be real trees (>= 16 children), one
    group is used twice (a shared subtree reached through different transforms), and one object is glass so that paths
    continue INSIDE the nest.  The lamp is a group nested three deep (Shape::sample keeps a limit of eight)."""
    scene = Scene()
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xA0A0A0))))
    mesh = Mesh(scenes.knot_mesh(24, 6, seed=0x7E57))
    g = KdTree([mesh.scale((0.45, 0.45, 0.45)), sphere().scale((0.18, 0.18, 0.18)).translate((0.7, 0.0, 0.0))] +
               [sphere().scale((0.06, 0.06, 0.06)).translate((-0.8 + 0.1 * i, -0.45, 0.25)) for i in range(17)])       # level 6
    for lvl in range(5, -1, -1):  # levels 5 .. 0, each wrapping the one below twice (once as it is, once transformed)
        kids = [g.rotate_y(0.2 + 0.1 * lvl).translate((0.15 * lvl, 0.05 * lvl, 0.0)),
                g.scale((0.55, 0.55, 0.55)).rotate_z(0.25).translate((1.1 + 0.1 * lvl, 0.5, -0.3 + 0.1 * lvl)),
                (sphere() if lvl % 2 else cube().rotate_y(0.37).rotate_x(0.21)).scale((0.22, 0.22, 0.22)).translate((-1.0 - 0.1 * lvl, 0.1 * lvl, 0.5))]
        if lvl == 3:
            kids.append(mesh.scale((0.3, 0.3, 0.3)).translate((0.0, 1.0, 0.6)))
            kids += [cube().scale((0.08, 0.08, 0.08)).rotate_y(0.3 * i + 0.1).rotate_x(0.15).translate((-1.2 + 0.15 * i, -0.6, 0.9)) for i in range(16)]
        if lvl == 1:
            kids.append(monomial_surface(0.6, 4.0).scale((0.35, 0.35, 0.35)).translate((-0.6, 0.9, 0.2)))
        g = KdTree(kids)
    scene.add(Object(g.scale((0.8, 0.8, 0.8)).translate((-0.3, 0.1, 0.0))).material(Material.specular(hex_color(0x7799BB), 0.3)))
    inner = KdTree([KdTree([KdTree([mesh.scale((0.35, 0.35, 0.35)), sphere().scale((0.3, 0.3, 0.3)).translate((0.6, 0.0, 0.0))])
                            .rotate_x(0.3), cube().rotate_y(0.45).rotate_z(0.2).scale((0.25, 0.25, 0.25)).translate((-0.6, 0.1, 0.0))]).translate((0.0, 0.2, 0.0)),
                    sphere().scale((0.2, 0.2, 0.2)).translate((0.0, 0.9, 0.0))])
    scene.add(Object(inner.translate((2.3, 0.0, 1.2))).material(Material.clear(1.5, 0.05)))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((28.0, 28.0, 26.0), (1.5, 4.0, 4.5)))
    scene.add(Light.Directional((0.3, 0.3, 0.35), (0.0, -0.6, -0.8)))  # a sun in a coordinate plane: a zero direction component in every
    # shadow ray (no face of the scene is exactly parallel to it: bsdf would be 0/0 there, as in the reference)
    lamp = KdTree([KdTree([KdTree([sphere().scale((0.12, 0.12, 0.12)).translate((0.3 * i, 0.0, 0.0)) for i in range(3)])
                           .translate((0.0, 0.0, 0.2)), cube().scale((0.2, 0.05, 0.2)).translate((1.0, 0.0, 0.0))]).rotate_y(0.4),
                   sphere().scale((0.1, 0.1, 0.1)).translate((-0.8, 0.1, 0.0))])
    scene.add(Light.Object(Object(lamp.translate((0.0, 3.2, 0.8))).material(Material.light((1.0, 0.95, 0.85), 45.0))))
    camera = Camera.look_at((0.8, 2.3, 6.6), (0.2, 0.4, 0.0), (0.0, 1.0, 0.0), 0.8)
    return scene, camera


def axis_sun(oblique=False):
    """(oblique: the same geometry under lights off every axis and coordinate plane — rays with a zero direction
    component are then RARE for the library, which hands them to rpt_tree_generic instead of launching the ZEROS form of
    rpt_tree_trace: launch_query, StackSpill::zeros_common.)
    Directional lights along the axes and in a coordinate plane over an UNTRANSFORMED deep mesh, a group of spheres and
    a plane: every shadow ray has one or two direction components that are exactly zero (the compact traversal's and the
    leaf-box filter's special case, kernels/traversal.inc, shapes.inc)."""
    scene = Scene()
    rows = scenes.knot_mesh(96, 16, seed=0xA715)
    rows[:, :9] *= 2.2
    scene.add(Object(Mesh(rows)).material(Material.specular(hex_color(0xB7CA79), 0.3)))
    kids = [sphere().scale((0.12, 0.12, 0.12)).translate((-2.0 + 0.4 * (i % 11), -0.9 + 0.35 * (i // 11), 1.6 - 0.2 * (i % 3)))
            for i in range(66)]
    scene.add(Object(KdTree(kids)).material(Material.diffuse(hex_color(0xCC7755))))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.2)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    if oblique:
        scene.add(Light.Directional((0.9, 0.9, 0.8), (0.1, -1.0, 0.2)))
        scene.add(Light.Directional((0.3, 0.3, 0.5), (-1.0, -1.0, 0.3)))
    else:
        scene.add(Light.Directional((0.9, 0.9, 0.8), (0.0, -1.0, 0.0)))
        scene.add(Light.Directional((0.3, 0.3, 0.5), (-1.0, -1.0, 0.0)))  # (exactly horizontal light would be 0/0 in bsdf on the plane)
        scene.add(Light.Directional((0.4, 0.3, 0.3), (0.0, -0.6, -0.8)))
    camera = Camera.look_at((1.5, 2.5, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.7)
    return scene, camera


def small(name):
    """-> (scene, camera, params) for the named small config."""
    if name == "sphere":
        s, c, d = scenes.sphere_scene()
        return s, c, make_params(64, 36, 2, 8, seed=101)
    if name == "cornell":
        s, c, d = scenes.cornell()
        return s, c, make_params(64, 36, 8, 8, seed=102)
    if name == "dragon":
        s, c, d = scenes.dragon(nu=96, nv=16)
        return s, c, make_params(64, 36, 4, 4, seed=103)
    if name == "fractal_spheres":
        s, c, d = scenes.fractal_spheres()
        return s, c, make_params(64, 36, 4, 4, seed=104)
    if name == "glass":
        s, c, d = scenes.glass(hdri_size=(256, 128))
        return s, c, make_params(64, 48, 6, 4, seed=105)
    if name == "wine_glass":
        s, c, d = scenes.wine_glass(hdri_size=(256, 128), segments=24)
        return s, c, make_params(64, 36, 8, 4, seed=106)
    if name == "coverage":
        s, c = coverage()
        return s, c, make_params(64, 36, 5, 4, seed=107, exposure_value=0.5)
    if name == "monomial":
        s, c = monomial()
        return s, c, make_params(64, 36, 4, 4, seed=108)
    if name == "monomial_glass":
        s, c, d = scenes.monomial_glass(hdri_size=(128, 64))
        return s, c, make_params(64, 48, 1, 4, seed=109)
    if name == "basic":
        s, c, d = scenes.basic()
        return s, c, make_params(64, 48, 0, 1, seed=110)
    if name == "spheres":
        s, c, d = scenes.spheres()
        return s, c, make_params(64, 48, 6, 4, seed=111)
    if name == "compound":
        s, c, d = scenes.compound()
        return s, c, make_params(48, 48, 5, 4, seed=112)
    if name == "fractal_teapots":
        m = Mesh(scenes._teapot_stand_in(24, 6))
        s, c, d = scenes.fractal_teapots(levels=3, mesh=m)
        # additions to the example: a lamp that is itself a kd-tree of kd-trees (KdTree::sample over a
        # Mesh child), and bounces (the example renders with the Renderer defaults, 0 bounces) so that
        # shadow and bounce rays go through the nested traversal too
        s.add(Light.Object(Object(KdTree([m.scale((0.3, 0.3, 0.3)).translate((1.5, 2.5, 3.0)),
                                          sphere().scale((0.2, 0.2, 0.2)).translate((2.2, 2.5, 3.0))]))
                           .material(Material.light((1.0, 0.9, 0.8), 25.0))))
        return s, c, make_params(64, 48, 3, 4, seed=113)
    if name == "nested_groups":
        s, c = nested_groups()
        return s, c, make_params(64, 40, 4, 4, seed=114)
    # the asset-driven examples with small stand-ins for their assets
    if name == "teapot":
        s, c, d = scenes.teapot(mesh=Mesh(scenes._teapot_stand_in(24, 6)))
        return s, c, make_params(48, 48, 3, 4, seed=115)     # the example itself renders with 0 bounces
    if name == "cylinder":
        s, c, d = scenes.cylinder(mesh=scenes._cylinder_stand_in(12))
        return s, c, make_params(48, 48, 2, 4, seed=116)
    if name == "rustacean":
        rows = scenes.knot_mesh(48, 8, seed=0xFE2215)
        rows[:, :9] *= 1.2
        s, c, d = scenes.rustacean(mesh=Mesh(rows).translate((0.0, 0.45, 0.0)))
        return s, c, make_params(64, 64, 4, 4, seed=117)
    if name == "pegasus":
        rows = scenes.knot_mesh(48, 8, seed=0x9E6A)
        rows[:, :9] *= 0.55
        s, c, d = scenes.pegasus(mesh=Mesh(rows).translate((0.0, 0.72, 0.0)), hdri_size=(128, 64))
        return s, c, make_params(48, 48, 8, 4, seed=118, exposure_value=d["exposure_value"])
    if name == "metal":
        s, c, d = scenes.metal(mesh=Mesh(scenes._teapot_stand_in(24, 6)), hdri_size=(128, 64))
        return s, c, make_params(64, 48, 5, 4, seed=119)
    if name == "simple_video":
        s, c, d = scenes.simple_video(frame=7)
        return s, c, make_params(64, 48, 1, 4, seed=120)
    if name == "deep_nest":
        s, c = deep_nest()
        return s, c, make_params(64, 40, 4, 4, seed=122)
    if name == "axis_sun":
        s, c = axis_sun()
        return s, c, make_params(64, 40, 4, 4, seed=121)
    if name == "seven_nest":
        s, c = seven_nest()
        return s, c, make_params(64, 40, 5, 4, seed=123)
    # the same scenes at 256x144 with 32 spp: ~10^6 samples each, so that draw sequences a 64x36 frame at 4 spp
    # hardly ever produces (long rejection loops, TIR, gen_range redraws, deep clamp chains) do occur
    if name == "cornell_hi":
        s, c, d = scenes.cornell()
        return s, c, make_params(256, 144, 8, 32, seed=202)
    if name == "coverage_hi":
        s, c = coverage()
        return s, c, make_params(256, 144, 5, 32, seed=207, exposure_value=0.5)
    raise KeyError(name)


HI_NAMES = ["cornell_hi", "coverage_hi"]
NAMES = ["sphere", "cornell", "dragon", "fractal_spheres", "glass", "wine_glass", "coverage",
         "monomial", "monomial_glass", "basic", "spheres", "compound", "fractal_teapots", "nested_groups",
         "teapot", "cylinder", "rustacean", "pegasus", "metal", "simple_video", "axis_sun", "deep_nest", "seven_nest"]


# ---- kd-trees whose median is a zero of either sign (tests/test_kdtree.py, tests/test_gpu_kdbuild.py)
def mixed_zero_boxes(order):
    """100 boxes whose sorted x edges have twenty zeros around the middle: forty boxes left of x = 0, forty right of it,
    ten that END at -0.0 ('A') and ten that START at +0.0 ('B'), the twenty in the index order given.  The root splits on
    x (the largest extent; 60 of 100 boxes on either side) at a median that is (zero + zero) / 2 — its SIGN is decided by
    which two zeros the stable sort leaves in the middle: the tenth and the eleventh of the twenty, in index order."""
    rs = np.random.RandomState(11)
    boxes = []
    for i in range(40):
        boxes.append([-2.0 + 0.01 * i, 0.0, 0.0, -1.0 + 0.01 * i, 0.3, 0.3])
    touching = []
    for kind in order:
        touching.append([-1.0, 0.0, 0.0, -0.0, 0.3, 0.3] if kind == "A" else [0.0, 0.0, 0.0, 1.0, 0.3, 0.3])
    boxes += touching
    for i in range(40):
        boxes.append([1.0 + 0.01 * i, 0.0, 0.0, 2.0 + 0.01 * i, 0.3, 0.3])
    b = np.array(boxes)
    b[:, [1, 2]] = rs.rand(len(b), 2) * 0.2  # y, z: small, distinct
    b[:, [4, 5]] = b[:, [1, 2]] + 0.05 + rs.rand(len(b), 2) * 0.1
    return b


MIXED_ZERO_ORDERS = [
    ("B" * 8 + "A" * 10 + "B" * 2, True),    # the 10th and 11th zero are both -0.0: (-0 + -0) / 2 = -0
    ("A" * 10 + "B" * 10, False),            # -0.0, +0.0 in the middle: +0 (what a sort that puts every -0.0 first always gives)
    ("A" * 8 + "B" * 4 + "A" * 2 + "B" * 6, False),  # +0.0, +0.0: +0 — where a -0.0-first order has -0.0, -0.0
    ("AB" * 10, False),
    ("B" + "A" * 10 + "B" * 9, True),
]


def mixed_zero_mesh(order):
    """A mesh of one triangle per box of mixed_zero_boxes(order), each spanning its box exactly — the vertices' x are the
    box's two x edges, signed zeros included — with flat normals: the kd-tree's root splits on x at a zero median."""
    b = mixed_zero_boxes(order)
    lo, hi = b[:, :3], b[:, 3:]
    v0 = lo.copy()
    v1 = np.stack([hi[:, 0], hi[:, 1], lo[:, 2]], axis=1)
    v2 = np.stack([lo[:, 0], hi[:, 1], hi[:, 2]], axis=1)
    n = np.cross(v1 - v0, v2 - v0)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return np.concatenate([v0, v1, v2, n, n, n], axis=1)
