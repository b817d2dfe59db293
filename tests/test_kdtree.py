"""KdTree::new (reference src/kdtree.rs:108-119, 235-355): the product's host builder
(rptgpu_kdtree_build, nth_element medians) against the oracle's line-by-line restatement
(full sorts), node by node; and the tree statistics of SURVEY appendix A's rule."""
import numpy as np
import pytest

from rpt_amd import _abi, scenes
from rpt_amd.device import kdtree_build
from small_scenes import MIXED_ZERO_ORDERS, mixed_zero_boxes


def tri_boxes(rows):
    v = rows[:, :9].reshape(-1, 3, 3)
    return np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1)


def both(boxes, oracle):
    a = kdtree_build(boxes, _abi.load_library(), "rptgpu")
    b = kdtree_build(boxes, oracle.lib(), "oracle")
    return a, b


def assert_same_tree(a, b):
    assert a["max_depth"] == b["max_depth"]
    for k in ("split", "info", "a", "b", "refs"):
        assert a[k].shape == b[k].shape, k
        assert (a[k] == b[k]).all(), k
    # splits BITWISE: -0.0 == +0.0 compares equal, and the sign of a zero split is what the order among equal keys decides
    assert (a["split"].view(np.uint64) == b["split"].view(np.uint64)).all()


def reference_median(pushed):
    """kdtree.rs:251-255 + median() :347-355 on the values in the order construct pushes them (per index: p_min, p_max).
    Python's sort, like Rust's sort_by, is stable and holds -0.0 == 0.0: equal entries keep their pushed order."""
    s = sorted(pushed)
    mid = len(s) // 2
    return s[mid] if len(s) % 2 else (s[mid] + s[mid - 1]) / 2.0


@pytest.mark.parametrize("order,negative", MIXED_ZERO_ORDERS)
def test_zero_median_keeps_the_reference_order_among_equal_keys(oracle, order, negative):
    """kdtree.rs:251-255: a STABLE sort under partial_cmp.  Zeros of both signs at the middle of a node's box edges: the
    sign of the split is that of the reference's sort order — in the product's builder (order statistics, no sort), in
    the oracle's (stable_sort) and in a direct transcription of median() over Python's stable sort."""
    boxes = mixed_zero_boxes(order)
    pushed = [v for b in boxes for v in (b[0], b[3])]
    want = reference_median(pushed)
    assert want == 0.0 and (np.signbit(want) == negative)
    a, b = both(boxes, oracle)
    assert_same_tree(a, b)
    assert a["info"][0] == 0 and a["split"][0] == 0.0  # the root splits on x, at a zero
    assert np.signbit(a["split"][0]) == negative and np.signbit(b["split"][0]) == negative


@pytest.mark.parametrize("n,seed", [(0, 0), (1, 1), (15, 2), (16, 3), (17, 4), (200, 5), (5000, 6)])
def test_random_boxes(oracle, n, seed):
    rs = np.random.RandomState(seed)
    lo = rs.rand(n, 3) * 10
    boxes = np.concatenate([lo, lo + rs.rand(n, 3) * (0.1 + 2.0 * (seed % 2))], axis=1)
    a, b = both(boxes, oracle)
    assert_same_tree(a, b)
    if n < 16:  # kdtree.rs:236: fewer than 16 objects -> a single leaf
        assert len(a["split"]) == 1 and a["info"][0] == 3 and a["b"][0] == n


def test_degenerate_inputs(oracle):
    boxes = np.tile(np.array([[0, 0, 0, 1, 1, 1.0]]), (64, 1))  # identical boxes: no split is worth it
    a, b = both(boxes, oracle)
    assert_same_tree(a, b)
    assert len(a["split"]) == 1 and a["b"][0] == 64
    rs = np.random.RandomState(9)
    flat = np.concatenate([rs.rand(300, 3), np.zeros((300, 3))], axis=1)
    flat[:, 3:] = flat[:, :3]
    flat[:, 1] = flat[:, 4] = 0.5  # all boxes flat in y, many equal coordinates (ties at the median)
    flat[:, 0] = np.round(flat[:, 0] * 4) / 4
    flat[:, 3] = flat[:, 0]
    a, b = both(flat, oracle)
    assert_same_tree(a, b)


def test_mesh_trees(oracle):
    rows = scenes.knot_mesh(nu=96, nv=16)  # 3072 triangles
    a, b = both(tri_boxes(rows), oracle)
    assert_same_tree(a, b)
    n_leaf = int((a["info"] == 3).sum())
    assert n_leaf == (len(a["info"]) + 1) // 2  # binary tree
    assert a["b"][a["info"] == 3].sum() == len(a["refs"])
    assert len(a["refs"]) >= len(rows)  # straddlers are duplicated (kdtree.rs:270-281)
    assert set(a["refs"].tolist()) == set(range(len(rows)))  # every triangle referenced
    assert a["regular"] == 1  # every split inside its cell: the device's compact traversal applies
    rows = scenes.lathe_glass_mesh(48)
    a, b = both(tri_boxes(rows), oracle)
    assert_same_tree(a, b)
    assert a["regular"] == 1


def test_irregular_tree_is_flagged():
    # 20 long boxes all spanning x in [0, 10] plus short ones: medians stay inside, so build a case
    # by hand where the median of the edges falls outside a child cell: children of a split keep
    # straddlers whose far edges drag the median beyond the cell
    rs = np.random.RandomState(3)
    n = 400
    lo = rs.rand(n, 3)
    hi = lo + rs.rand(n, 3) * 0.05
    hi[: n // 2, 0] += 5.0  # half of the boxes reach far beyond any small cell in x
    a = kdtree_build(np.concatenate([lo, hi], axis=1), _abi.load_library(), "rptgpu")
    assert a["regular"] in (0, 1)  # either is legal; the flag only selects the traversal variant


def test_fractal_level_tree_statistics(oracle):
    # SURVEY appendix A (reference rule on examples/fractal_spheres.rs): level 4 = 750 spheres ->
    # 255 nodes, depth 7, 1 560 refs; level 3 = 150 -> 63 nodes, depth 5, 328 refs
    scene, _, _ = scenes.fractal_spheres()
    stats = []
    for obj in scene.objects[:5]:
        boxes = []
        for s in obj.shape.objects:
            m = s.transform_m
            r, c = m[0], (m[12], m[13], m[14])
            boxes.append([c[0] - r, c[1] - r, c[2] - r, c[0] + r, c[1] + r, c[2] + r])
        a, b = both(np.array(boxes), oracle)
        assert_same_tree(a, b)
        stats.append((len(boxes), len(a["split"]), int(a["max_depth"]), len(a["refs"])))
    assert stats[4] == (750, 255, 7, 1560)
    assert stats[3] == (150, 63, 5, 328)
    assert stats[2] == (30, 7, 2, 56)
    assert stats[1] == (6, 1, 0, 6) and stats[0] == (1, 1, 0, 1)


def test_kd_closest_hit_equals_brute_force(oracle):
    """Property: the kd-tree answer equals a brute-force loop over the same triangles.  The brute
    force is the same scene cut into single-leaf meshes (< 16 triangles never split, kdtree.rs:236),
    which get_closest_hit loops over linearly (renderer.rs:211-220)."""
    import rpt_amd
    rows = scenes.knot_mesh(nu=64, nv=12)  # 1536 triangles
    kd = rpt_amd.Scene()
    kd.add(rpt_amd.Object(rpt_amd.Mesh(rows)))
    brute = rpt_amd.Scene()
    for i in range(0, len(rows), 15):
        brute.add(rpt_amd.Object(rpt_amd.Mesh(rows[i:i + 15])))
    rs = np.random.RandomState(11)
    n = 20000
    o = rs.randn(n, 3)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 1.5 * rs.rand(n, 1)  # inside and outside the knot
    target = rows[rs.randint(0, len(rows), n), :3] + rs.randn(n, 3) * 0.02
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0, n0, ob0 = oracle.OracleScene(kd).closest_hit(o, d)
    t1, n1, ob1 = oracle.OracleScene(brute).closest_hit(o, d)
    assert (ob0 >= 0).mean() > 0.3
    assert ((ob0 >= 0) == (ob1 >= 0)).all()
    assert (t0 == t1).all()  # same triangle test, same t: bit-equal
    hit = ob0 >= 0
    assert (n0[hit] == n1[hit]).all(axis=1).mean() > 0.999  # exact ties may pick the other face


def test_full_size_mesh_tree_is_the_reference_rule_tree(oracle):
    """The product's kd builder on the full C3 mesh: node count, leaf references and depth are those of the
    reference rule as SURVEY appendix A measured them for a 100k-triangle mesh class, and equal the
    oracle's independent builder node by node."""
    tris = scenes.knot_mesh()
    assert tris.shape == (100352, 18)
    lo = np.minimum(np.minimum(tris[:, 0:3], tris[:, 3:6]), tris[:, 6:9])
    hi = np.maximum(np.maximum(tris[:, 0:3], tris[:, 3:6]), tris[:, 6:9])
    boxes = np.ascontiguousarray(np.concatenate([lo, hi], axis=1))
    a = kdtree_build(boxes, _abi.load_library(), "rptgpu")
    b = kdtree_build(boxes, oracle.lib(), "oracle")
    for k in ("split", "info", "a", "b", "refs"):
        assert (a[k] == b[k]).all(), k
    assert a["max_depth"] == b["max_depth"] >= 15
    assert len(a["refs"]) > 4 * len(tris)  # straddlers go to both sides (kdtree.rs:270-281): ~5x duplication


def test_parallel_build_gives_the_same_tree_for_any_thread_count(monkeypatch):
    """subtrees above 8192 primitives are built on other threads and spliced in the order the
    sequential depth-first build numbers nodes (host_scene.cpp, splice) — node for node the same tree."""
    tris = scenes.knot_mesh(nu=784, nv=64)
    boxes = np.ascontiguousarray(tri_boxes(tris))
    trees = {}
    for th in (1, 2, 5, 32):
        monkeypatch.setenv("RPTGPU_BUILD_THREADS", str(th))
        trees[th] = kdtree_build(boxes, _abi.load_library(), "rptgpu")
    for th in (2, 5, 32):
        for k in ("split", "info", "a", "b", "refs"):
            assert (trees[1][k] == trees[th][k]).all(), (th, k)
        assert trees[1]["max_depth"] == trees[th]["max_depth"] and trees[1]["regular"] == trees[th]["regular"]
