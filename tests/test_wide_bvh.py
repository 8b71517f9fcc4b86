"""The 4-wide tree of 48-byte nodes (bvh.hpp, BvhNode4) that experiment builds of the kernels walk (-DMI355RT_WIDE=1; profiles/r03_notes.md:
parity green, 17 % slower than the binary walk, not shipped).  Host code: built and checked without a GPU through mi355rt_debug_wide_bvh, whose
walk decodes the node words the way traverse.hpp does — every triangle reached exactly once, every child box holds every vertex below it, the
binary tree still valid over the re-ordered triangles."""
import os
import numpy as np
import pytest


def _blob(n, seed, spread=0.2, extent=10.0):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.3 * extent, 0.7 * extent, (n, 1, 3))
    return (c + rng.uniform(-spread, spread, (n, 3, 3))).astype(np.float32).reshape(n, 9)


@pytest.mark.parametrize("n,seed", [(2, 1), (5, 2), (37, 3), (300, 4), (5000, 5), (40000, 6)])
def test_wide_tree_reaches_every_triangle_once_and_its_boxes_hold_what_is_below(pkg, n, seed):
    f = pkg.debug_wide_bvh(_blob(n, seed))
    assert f["wide_nodes"] > 0
    assert f["tris_once_wide"] == n and f["tris_once_binary"] == n and f["bad_boxes"] == 0
    assert f["wide_nodes"] <= f["binary_nodes"] and 2 * f["wide_nodes"] <= f["children"] <= 4 * f["wide_nodes"]
    assert f["binary_depth"] <= f["stack_need"] <= 3 * f["binary_depth"]          # up to three pushes per step


def test_scene_files_and_flat_or_tiny_geometry(pkg, scenes):
    for name in ("thai2", "ico2", "4boxes"):
        v = scenes(name)["tri_verts"].reshape(-1, 9)
        f = pkg.debug_wide_bvh(v)
        assert f["wide_nodes"] > 0 and f["tris_once_wide"] == len(v) and f["tris_once_binary"] == len(v) and f["bad_boxes"] == 0, name
    flat = _blob(500, 7); flat[:, 2::3] = 1.25                                   # every vertex in one plane: a zero extent on one axis
    f = pkg.debug_wide_bvh(flat)
    assert f["wide_nodes"] > 0 and f["tris_once_wide"] == 500 and f["bad_boxes"] == 0
    far = _blob(800, 8, spread=1e-3, extent=1e-2) + np.float32(5000.0)           # tiny geometry far from the origin (coarse half-precision boxes)
    f = pkg.debug_wide_bvh(far)
    assert f["tris_once_binary"] == 800 and (f["wide_nodes"] == 0 or (f["tris_once_wide"] == 800 and f["bad_boxes"] == 0))


def test_the_format_stands_down_where_it_does_not_apply(pkg):
    one = _blob(1, 9)                                                           # a single leaf: no inner node to collapse
    assert pkg.debug_wide_bvh(one)["wide_nodes"] == 0
    huge = _blob(50, 10) * np.float32(1e4)                                      # beyond the half range: the binary boxes hold +-inf
    f = pkg.debug_wide_bvh(huge)
    assert f["wide_nodes"] == 0 and f["tris_once_wide"] == 0
