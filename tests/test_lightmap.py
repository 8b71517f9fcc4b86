"""The depth cube maps around the lights (csrc/lightmap.cpp; host code, runs without a GPU): the bound they hold must never exceed the squared
distance of ANY point of ANY triangle seen in a texel's directions — that is the whole argument for leaving a shadow ray out (kernels.hip,
light_proves_unoccluded).  Checked here against brute force: points sampled on every triangle are looked up the way the kernel looks them up (f32)."""
import numpy as np
import pytest


def texel_of(v):
    """kernels.hip, light_proves_unoccluded, in f32: direction v (from the light) -> (face, i, j) for a map of R texels per edge; R applied by the caller"""
    v = v.astype(np.float32)
    a = np.abs(v)
    m = np.where((a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]), 0, np.where(a[:, 1] >= a[:, 2], 1, 2))
    idx = np.arange(len(v))
    vm = v[idx, m]
    ax_a = np.array([1, 0, 0])[m]; ax_b = np.array([2, 2, 1])[m]
    am = np.abs(vm)
    u = (v[idx, ax_a] / am).astype(np.float32); w = (v[idx, ax_b] / am).astype(np.float32)
    face = 2 * m + (vm < 0)
    return face, u, w


def lookup(dist2, v):
    R = dist2.shape[1]
    face, u, w = texel_of(v)
    i = np.clip(((u * np.float32(0.5) + np.float32(0.5)) * np.float32(R)).astype(np.int64), 0, R - 1)
    j = np.clip(((w * np.float32(0.5) + np.float32(0.5)) * np.float32(R)).astype(np.int64), 0, R - 1)
    return dist2[face, i, j]


def sample_points(tris, per_tri, rng):
    a = rng.random((tris.shape[0], per_tri)); b = rng.random((tris.shape[0], per_tri))
    f = a + b > 1; a[f] = 1 - a[f]; b[f] = 1 - b[f]
    # corners and edges too: the extremes of a triangle are where a bound breaks first
    a[:, 0] = 0; b[:, 0] = 0; a[:, 1] = 1; b[:, 1] = 0; a[:, 2] = 0; b[:, 2] = 1; a[:, 3] = 0.5; b[:, 3] = 0.5; b[:, 4] = 0; a[:, 5] = 0
    v0, v1, v2 = tris[:, None, 0:3], tris[:, None, 3:6], tris[:, None, 6:9]
    return (v0 + a[..., None] * (v1 - v0) + b[..., None] * (v2 - v0)).reshape(-1, 3)


@pytest.mark.parametrize("name,light,res", [("thai2", None, 256), ("thai2", (0.0, 0.0, 6.0), 128), ("ico2", None, 128), ("ico2", (0.05, 0.02, 0.01), 64),
                                            ("4boxes", None, 64), ("4boxes", (1.0, 0.4, 1.0), 32), ("ico3_tex", (30.0, 40.0, -25.0), 512)])
def test_bound_is_below_every_point_of_every_triangle(pkg, scenes, name, light, res):
    sc = scenes(name)
    tris = sc["tri_verts"].astype(np.float64)
    L = np.asarray(sc["lights"][0, :3] if light is None else light, np.float64)
    pts_all = tris.reshape(-1, 3)
    pad = 2e-4 * np.linalg.norm(pts_all.max(0) - pts_all.min(0))
    dist2, nearest = pkg.debug_light_map(sc["tri_verts"], L.astype(np.float32), pad, res)
    L = L.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(3)
    pts = sample_points(tris, 24, rng)
    d = pts - L
    d2 = (d * d).sum(1)
    ok = d2 > 1e-12
    bound = lookup(dist2, d[ok])
    assert np.all(np.isfinite(bound))                          # a texel that sees a triangle point knows about it
    assert np.all(bound.astype(np.float64) <= d2[ok])
    assert nearest ** 2 <= d2.min() * (1 + 1e-9)
    # the padding is in there: the bound stays below points pushed `pad` towards the light as well
    pushed = d[ok] * (1.0 - np.minimum(pad / np.sqrt(d2[ok]), 1.0))[:, None]
    assert np.all(bound.astype(np.float64) <= (pushed * pushed).sum(1) * (1 + 1e-6) + 1e-12)
    # and it is not trivially zero: most texels that see something hold a bound within a few percent of what they see
    seen = np.isfinite(dist2)
    assert seen.sum() > 0
    if light is None and name == "thai2":
        near = np.full(dist2.shape, np.inf)
        face, u, w = texel_of(d[ok]); R = res
        i = np.clip(((u * 0.5 + 0.5) * R).astype(np.int64), 0, R - 1); j = np.clip(((w * 0.5 + 0.5) * R).astype(np.int64), 0, R - 1)
        np.minimum.at(near, (face, i, j), d2[ok])
        both = np.isfinite(near)
        ratio = np.sqrt(dist2[both].astype(np.float64) / near[both])
        assert np.median(ratio) > 0.9


def test_random_soup_and_degenerate_triangles(pkg):
    """random triangles all around a light, some through it, some degenerate (zero area, repeated vertices)"""
    rng = np.random.default_rng(9)
    c = rng.uniform(-4, 4, (400, 1, 3)); tris = (c + rng.normal(0, 0.8, (400, 3, 3))).reshape(400, 9)
    tris[0] = [-1, -1, 0.0, 1, -1, 0.0, 0, 2, 0.0]             # through the light at the origin
    tris[1] = [1, 1, 1, 1, 1, 1, 1, 1, 1]                       # a point
    tris[2] = [0, 2, 0, 0, 3, 0, 0, 4, 0]                       # a segment
    tris[3] = [2, 0, 0, 2, 0, 0, 2, 1, 0]
    L = np.zeros(3)
    dist2, nearest = pkg.debug_light_map(tris.astype(np.float32), L.astype(np.float32), 1e-3, 64)
    assert nearest == 0.0
    pts = sample_points(tris.astype(np.float32).astype(np.float64), 40, rng)
    d2 = (pts * pts).sum(1)
    ok = d2 > 1e-12
    bound = lookup(dist2, pts[ok])
    assert np.all(bound.astype(np.float64) <= d2[ok])
    assert (dist2 == 0).sum() > 0                               # the texels the triangle through the light is seen in prove nothing
