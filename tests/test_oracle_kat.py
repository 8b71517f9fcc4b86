"""CPU tests pinning the oracle (oracle/oracle.c):
  - every known-answer vector the reference's own tests hold for this path:
    4 slab tests (oct_tree_intersector.rs:475-512), 2 matrix identities (vecmath.rs:343-359),
    2 COLLADA matrix conversions (collada_types.rs:98-125);
  - the scene / octree facts of SURVEY.md §6.2;
  - hand-computed answers for the pieces no reference test covers (parity unpinned there: the
    reference source text is the only specification);
  - the committed golden renders (regression pin of the oracle itself)."""
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- reference KATs ---------------------------------------------------------------------------
CUBE = [-1, -1, -1, 1, 1, 1]


def inv(d):
    with np.errstate(divide="ignore"):
        return list((np.float32(1.0) / np.asarray(d, np.float32)))


def test_slab_kat_basic(oracle):           # test_intersect_cube_inverse_ray
    hit, t = oracle.slab([2, 0, 0] + inv([-1, 0.1, 0.1]), CUBE)
    assert hit and t == 1.0


def test_slab_kat_handles_inf(oracle):     # ..._handles_inf: 1/0 = inf by design
    hit, t = oracle.slab([2, 0, 0] + inv([-1, 0.0, 0.0]), CUBE)
    assert hit and t == 1.0


def test_slab_kat_start_inside(oracle):    # ..._start_inside: negative entry distance
    hit, t = oracle.slab([-0.9, 0, 0] + inv([1, 0.1, 0.1]), CUBE)
    assert hit and t < 0.0


def test_slab_kat_should_miss(oracle):     # ..._should_miss: box behind the ray
    hit, _ = oracle.slab([-2, 0, 0] + inv([-1, 0.1, 0.1]), CUBE)
    assert not hit


def test_matrix_identity_kats(oracle):     # test_mul_identities, test_mul_vec_mat
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    assert np.array_equal(oracle.mat_mul(ident, ident), ident)
    assert np.array_equal(oracle.mat_vec(ident, [1, 2, 3, 4]), np.array([1, 2, 3, 4], np.float32))


def test_collada_matrix_kats(oracle):      # collada_mat_to_vecmat_translation / _z_vec
    m = oracle.collada_matrix_to_vecmath([0, 0, 0, 10, 0, 0, 0, 20, 0, 0, 0, 30, 0, 0, 0, 1])
    assert np.array_equal(m, np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 10, 30, 20, 1], np.float32))
    m = oracle.collada_matrix_to_vecmath(np.eye(4).reshape(-1))
    assert np.array_equal(oracle.mat_vec(m, [0, 0, 1, 1])[:3], np.array([0, -1, 0], np.float32))


# ---- SURVEY.md §6.2 scene facts -----------------------------------------------------------------
FACTS = {
    "ico2": dict(ntri=608, stats=(145, 18, 127, 2, 4, 2275), cam=(3.1533, 8.9449, -4.0845), light=(1.8876, 3.0296, -0.2640, 10, 10, 10),
                 bbox=((-2.5499, -1.8838, -2.3675), (6.1486, 1.8838, 6.6203))),
    "4boxes": dict(ntri=48, stats=(1, 0, 1, 0, 0, 48), cam=(7.3589, 4.9583, -6.9258), light=(1.8876, 1.4339, -4.1944, 10, 10, 10),
                   bbox=((-1, -1, -1), (5.0735, 1, 5.1003))),
    "thai2": dict(ntri=20049, stats=(2177, 272, 1905, 596, 6, 38423), cam=(7.3589, 4.9583, -6.9258), light=(4.0762, 3.6388, -4.3055, 1, 1, 1),
                  bbox=((-7.8753, -9.7007, -0.9656), (4.9845, 5.1653, 13.9980))),
}


@pytest.mark.parametrize("name", sorted(FACTS))
def test_scene_and_octree_facts(oracle, scenes, name):
    f = FACTS[name]
    sc = scenes(name)
    assert sc["tri_geom"].size == f["ntri"]
    v = sc["tri_verts"].reshape(-1, 3)
    assert np.allclose(v.min(0), f["bbox"][0], atol=1e-4) and np.allclose(v.max(0), f["bbox"][1], atol=1e-4)
    assert np.allclose(sc["camera_matrix"][12:15], f["cam"], atol=1e-4)
    assert np.allclose(sc["lights"][0], f["light"], atol=1e-4)
    assert abs(float(sc["camera_fov"]) - 39.59775) < 1e-5
    orc = oracle.Oracle(sc, 32, 32)                       # octree at DEFAULT_TRIANGLES_PER_LEAF = 70
    st = orc.octree_stats()
    assert (st["nodes"], st["inner"], st["leaves"], st["empty"], st["depth"], st["tri_refs"]) == f["stats"]
    assert st["max_leaf"] <= 70 or st["depth"] == 9


# ---- pieces without a reference test: hand-computed answers ---------------------------------------
def test_moller_trumbore_known_answers(oracle):
    tri = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    hit, tuv = oracle.moller_trumbore([0.25, 0.25, 1, 0, 0, -1], tri)
    assert hit and np.array_equal(tuv, np.array([1.0, 0.25, 0.25], np.float32))
    hit, tuv = oracle.moller_trumbore([0.25, 0.25, -1, 0, 0, 2], tri)      # back face, unnormalised dir: t in units of dir
    assert hit and np.array_equal(tuv, np.array([0.5, 0.25, 0.25], np.float32))
    assert not oracle.moller_trumbore([0.75, 0.75, 1, 0, 0, -1], tri)[0]     # u + v > 1
    assert not oracle.moller_trumbore([0.25, 0.25, 1, 0, 0, 1], tri)[0]      # t < 0
    assert not oracle.moller_trumbore([0.25, 0.25, 1, 1, 0, 0], tri)[0]      # parallel: |det| < EPSILON
    assert oracle.moller_trumbore([0.25, 0.25, 0, 0, 0, -1], tri)[0]         # t == 0 is a hit
    assert oracle.moller_trumbore([0.0, 0.5, 1, 0, 0, -1], tri)[0]           # u == 0 edge is inside
    assert oracle.moller_trumbore([0.5, 0.5, 1, 0, 0, -1], tri)[0]           # u + v == 1 edge is inside
    small = [0, 0, 0, 3e-4, 0, 0, 0, 3e-4, 0]                                # det = 9e-8 < f32::EPSILON: unhittable
    assert not oracle.moller_trumbore([1e-4, 1e-4, 1, 0, 0, -1], small)[0]


def test_tonemap_and_pack(oracle):
    assert oracle.tonemap_pack(0, 0, 0) == 0xFF000000
    assert oracle.tonemap_pack(1, 1, 1) == 0xFF7F7F7F                        # 0.5 * 255 = 127.5 -> truncates to 127
    assert oracle.tonemap_pack(3, 1, 0) == (0xFF000000 | (191 << 16) | (127 << 8))
    assert oracle.tonemap_pack(1e30, 1e30, 1e30) == 0xFFFFFFFF
    assert oracle.tonemap_pack(float("inf"), 0, 0) == 0xFFFF0000             # inf/inf = NaN -> min(NaN,1) = 1
    assert oracle.tonemap_pack(float("nan"), float("nan"), float("nan")) == 0xFFFFFFFF   # unsampled pixel: white
    assert oracle.tonemap_pack(-0.5, 0, 0) == 0xFF000000                     # negative clamps to 0


def test_pow32_matches_libm_powf(oracle):
    """x.powf(32.0) restated as five f64 squarings: agrees with a correctly rounded power and (to 1 ulp)
    with this libm's powf, for the negative bases mod.rs:255 routinely feeds it too."""
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-1.2, 1.2, 20000), [0.0, -0.0, 1.0, -1.0]]).astype(np.float32)
    got = np.array([oracle.pow32(x) for x in xs], np.float32)
    exact = (xs.astype(np.float64) ** 32).astype(np.float32)
    assert np.array_equal(bits(got), bits(exact))
    libm = np.array([np.float32(math.pow(float(x), 32.0)) for x in xs], np.float32)    # f64 pow rounded once == f32 powf here
    assert np.array_equal(bits(got), bits(libm))


def test_counter_rng_reference_implementation(oracle):
    """pcg4d (Jarzynski & Olano 2020) against a straight Python restatement."""
    def pcg4d(v):
        M = 0xFFFFFFFF
        v = [(x * 1664525 + 1013904223) & M for x in v]
        v[0] = (v[0] + v[1] * v[3]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M; v[3] = (v[3] + v[1] * v[2]) & M
        v = [x ^ (x >> 16) for x in v]
        v[0] = (v[0] + v[1] * v[3]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M; v[3] = (v[3] + v[1] * v[2]) & M
        return v
    rng = np.random.default_rng(3)
    for _ in range(200):
        v = [int(x) for x in rng.integers(0, 2 ** 32, 4)]
        assert list(oracle.pcg4d(v)) == pcg4d(v)
    assert list(oracle.pcg4d([0, 0, 0, 0])) == pcg4d([0, 0, 0, 0])


def test_sample_table_is_uniform_on_the_sphere(oracle, scenes):
    t = oracle.Oracle(scenes("4boxes"), 8, 8, seed=1).sample_table().astype(np.float64)
    assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-6)
    assert np.abs(t.mean(0)).max() < 0.01                                    # sigma = 1/sqrt(3*65536) = 0.00226
    assert np.abs((t * t).mean(0) - 1 / 3).max() < 0.01
    t2 = oracle.Oracle(scenes("4boxes"), 8, 8, seed=2).sample_table()
    assert not np.array_equal(t2, t.astype(np.float32))


def test_camera_ray_formula(oracle, scenes):
    """camera.rs:80-90 by hand: dir = (dx, -dy, 1, 1) * R, pos = last row of the orientation matrix."""
    sc = scenes("ico2")
    w, h = 200, 100
    orc = oracle.Oracle(sc, w, h)
    rot, orient, mx = orc.camera_matrices()
    assert mx[0] == mx[1] == np.float32(math.tan(np.float32(0.5) * (np.float32(39.59775) * np.float32(math.pi) / np.float32(180.0))))
    r = orc.get_ray(50, 20, 0.25, 0.75)
    f = np.float32
    dx = -mx[0] + f(2.0) * mx[0] * ((f(50) + f(0.25)) / f(w))
    dy = -mx[1] + f(2.0) * mx[1] * ((f(20) + f(0.75)) / f(h))
    v = np.array([dx, -dy, 1, 1], np.float32)
    expect = np.array([((v[0] * rot[j] + v[1] * rot[4 + j]) + v[2] * rot[8 + j]) + v[3] * rot[12 + j] for j in range(3)], np.float32)
    assert np.array_equal(bits(r[3:]), bits(expect))
    assert np.array_equal(r[:3], orient[12:15])
    assert np.array_equal(orient[12:15], sc["camera_matrix"][12:15])        # no move yet: camera position from the file


def test_row_index_quirk_and_fix(oracle, scenes):
    """mod.rs:93-96: v = idx / height (not width).  Identical to the fixed mapping iff width == height."""
    sc = scenes("ico2")
    w, h = 60, 30
    quirk = oracle.Oracle(sc, w, h, seed=1)
    fixed = oracle.Oracle(sc, w, h, seed=1, flags=oracle.FLAG_FIX_ROW_INDEX)
    pixel = 7 * w + 11
    rq = quirk.primary_ray(pixel, 0); rf = fixed.primary_ray(pixel, 0)
    assert not np.array_equal(rq, rf)
    # the same jitter, only v differs: recompute both from get_ray with u = idx % w and v = idx / h | idx / w
    u, vq, vf = pixel % w, pixel // h, pixel // w
    h4 = oracle.pcg4d([pixel, 0, 0, 1])
    xi = [np.float32((int(x) >> 9) * (1.0 / 8388608.0)) for x in h4[:2]]
    assert np.array_equal(bits(rq), bits(quirk.get_ray(u, vq, xi[0], xi[1])))
    assert np.array_equal(bits(rf), bits(quirk.get_ray(u, vf, xi[0], xi[1])))
    sq = oracle.Oracle(sc, 32, 32, seed=1); sf = oracle.Oracle(sc, 32, 32, seed=1, flags=oracle.FLAG_FIX_ROW_INDEX)
    assert np.array_equal(sq.primary_ray(500, 3), sf.primary_ray(500, 3))


def test_film_and_progressive_frames(oracle, scenes):
    """trace_frame_additive: 50 rows per call, cursor wraps, returns 50*w; film.rs mean / variance readout."""
    sc = scenes("4boxes")
    w, h = 24, 70
    orc = oracle.Oracle(sc, w, h, seed=3)
    assert orc.trace_frame_additive() == 50 * w and orc.current_row == 50
    _, _, n = orc.film()
    assert n.reshape(h, w)[:50].min() == 1 and n.reshape(h, w)[50:].max() == 0
    ldr = orc.get_tonemapped_pixels().reshape(h, w)
    assert np.all(ldr[50:] == 0xFFFFFFFF)                                   # unsampled rows: NaN -> white
    assert orc.trace_frame_additive() == 50 * w and orc.current_row == 30
    s, q, n = orc.film()
    assert n.reshape(h, w)[:30].min() == 2 and n.reshape(h, w)[30:50].max() == 1
    mean = orc.get_pixels()
    assert np.array_equal(bits(mean), bits(s * (np.float32(1.0) / n.astype(np.float32))[:, None]))
    var = orc.get_estimated_variances().reshape(h, w, 3)[:30]                # n == 2 rows
    s2 = s.reshape(h, w, 3)[:30]; q2 = q.reshape(h, w, 3)[:30]
    expect = (q2 / np.float32(2.0) - s2 * s2 / np.float32(4.0)) * np.float32(50.0)
    assert np.array_equal(bits(var), bits(expect))
    orc.film_clear()
    assert orc.film()[2].max() == 0 and orc.current_row == 30               # clear() keeps the row cursor


def test_radiance_tree_weights(oracle, scenes):
    """compute_radiance: colour = L0 + ((0 + C1) + C2) * 0.5 with C = L + (0 + G) * 1.0 (mod.rs:154-175)."""
    sc = scenes("ico2")
    orc = oracle.Oracle(sc, 64, 64, seed=5, flags=oracle.FLAG_BRUTE_FORCE)
    done = 0
    for pixel in range(0, 4096, 37):
        c, L, hit = orc.sample_debug(pixel, 1)
        if not hit[0]:
            assert not c.any()
            continue
        f = np.float32
        c1 = L[1] + (f(0) + L[3]) * f(1.0); c2 = L[2] + (f(0) + L[4]) * f(1.0)
        expect = L[0] + ((np.zeros(3, f) + c1) + c2) * f(0.5)
        assert np.array_equal(bits(c), bits(expect))
        done += 1
    assert done > 20


# ---- committed golden renders -----------------------------------------------------------------------
@pytest.mark.parametrize("name", ["4boxes", "ico2", "thai2", "ico3_tex"])
def test_oracle_reproduces_golden(oracle, scenes, name):
    g = np.load(os.path.join(GOLDEN, "render_%s.npz" % name))
    for mode, flags in (("octree", 0), ("brute", oracle.FLAG_BRUTE_FORCE)):
        orc = oracle.Oracle(scenes(name), 64, 64, seed=1, flags=flags)
        c = orc.render(4, nthreads=8)
        s, q, n = orc.film()
        assert np.array_equal(bits(s), bits(g[mode + "_sum"])) and np.array_equal(bits(q), bits(g[mode + "_sumsq"]))
        assert np.array_equal(n, g[mode + "_n"]) and np.array_equal(orc.get_tonemapped_pixels(), g[mode + "_ldr"])
        assert [c["primary"], c["bounce"], c["shadow"], c["primary_hits"]] == [int(x) for x in g[mode + "_counts"]]
    orc = oracle.Oracle(scenes(name), 64, 64, seed=1)
    for brute in (False, True):
        tuv, prim = orc.intersect(g["rays"], brute=brute)
        key = "brute" if brute else "octree"
        assert np.array_equal(prim, g[key + "_prim"])
        m = prim != 0xFFFFFFFF
        assert np.array_equal(bits(tuv[m]), bits(g[key + "_tuv"][m]))


def test_octree_vs_true_closest_hit_delta_is_the_documented_one():
    """The reference's octree accepts a leaf's closest triangle only if the hit point lies inside the
    leaf cube (OCT:160-169).  With the bundled scenes this changes results only for 4boxes, whose
    octree is a single leaf equal to the scene bounds with box faces ON those bounds."""
    for name, lo, hi in (("ico2", 0.0, 0.0), ("thai2", 0.0, 0.0), ("ico3_tex", 0.0, 0.0), ("4boxes", 0.01, 0.12)):
        g = np.load(os.path.join(GOLDEN, "render_%s.npz" % name))
        frac = float((g["octree_ldr"] != g["brute_ldr"]).mean())
        assert lo <= frac <= hi, (name, frac)
        if name == "4boxes":
            # every differing hit record is a hit the octree DROPPED (cube check), never a different triangle
            d = g["octree_prim"] != g["brute_prim"]
            assert d.any() and np.all(g["octree_prim"][d] == 0xFFFFFFFF)
