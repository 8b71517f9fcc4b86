"""GPU tests of the entries the reference binary actually calls (main.rs:197-207: trace_frame_additive +
get_tonemapped_pixels, served by the single-launch fused kernel and the dirty-row read-out) and of the parity
gaps that have reference-held or oracle data: the reference's slab known-answer vectors THROUGH the HIP path,
create_raytracer(collada_doc) end to end, the variance read-out, BASELINE config 1 (ico2 256x256x1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def make(pkg, scenes, name, w, h, **kw):
    return pkg.create_raytracer_from_arrays(scenes(name), kw.pop("triangles_per_leaf", pkg.DEFAULT_TRIANGLES_PER_LEAF), w, h, **kw)


# ---- (a) oct_tree_intersector.rs:475-512 through the device ------------------------------------------------
CUBE = [-1, -1, -1, 1, 1, 1]


def inv(d):
    with np.errstate(divide="ignore"):
        return list(np.float32(1.0) / np.asarray(d, np.float32))


def test_reference_slab_kats_through_hip(pkg, scenes, oracle):
    """The four known-answer vectors of the reference's own tests, run by the kernel code of the reference-exact
    intersector (cube_slab in csrc/traverse.hpp): t == 1.0 exactly, inf by design, t < 0 inside, miss behind."""
    rt = make(pkg, scenes, "4boxes", 16, 16)
    rays = [[2, 0, 0] + inv([-1, 0.1, 0.1]),          # test_intersect_cube_inverse_ray
            [2, 0, 0] + inv([-1, 0.0, 0.0]),          # ..._handles_inf
            [-0.9, 0, 0] + inv([1, 0.1, 0.1]),        # ..._start_inside
            [-2, 0, 0] + inv([-1, 0.1, 0.1])]         # ..._should_miss
    hit, t = rt.debug_slab(rays, [CUBE] * 4)
    assert list(hit) == [True, True, True, False]
    assert t[0] == 1.0 and t[1] == 1.0 and t[2] < 0.0
    # and bit-identical to the oracle's restatement on random boxes / rays, axis-parallel directions included
    rng = np.random.default_rng(4)
    n = 20000
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    lo = rng.uniform(-2, 1, (n, 3)).astype(np.float32); hi = lo + rng.uniform(0, 2, (n, 3)).astype(np.float32)
    d = (0.5 * (lo + hi) + rng.normal(scale=0.7, size=(n, 3)) - o).astype(np.float32)      # aimed near the box
    d[rng.random((n, 3)) < 0.1] = 0.0
    with np.errstate(divide="ignore"):
        r6 = np.concatenate([o, np.float32(1.0) / d], axis=1).astype(np.float32)
    c6 = np.concatenate([lo, hi], axis=1)
    ghit, gt = rt.debug_slab(r6, c6)
    for i in range(0, n, 7):
        oh, ot = oracle.slab(r6[i], c6[i])
        assert oh == ghit[i]
        if oh:                                                    # Option<f32>: the distance exists for hits only
            assert bits(np.float32(ot))[()] == bits(gt[i:i + 1])[0]
    assert 0.05 < ghit.mean() < 0.95


# ---- (b) create_raytracer(collada_doc, ...) end to end -------------------------------------------------------
def test_create_raytracer_from_collada_doc_renders_like_the_oracle(pkg, oracle):
    """lib.rs:15-20 on the device: the inline document of tests/test_loader.py goes through the library's own
    COLLADA reader, BVH build and kernels; the film equals the oracle fed the HAND-WRITTEN arrays of that
    document (one triangle, one light, one camera; collada_types.rs:76-90 conversion)."""
    from test_loader import DOC, to_vecmath, xform
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    m_tri = to_vecmath(ident)
    verts = np.concatenate([xform(m_tri, v) for v in ([0, 0, 0], [1, 0, 0], [0, 1, 0])]).reshape(1, 9)
    m_light = to_vecmath(np.array([1, 0, 0, 1, 0, 1, 0, 2, 0, 0, 1, 3, 0, 0, 0, 1], np.float32))
    m_cam = to_vecmath(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 5, 0, 0, 0, 1], np.float32))
    light = np.concatenate([xform(m_light, [0, 0, 0]), np.array([10, 10, 10], np.float32)]).reshape(1, 6)
    assert np.array_equal(light[0, :3], np.array([1, 3, 2], np.float32))        # COLLADA (x, y, z) -> (x, z, y), collada_types.rs:98-109
    scene = dict(tri_verts=verts.astype(np.float32), tri_geom=np.zeros(1, np.uint32), mat_kind=np.zeros(1, np.uint32),
                 mat_rgb=np.array([[0.8, 0.1, 0.2]], np.float32), mat_tex=np.zeros(1, np.uint32), lights=light.astype(np.float32), textures=[],
                 camera_matrix=m_cam, camera_fov=np.float32(39.59775))
    w, h, spp = 96, 96, 3
    for flags, oflags in ((0, 0), (pkg.FLAG_TRUE_CLOSEST_HIT, oracle.FLAG_BRUTE_FORCE)):
        rt = pkg.create_raytracer(DOC, pkg.DEFAULT_TRIANGLES_PER_LEAF, w, h, seed=2, flags=flags)
        assert rt.triangle_count == 1
        orc = oracle.Oracle(scene, w, h, seed=2, flags=oflags)
        c = rt.render(spp); oc = orc.render(spp, nthreads=4)
        assert (c.primary, c.bounce, c.shadow, c.primary_hits) == (oc["primary"], oc["bounce"], oc["shadow"], oc["primary_hits"])
        assert c.primary_hits > 50                                                   # the camera does see the triangle
        gs, gq, gn = rt.film.pixel_datas(); os_, oq, on = orc.film()
        assert np.array_equal(gn, on) and np.array_equal(bits(gs), bits(os_)) and np.array_equal(bits(gq), bits(oq))
        assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    # the direct octree walk on the same document
    rto = pkg.create_raytracer(DOC, pkg.DEFAULT_TRIANGLES_PER_LEAF, w, h, seed=2, flags=pkg.FLAG_OCTREE_SEMANTICS)
    orco = oracle.Oracle(scene, w, h, seed=2)
    rto.render(spp); orco.render(spp, nthreads=4)
    assert np.array_equal(bits(rto.film.pixel_datas()[0]), bits(orco.film()[0]))


# ---- (c) film.rs:51-67 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,spp", [("ico2", 5), ("ico3_tex", 2), ("thai2", 1)])
def test_variance_and_mean_readout_bit_equal_to_oracle(pkg, scenes, oracle, name, spp):
    """Film::get_estimated_variances and Film::get_pixels, computed on the device: bit-equal to the oracle,
    including the n == 1 case (0/0 = NaN and x/0 = inf exactly where the reference produces them)."""
    w, h = 80, 56
    rt = make(pkg, scenes, name, w, h, seed=3)
    orc = oracle.Oracle(scenes(name), w, h, seed=3)
    rt.render(spp); orc.render(spp, nthreads=8)
    with np.errstate(all="ignore"):
        gv, ov = rt.film.get_estimated_variances(), orc.get_estimated_variances()
        assert np.array_equal(bits(gv), bits(ov))
        assert np.array_equal(bits(rt.film.get_pixels()), bits(orc.get_pixels()))
        if spp > 1:
            assert np.isfinite(gv).all() and gv.max() > 0


# ---- (d) BASELINE config 1 ------------------------------------------------------------------------------------
def test_baseline_config_1_ico2_256x256_1spp(pkg, scenes, oracle):
    """ico2 256x256 x 1 spp — the one BASELINE config where idx / height == idx / width: GPU == oracle in every
    intersector mode (default = the reference's octree semantics; true closest hit vs brute force; direct octree walk), and
    through the reference's own entry (six 50-row frames sweep the 256 rows once and wrap)."""
    w = h = 256
    sc = scenes("ico2")
    for flags, oflags in ((0, 0), (pkg.FLAG_TRUE_CLOSEST_HIT, oracle.FLAG_BRUTE_FORCE), (pkg.FLAG_OCTREE_SEMANTICS, 0)):
        rt = make(pkg, scenes, "ico2", w, h, seed=1, flags=flags)
        orc = oracle.Oracle(sc, w, h, seed=1, flags=oflags)
        c = rt.render(1); oc = orc.render(1, nthreads=8)
        assert c.primary == w * h == 65536 and (c.bounce, c.shadow, c.primary_hits) == (oc["bounce"], oc["shadow"], oc["primary_hits"])
        assert 0.45 < c.primary_hits / c.primary < 0.57                              # 0.51 at 256x256
        gs, gq, gn = rt.film.pixel_datas(); os_, oq, on = orc.film()
        assert np.all(gn == 1) and np.array_equal(gn, on)
        assert np.array_equal(bits(gs), bits(os_)) and np.array_equal(bits(gq), bits(oq))
        assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    # quirk-free: fixing the row index changes nothing when width == height
    fixed = make(pkg, scenes, "ico2", w, h, seed=1, flags=pkg.FLAG_FIX_ROW_INDEX)
    fixed.render(1)
    plain = make(pkg, scenes, "ico2", w, h, seed=1)
    plain.render(1)
    assert np.array_equal(bits(fixed.film.pixel_datas()[0]), bits(plain.film.pixel_datas()[0]))
    # the drop-in loop on C1
    loop = make(pkg, scenes, "ico2", w, h, seed=1)
    orc = oracle.Oracle(sc, w, h, seed=1)
    for _ in range(6):
        assert loop.trace_frame_additive() == orc.trace_frame_additive() == 50 * w
    assert loop.current_row == orc.current_row == (6 * 50) % h
    assert np.array_equal(loop.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    assert np.array_equal(bits(loop.film.pixel_datas()[0]), bits(orc.film()[0]))


# ---- the drop-in loop at the reference binary's defaults ----------------------------------------------------------
def test_dropin_loop_thai2_1024x768_matches_oracle(pkg, scenes, oracle):
    """main.rs:13-15,197-207: thai2 at 1024x768, trace_frame_additive + get_tonemapped_pixels, 17 calls (one sweep
    of the 768 rows plus a wrap).  Pixels after every call, the film at the end and the ray counters of a call
    equal the oracle's; rows not sampled yet read back white."""
    w, h = 1024, 768
    rt = make(pkg, scenes, "thai2", w, h, seed=1)
    orc = oracle.Oracle(scenes("thai2"), w, h, seed=1)                       # both sides: the reference's default intersector
    buf = np.empty(w * h, np.uint32)
    for call in range(17):
        assert rt.trace_frame_additive() == 50 * w
        before = orc.counters()
        assert orc.trace_frame_additive() == 50 * w
        after = orc.counters()
        if call in (0, 7, 16):
            c = rt.last_counts()
            assert (c.primary, c.bounce, c.shadow, c.primary_hits) == tuple(after[k] - before[k] for k in ("primary", "bounce", "shadow", "primary_hits"))
            assert c.trace_launches == 1
        g = rt.get_tonemapped_pixels(buf)
        if call in (0, 1, 8, 15, 16):
            assert np.array_equal(g, orc.get_tonemapped_pixels()), call
        if call == 0:
            assert np.all(g[50 * w:] == 0xFFFFFFFF)
    assert rt.current_row == orc.current_row == (17 * 50) % h
    gs, gq, gn = rt.film.pixel_datas(); os_, oq, on = orc.film()
    assert gn.max() == 2 and np.array_equal(gn, on)
    assert np.array_equal(bits(gs), bits(os_)) and np.array_equal(bits(gq), bits(oq))
    # a camera move: film.clear() makes every row white again until it is re-sampled (main.rs:116-169)
    rt.camera.move_rel(0.1, 0.0, 0.0); orc.camera_move_rel(0.1, 0.0, 0.0)
    rt.film.clear(); orc.film_clear()
    rt.trace_frame_additive(); orc.trace_frame_additive()
    g = rt.get_tonemapped_pixels(buf)
    assert np.array_equal(g, orc.get_tonemapped_pixels())
    assert (g == 0xFFFFFFFF).mean() > 0.9


@pytest.mark.parametrize("speculate", [True, False])
def test_speculated_frames_are_taken_over_or_given_back(pkg, scenes, oracle, monkeypatch, speculate):
    """The library launches the NEXT 50-row frame behind the one asked for (the reference's loop alternates trace_frame_additive and
    get_tonemapped_pixels, main.rs:197-207).  Whatever the caller does instead of the predicted call must see the film as if nothing had been
    launched ahead: a film read-out, a camera move WITHOUT a clear, a clear, a new seed, a whole-frame render, the row-index flag, a call right after
    creation — each followed by more frames; film, pixels, current row and the per-call counters equal the oracle's after every step.  The same
    sequence with MI355RT_NO_SPECULATE is the control."""
    if not speculate:
        monkeypatch.setenv("MI355RT_NO_SPECULATE", "1")
    else:
        monkeypatch.delenv("MI355RT_NO_SPECULATE", raising=False)
    w, h = 320, 240
    name = "ico2"
    rt = make(pkg, scenes, name, w, h, seed=3)
    orc = oracle.Oracle(scenes(name), w, h, seed=3)
    buf = np.empty(w * h, np.uint32)

    def frame(check_counts=True):
        before = orc.counters()
        assert rt.trace_frame_additive() == orc.trace_frame_additive() == 50 * w
        after = orc.counters()
        if check_counts:
            c = rt.last_counts()
            assert (c.primary, c.bounce, c.shadow, c.primary_hits) == tuple(after[k] - before[k] for k in ("primary", "bounce", "shadow", "primary_hits"))

    def same_film():
        gs, gq, gn = rt.film.pixel_datas(); os_, oq, on = orc.film()
        assert np.array_equal(gn, on) and np.array_equal(bits(gs), bits(os_)) and np.array_equal(bits(gq), bits(oq))
        assert rt.current_row == orc.current_row

    def same_pixels():
        assert np.array_equal(rt.get_tonemapped_pixels(buf), orc.get_tonemapped_pixels())

    frame(); same_pixels(); frame(False); same_pixels(); frame()
    same_film()                                                  # a film read-out while the next frame is out
    frame(); same_pixels()
    rt.camera.move_rel(0.2, -0.1, 0.1); orc.camera_move_rel(0.2, -0.1, 0.1)          # the camera moves, the film stays (not what the reference's loop does, but allowed)
    frame(); same_pixels(); same_film()
    frame(False); frame(False); same_pixels()                    # frames without read-outs in between
    rt.film.clear(); orc.film_clear()
    same_pixels()                                                # everything white
    frame(); same_pixels(); same_film()
    rt.set_seed(77); orc.set_seed(77)
    frame(); frame(); same_film()
    rt.render(2); orc.render(2, nthreads=4)                      # a whole-frame render in between
    same_film(); frame(); same_pixels(); same_film()
    rt.set_flags(pkg.FLAG_FIX_ROW_INDEX); orc.set_flags(oracle.FLAG_FIX_ROW_INDEX)
    frame(); same_pixels(); frame(); same_film()
    for _ in range(6):                                           # through the wrap of the row cursor
        frame(False)
    same_pixels(); same_film()
    launched, taken = rt.debug_speculation()
    if speculate:
        assert launched > 10 and 5 < taken < launched           # both outcomes happened
    else:
        assert (launched, taken) == (0, 0)


def test_fused_frame_equals_wavefront_rounds(pkg, scenes):
    """The single-launch frame kernel and the multi-launch wavefront rounds are the same arithmetic: identical
    films and counters — also with stripes (rank 1 of 3 owns every third block of 4 rows, so a 50-row window holds
    16-18 owned rows and wraps), other recursion settings, a texture, and height < 50."""
    cases = [("ico2", 200, 120, {}), ("ico3_tex", 96, 70, {}), ("ico2", 120, 130, dict(stripe_rows=4, stripe_rank=1, stripe_world=3)),
             ("4boxes", 64, 20, {}), ("ico2", 64, 64, dict(recursions=3, spread=1)), ("ico2", 64, 64, dict(recursions=1, spread=2)),
             ("ico2", 33, 7, dict(stripe_rows=2, stripe_rank=0, stripe_world=2)),
             ("4boxes", 96, 64, dict(flags=32)), ("ico2", 96, 64, dict(flags=32)), ("thai2", 160, 100, {})]      # 32 = FLAG_TRUE_CLOSEST_HIT
    for name, w, h, kw in cases:
        a = make(pkg, scenes, name, w, h, seed=6, **kw)
        b = make(pkg, scenes, name, w, h, seed=6, **kw)
        for call in range(5):
            assert a.trace_frame_additive() == 50 * w
            ca = a.last_counts()
            os.environ["MI355RT_NO_FUSED"] = "1"
            try:
                assert b.trace_frame_additive() == 50 * w
            finally:
                del os.environ["MI355RT_NO_FUSED"]
            cb = b.last_counts()
            assert (ca.primary, ca.bounce, ca.shadow, ca.primary_hits) == (cb.primary, cb.bounce, cb.shadow, cb.primary_hits), (name, w, h, call)
            assert ca.trace_launches < cb.trace_launches
            assert np.array_equal(a.get_tonemapped_pixels(), b.get_tonemapped_pixels()), (name, w, h, call)
        sa, qa, na = a.film.pixel_datas(); sb, qb, nb = b.film.pixel_datas()
        assert np.array_equal(na, nb) and na.sum() > 0
        assert np.array_equal(bits(sa), bits(sb)) and np.array_equal(bits(qa), bits(qb)), (name, w, h)
        assert a.current_row == b.current_row


def test_mixing_render_and_frames_keeps_the_readout_current(pkg, scenes, oracle):
    """get_tonemapped_pixels only re-maps rows written since the last read-out: interleave render(), 50-row
    frames, clears and read-outs and compare every read-out with the oracle's full-frame mapping."""
    name, w, h = "ico2", 72, 130
    rt = make(pkg, scenes, name, w, h, seed=9)
    orc = oracle.Oracle(scenes(name), w, h, seed=9)
    assert np.all(rt.get_tonemapped_pixels() == 0xFFFFFFFF)
    rt.trace_frame_additive(); orc.trace_frame_additive()
    assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())      # nothing dirty: same frame again
    rt.render(2); orc.render(2, nthreads=4)
    rt.trace_frame_additive(); orc.trace_frame_additive()
    rt.trace_frame_additive(); orc.trace_frame_additive()
    assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    rt.film.clear(); orc.film_clear()
    assert np.all(rt.get_tonemapped_pixels() == 0xFFFFFFFF)
    rt.trace_frame_additive(); orc.trace_frame_additive()
    assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    assert np.array_equal(bits(rt.film.pixel_datas()[0]), bits(orc.film()[0]))


def test_tonemap_into_device_memory_on_the_callers_stream(pkg, scenes):
    """mi355rt_tonemap_owned_rows_device_on_stream: the packed stripes land in a torch buffer, ordered with the
    work the caller queued on ITS stream before (a fill) and after (a copy), without a host synchronisation."""
    import torch
    name, w, h = "ico2", 64, 48
    rt = make(pkg, scenes, name, w, h, seed=3, stripe_rows=8, stripe_rank=1, stripe_world=2)
    rt.render(2)
    rows = rt.owned_rows()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        buf = torch.full((rows.size * w,), 7, dtype=torch.int32, device="cuda")       # queued BEFORE the tonemap on the same stream
        rt.tonemap_owned_rows_device(buf.data_ptr(), rows.size * w, stream=side.cuda_stream)
        copy = buf.clone()                                                              # queued AFTER it
    side.synchronize()
    expect = rt.get_tonemapped_pixels().reshape(h, w)[rows].reshape(-1)
    assert np.array_equal(copy.cpu().numpy().view(np.uint32), expect)
    # the next frame of the handle waits for that read of the film
    rt.film.clear(); rt.render(1)
    sync = torch.empty(rows.size * w, dtype=torch.int32, device="cuda")
    rt.tonemap_owned_rows_device(sync.data_ptr(), rows.size * w)                        # synchronous variant
    assert np.array_equal(sync.cpu().numpy().view(np.uint32), rt.get_tonemapped_pixels().reshape(h, w)[rows].reshape(-1))


def test_set_seed_and_set_flags_semantics(pkg, scenes, oracle):
    """mi355rt_set_seed: a re-seeded handle renders what a handle created with that seed renders (hash key AND
    direction table), like the oracle.  mi355rt_set_flags: run-time bits only — the create-time intersector bit
    cannot be flipped on a live handle, and setting other bits does not disturb it."""
    name, w, h = "ico2", 64, 48
    a = make(pkg, scenes, name, w, h, seed=1)
    a.set_seed(5)
    b = make(pkg, scenes, name, w, h, seed=5)
    assert np.array_equal(bits(a.sample_table()), bits(b.sample_table()))
    a.render(2); b.render(2)
    assert np.array_equal(bits(a.film.pixel_datas()[0]), bits(b.film.pixel_datas()[0]))
    orc = oracle.Oracle(scenes(name), w, h, seed=1)
    orc.set_seed(5); orc.render(2, nthreads=4)
    assert np.array_equal(bits(a.film.pixel_datas()[0]), bits(orc.film()[0]))
    # flags
    with pytest.raises(RuntimeError, match="create-time"):
        a.set_flags(pkg.FLAG_OCTREE_SEMANTICS)
    with pytest.raises(RuntimeError, match="create-time"):
        a.set_flags(pkg.FLAG_TRUE_CLOSEST_HIT)
    o = make(pkg, scenes, "4boxes", w, h, seed=2, flags=pkg.FLAG_OCTREE_SEMANTICS)
    with pytest.raises(RuntimeError, match="create-time"):
        o.set_flags(pkg.FLAG_COUNT_STEPS)                                    # would clear the octree bit
    o.set_flags(pkg.FLAG_OCTREE_SEMANTICS | pkg.FLAG_FIX_ROW_INDEX)          # run-time bit changes, create-time bit kept
    oo = oracle.Oracle(scenes("4boxes"), w, h, seed=2, flags=oracle.FLAG_FIX_ROW_INDEX)
    o.render(2); oo.render(2, nthreads=4)
    assert np.array_equal(bits(o.film.pixel_datas()[0]), bits(oo.film()[0]))   # still the reference-exact octree, 4boxes shows it


def test_no_launch_writes_past_a_pass_buffer(pkg, scenes, monkeypatch):
    """Guard regions (MI355RT_DEBUG_GUARD: 256 bytes of 0xA5 behind every pass buffer).  A FRESH handle whose first call is
    trace_frame_additive sizes its buffers for 50 rows and then cuts them into chunks of 16 / 32 / 64 samples for the fused
    launch (the per-chunk count arrays were once sized for 256-sample chunks only); then whole frames on the same handle
    (buffers grow), then 50-row frames again through the wavefront rounds.  No guard byte may change, and the films of the
    chunk sizes are identical."""
    monkeypatch.setenv("MI355RT_DEBUG_GUARD", "1")
    films = []
    for fchunk in ("16", "32", "64"):
        monkeypatch.setenv("MI355RT_FUSED_CHUNK", fchunk)
        rt = make(pkg, scenes, "thai2", 1024, 768, seed=4)
        for _ in range(3):
            assert rt.trace_frame_additive() == 50 * 1024
        rt.last_counts()
        assert rt.debug_check_guards() == 0, fchunk
        films.append(rt.film.pixel_datas())
        rt.render(2)
        assert rt.debug_check_guards() == 0, fchunk
        monkeypatch.setenv("MI355RT_NO_FUSED", "1")
        rt.trace_frame_additive(); rt.last_counts()
        monkeypatch.delenv("MI355RT_NO_FUSED")
        assert rt.debug_check_guards() == 0, fchunk
        assert rt.hbm_allocated_bytes() > 0
        del rt
    for s, q, n in films[1:]:
        assert np.array_equal(n, films[0][2]) and np.array_equal(bits(s), bits(films[0][0])) and np.array_equal(bits(q), bits(films[0][1]))
    # multi-slice frame: every slice's buffers are guarded
    monkeypatch.delenv("MI355RT_FUSED_CHUNK")
    rt = make(pkg, scenes, "ico2", 640, 360, seed=4)
    rt.set_slices(3)
    rt.render(4)
    assert rt.debug_check_guards() == 0


def test_render_async_is_the_same_frame_queued(pkg, scenes):
    """mi355rt_render_async queues the frame and returns; last_counts / any read-out waits.  Two queued frames + a read-out give the
    film and the counters of two synchronous frames, bit for bit; a queued frame followed by film.clear() and another queued frame
    is ordered on the handle's stream."""
    a = make(pkg, scenes, "ico2", 320, 200, seed=3)
    b = make(pkg, scenes, "ico2", 320, 200, seed=3)
    ca = [a.render(3), a.render(2)]
    assert b.render(3, wait=False) is None and b.render(2, wait=False) is None
    cb = b.last_counts()                                                   # waits; the counters of the LAST call
    assert (cb.primary, cb.bounce, cb.shadow, cb.primary_hits) == (ca[1].primary, ca[1].bounce, ca[1].shadow, ca[1].primary_hits)
    assert cb.total_ms > 0
    (sa, qa, na), (sb, qb, nb) = a.film.pixel_datas(), b.film.pixel_datas()
    assert np.array_equal(na, nb) and int(na.max()) == 5
    assert np.array_equal(bits(sa), bits(sb)) and np.array_equal(bits(qa), bits(qb))
    b.film.clear(); b.render(4, wait=False); b.film.clear(); b.render(1, wait=False)
    a.film.clear(); a.render(1)
    assert np.array_equal(a.get_tonemapped_pixels(), b.get_tonemapped_pixels())
    assert np.array_equal(bits(a.film.pixel_datas()[0]), bits(b.film.pixel_datas()[0]))


def test_gather_rate_hook_reports_a_plausible_rate(pkg, scenes):
    """mi355rt_debug_gather_rate (bench.py's roofline peak): 64 lanes x 2 loads per wave-step, every lane on its own line.  An MI355X
    serves a few hundred G of those per second; the node rate is half the access rate by construction."""
    rt = make(pkg, scenes, "4boxes", 64, 64, seed=1)
    g = rt.debug_gather_rate(48000, 500)
    assert 5e10 < g["line_accesses_per_s"] < 5e12 and g["ms"] > 0
    assert abs(g["line_accesses_per_s"] / g["node_fetches_per_s"] - 2.0) < 1e-9
    with pytest.raises(RuntimeError):
        rt.debug_gather_rate(10, 10)                                       # table too small
