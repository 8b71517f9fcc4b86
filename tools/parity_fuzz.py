"""Randomised parity: random scene / size / seed / recursion setting / leaf size / semantics / camera moves / call sequence, the HIP
path against the CPU oracle bit for bit (film sums, sums of squares, counts, packed pixels, ray counters).  Test infrastructure.
usage: [FUZZ_WILD=1] [FUZZ_SPP=1] [FUZZ_SOUP=1] [FUZZ_BUILD=1] parity_fuzz.py [cases] [seed]   — prints one line per case, exits 1 on the first difference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge




def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def soup(base, rng):
    sc = dict(base)
    v = base["tri_verts"].reshape(-1, 3); lo, hi = v.min(0), v.max(0); ext = float((hi - lo).max())
    n = int(rng.choice([1, 2, 3, 7, 30, 150, 600]))
    centre = rng.uniform(lo, hi, (n, 1, 3))
    size = ext * 10.0 ** rng.uniform(-4.0, 0.0, (n, 1, 1))
    t = centre + rng.uniform(-1.0, 1.0, (n, 3, 3)) * size
    kind = rng.random(n)
    for i in range(n):
        if kind[i] < 0.10: t[i, 2] = t[i, 1]                                        # two equal vertices
        elif kind[i] < 0.15: t[i, 2] = 0.5 * (t[i, 0] + t[i, 1])                    # three vertices on a line
        elif kind[i] < 0.25 and i: t[i] = t[int(rng.integers(0, i))]                # the same triangle twice (ties go to the lower index)
        elif kind[i] < 0.40: t[i, :, int(rng.integers(0, 3))] = t[i, 0, int(rng.integers(0, 3))]   # axis-parallel
        elif kind[i] < 0.45: t[i] = centre[i] + rng.uniform(-3.0, 3.0, (3, 3)) * ext   # reaches far beyond the rest
        elif kind[i] < 0.50: t[i, 1] = t[i, 0] + (t[i, 1] - t[i, 0]) * 1e-3          # a sliver
    sc["tri_verts"] = t.astype(np.float32).reshape(n, 9)
    sc["tri_geom"] = rng.integers(0, len(base["mat_kind"]), size=n).astype(np.uint32)
    return sc


def one_case(pkg, O, scenes, rng, verbose=True):
    WILD = bool(os.environ.get("FUZZ_WILD"))
    name = rng.choice(["ico2", "4boxes", "ico3_tex", "thai2"], p=[0.35, 0.2, 0.2, 0.25])
    w, h = int(rng.integers(4, 161)), int(rng.integers(3, 121))
    if rng.random() < 0.4:                              # sizes whose wave tiles do not straddle row groups: the tile bins of the primary rays engage
        w, h = 8 * int(rng.integers(1, 21)), 8 * int(rng.integers(1, 16))
    seed = int(rng.integers(0, 2**31))
    rec, spread = [(2, 1), (2, 1), (1, 2), (3, 1), (0, 1), (1, 1), (2, 2)][int(rng.integers(0, 7))]
    tpl = int(rng.choice([1, 5, 20, 70, 100]))
    sem = int(rng.integers(0, 3))                       # 0 reference default, 1 true closest hit, 2 default + fixed row index
    gflags = [0, pkg.FLAG_TRUE_CLOSEST_HIT, pkg.FLAG_FIX_ROW_INDEX][sem]
    oflags = [0, O.FLAG_BRUTE_FORCE, O.FLAG_FIX_ROW_INDEX][sem]
    stripes = {}
    if rng.random() < 0.3:
        world = int(rng.integers(2, 5)); stripes = dict(stripe_rows=int(rng.choice([1, 2, 4, 8])), stripe_rank=int(rng.integers(0, world)), stripe_world=world)
    sc = scenes(name)
    if os.environ.get("FUZZ_SOUP") and rng.random() < 0.6:   # FUZZ_SOUP=1: random triangle soups in front of the 4boxes camera — slivers, degenerate and duplicate
        name = "soup"; sc = soup(scenes("4boxes"), rng)        # triangles, axis-parallel ones, huge ones, sizes over four decades
    if rng.random() < 0.35:                             # the light somewhere else: in, on or around the geometry (the lights' depth maps, the ray's tail behind the light)
        sc = dict(sc); sc["lights"] = sc["lights"].copy()
        v = sc["tri_verts"].reshape(-1, 3); lo, hi = v.min(0), v.max(0)
        kind = rng.random()
        if kind < 0.4: sc["lights"][0, :3] = rng.uniform(lo, hi)
        elif kind < 0.7: sc["lights"][0, :3] = v[int(rng.integers(0, len(v)))] + rng.normal(0, 1e-3, 3)
        else: sc["lights"][0, :3] = rng.uniform(lo - 2 * (hi - lo), hi + 2 * (hi - lo))
    extra = {}
    if os.environ.get("FUZZ_BUILD"):                       # FUZZ_BUILD=1: the tree built on the device (LBVH), several members of a device group sharing the one GPU
        if rng.random() < 0.4: gflags |= pkg.FLAG_DEVICE_LBVH
        if sem != 1 and rng.random() < 0.25: gflags |= pkg.FLAG_OCTREE_SEMANTICS     # the direct octree walk (the reference's own traversal, no BVH): same answers as the default path
        if rng.random() < 0.2: gflags |= pkg.FLAG_COUNT_STEPS                        # the instrumented kernels
        if not stripes and rng.random() < 0.3:
            extra["device_count"] = int(rng.integers(2, 5)); gflags |= pkg.FLAG_GROUP_SHARES_DEVICE
    big_spp = os.environ.get("FUZZ_SPP") and rng.random() < 0.6      # FUZZ_SPP=1: many samples per pixel on small images (sample groups, passes of odd sample counts)
    if big_spp:
        w, h = min(w, 8 * int(rng.integers(1, 9))), min(h, 8 * int(rng.integers(1, 7)))
        if rng.random() < 0.5:
            extra["samples_per_pass"] = int(rng.choice([1, 2, 3, 5, 8, 12]))
    rt = pkg.create_raytracer_from_arrays(sc, tpl, w, h, seed=seed, recursions=rec, spread=spread, flags=gflags, **stripes, **extra)
    orc = O.Oracle(sc, w, h, tris_per_leaf=tpl, recursions=rec, spread=spread, seed=seed, flags=oflags)
    desc = "%s %dx%d seed %d rec %d spread %d tpl %d sem %d flags %#x %s %s" % (name, w, h, seed, rec, spread, tpl, sem, gflags, stripes or "", extra or "")
    steps = []
    for _ in range(int(rng.integers(1, 5)) + (1 if WILD else 0)):
        kind = rng.choice(["render", "frame", "move", "clear"], p=[0.4, 0.3, 0.2, 0.1])
        if WILD and not steps:
            kind = "move"                                   # FUZZ_WILD=1: every case starts with a wild camera move
        if kind == "move":
            dx, dy, dz = (float(x) for x in rng.normal(0, 0.4, 3)); ax, ay = float(rng.normal(0, 0.2)), float(rng.normal(0, 0.2))
            if WILD or rng.random() < 0.3:                  # a wild move: into, through or behind the geometry, looking anywhere (triangles across the eye plane:
                ext = float(np.ptp(sc["tri_verts"].reshape(-1, 3), axis=0).max())       # the screen-space structures must stand down, the tree walk answers)
                dx, dy, dz = (float(x) for x in rng.normal(0, 0.5 * ext, 3)); ax, ay = float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5))
            rt.camera.move_rel(dx, dy, dz); orc.camera_move_rel(dx, dy, dz)
            rt.camera.add_x_angle(ax); orc.camera_add_x_angle(ax); rt.camera.add_y_angle(ay); orc.camera_add_y_angle(ay)
            rt.film.clear(); orc.film_clear()               # what the reference's binary does after a camera move (main.rs:116-169)
        elif kind == "clear":
            rt.film.clear(); orc.film_clear()
        elif kind == "render" and not stripes:
            spp = int(rng.choice([5, 8, 12, 16, 17, 24, 33])) if big_spp else int(rng.integers(1, 4))
            c = rt.render(spp); oc = orc.render(spp, nthreads=8)
            assert (c.primary, c.bounce, c.shadow, c.primary_hits) == (oc["primary"], oc["bounce"], oc["shadow"], oc["primary_hits"]), (desc, steps, c.as_dict(), oc)
        elif kind == "render":
            rt.render(int(rng.choice([5, 8, 12, 17])) if big_spp else int(rng.integers(1, 3)))              # striped handle: compared on its owned rows below (the oracle renders every row)
            steps.append("render(striped)")
            continue
        else:
            if stripes:
                continue
            assert rt.trace_frame_additive() == orc.trace_frame_additive()
        steps.append(kind)
    gs, gq, gn = rt.film.pixel_datas()
    if stripes:
        # the oracle has no stripes: render the same number of samples into every row, compare the owned rows only
        owned = rt.owned_rows()
        n = gn.reshape(h, w)
        spp = int(n[owned[0], 0]) if owned.size else 0
        assert np.all(n[owned] == spp), desc
        rest = np.setdiff1d(np.arange(h), owned)
        assert not n[rest].any(), desc
        orc.film_clear()
        if spp:
            orc.render(spp, nthreads=8)
        os_, oq, on = orc.film()
        assert np.array_equal(bits(gs).reshape(h, w, 3)[owned], bits(os_).reshape(h, w, 3)[owned]), (desc, steps)
        assert np.array_equal(bits(gq).reshape(h, w, 3)[owned], bits(oq).reshape(h, w, 3)[owned]), (desc, steps)
    else:
        os_, oq, on = orc.film()
        assert np.array_equal(gn, on), (desc, steps)
        assert np.array_equal(bits(gs), bits(os_)), (desc, steps)
        assert np.array_equal(bits(gq), bits(oq)), (desc, steps)
        assert np.array_equal(rt.get_tonemapped_pixels(), orc.get_tonemapped_pixels()), (desc, steps)
    if verbose:
        print("ok  ", desc, steps, flush=True)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = ge.load_package(); O = ge.load_oracle()
    import importlib
    sio = importlib.import_module("raytracer_rs_amd.scene_io")
    cache = {}

    def scenes(name):
        if name not in cache:
            cache[name] = sio.load_scene_file(os.path.join(ge.SCENES, name + ".scene"))
        return cache[name]
    rng = np.random.default_rng(seed)
    for _ in range(cases):
        one_case(pkg, O, scenes, rng)
    print("%d cases, all bit-exact" % cases)


if __name__ == "__main__":
    main()
