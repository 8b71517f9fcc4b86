"""Replays tools/parity_fuzz.py cases of a seed until one fails, then replays THAT case under the library's switches to see which structure is at fault.
usage: [FUZZ_*=1] fuzz_repro.py <seed> [max cases]"""
import copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
import parity_fuzz as fz
import importlib
pkg = ge.load_package(); O = ge.load_oracle()
sio = importlib.import_module("raytracer_rs_amd.scene_io")
cache = {}
def scenes(name):
    if name not in cache:
        cache[name] = sio.load_scene_file(os.path.join(ge.SCENES, name + ".scene"))
    return cache[name]
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
for i in range(n):
    state = copy.deepcopy(rng.bit_generator.state)
    try:
        fz.one_case(pkg, O, scenes, rng, verbose=False)
    except AssertionError as e:
        print("case %d fails: %s" % (i, str(e)[:300]), flush=True)
        for sw in ("MI355RT_NO_RASTER", "MI355RT_NO_CULL_MASK", "MI355RT_NO_LIGHT_MAP", "MI355RT_NO_SPECULATE", "MI355RT_NO_FUSED", "MI355RT_NO_LIVE", "MI355RT_NO_CULL_CACHE", "MI355RT_NO_CULL", "MI355RT_NO_FUSE_PRIMARY"):
            os.environ[sw] = "1"
            r = np.random.default_rng(0); r.bit_generator.state = copy.deepcopy(state)
            try:
                fz.one_case(pkg, O, scenes, r, verbose=False); print("  with %s: passes" % sw, flush=True)
            except AssertionError as e2:
                print("  with %s: still fails" % sw, flush=True)
            del os.environ[sw]
        # once more, keeping the two renderers: which pixels differ, and what the culling stage saw
        os.environ["MI355RT_DEBUG_CULL"] = "1"
        kept = {}
        mk, Orc = pkg.create_raytracer_from_arrays, O.Oracle
        def mk2(*a, **k):
            kept["rt"] = mk(*a, **k); kept["args"] = (a[1:], k); return kept["rt"]
        class Orc2(Orc):
            def __init__(self, *a, **k):
                super().__init__(*a, **k); kept["orc"] = self
        pkg.create_raytracer_from_arrays = mk2; O.Oracle = Orc2
        r = np.random.default_rng(0); r.bit_generator.state = copy.deepcopy(state)
        try:
            fz.one_case(pkg, O, scenes, r, verbose=False)
        except AssertionError:
            pass
        rt, orc = kept["rt"], kept["orc"]
        print("create args:", kept["args"])
        gs, gq, gn = rt.film.pixel_datas(); os_, oq, on = orc.film()
        w = kept["args"][0][1]; h = kept["args"][0][2]
        bad = np.nonzero((gs.view(np.uint32).reshape(-1, 3) != os_.view(np.uint32).reshape(-1, 3)).any(1))[0]
        print("samples equal:", np.array_equal(gn, on), " pixels whose sums differ: %d of %d" % (len(bad), w * h))
        print("  columns:", sorted(set(int(b % w) for b in bad)), " rows:", sorted(set(int(b // w) for b in bad)))
        print("  film n (gpu) row 0:", gn.reshape(h, w)[0].tolist(), " col 20:", gn.reshape(h, w)[:, min(20, w - 1)].tolist())
        lit = (os_.reshape(h, w, 3) != 0).any(2); glit = (gs.reshape(h, w, 3) != 0).any(2)
        for y in range(h):
            print("   ", "".join("#" if lit[y, x] and glit[y, x] else "o" if lit[y, x] else "g" if glit[y, x] else "." for x in range(w)))
        sys.exit(1)
print("no failure in %d cases" % n)
