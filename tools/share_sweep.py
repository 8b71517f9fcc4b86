"""Rank 0's share of the strong-scaled headline frame (thai2 1920x1080x64 over N ranks, 2-row stripes) and the N = 1 frame on ONE GPU,
under environment-variable variants of the library (read at create time).  usage: share_sweep.py VAR=a,b,c [VAR2=...] [--worlds 8,1]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
worlds = [8, 1]
sweeps = []
for a in sys.argv[1:]:
    if a.startswith("--worlds"):
        worlds = [int(x) for x in a.split("=", 1)[1].split(",")]
    else:
        k, v = a.split("=", 1)
        sweeps.append([(k, x) for x in v.split(",")])
for combo in itertools.product(*sweeps) if sweeps else [()]:
    for k, v in combo:
        if v == "-":
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    out = []
    for world in worlds:
        rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=int(os.environ.get("SWEEP_STRIPE_ROWS", "2")), stripe_rank=0, stripe_world=world)
        ts = []
        for it in range(7):
            rt.film.clear()
            t = time.time(); c = rt.render(64); ts.append(time.time() - t)
        ts.sort()
        out.append("N=%d share: min %.3f med %.3f ms (gpu %.3f; %.1f GB HBM)" % (world, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, c.total_ms, rt.hbm_allocated_bytes() / 1e9))
        if os.environ.get("SWEEP_COUNT"):          # nodes visited / triangles tested per ray (instrumented frame, 4 spp)
            rt.set_flags(pkg.FLAG_COUNT_STEPS); rt.film.clear(); cc = rt.render(4); rt.set_flags(0)
            rays = cc.primary - cc.primary_culled + cc.bounce + cc.shadow - cc.shadow_skipped
            st = rt.accel_stats()
            out[-1] += " [%.2f nodes %.2f tris per traced ray; %d nodes depth %d build %.1f ms]" % (cc.nodes_visited / rays, cc.tris_tested / rays, st["nodes"], st["max_depth"], st["bvh_build_ms"])
        del rt
    print(" ".join("%s=%s" % kv for kv in combo) or "default", "|", " | ".join(out), flush=True)
