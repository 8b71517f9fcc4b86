"""Scratch timing of the render path on the GPU box (not a test, not the bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["thai2"]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name in names:
    sc = sio.load_scene_file(os.path.join(ge.SCENES, name + ".scene"))
    rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, flags=flags)
    print(name, rt.accel_stats(), flush=True)
    for it in range(int(os.environ.get("ITERS", "3"))):
        t = time.time(); c = rt.render(spp); dt = time.time() - t
        d = c.as_dict()
        print("%s spp=%d wall %.1f ms  gpu %.1f ms trace %.1f ms  rays %.1fM (p %.1fM b %.1fM s %.1fM) -> %.1f Mrays/s total, %.1f Mprimary/s  nodes/ray %.1f tris/ray %.1f" % (
            name, spp, dt * 1e3, d["total_ms"], d["trace_ms"], d["total_rays"] / 1e6, d["primary"] / 1e6, d["bounce"] / 1e6, d["shadow"] / 1e6,
            d["total_rays"] / d["total_ms"] / 1e3, d["primary"] / d["total_ms"] / 1e3,
            d["nodes_visited"] / max(d["total_rays"], 1), d["tris_tested"] / max(d["total_rays"], 1)), flush=True)
