"""What do the chunks that hold no ray cost?  thai2 1920x1080x64 with the camera turned away from the scene: every primary chunk is culled, every
later launch finds only empty chunks — the frame time is the per-chunk overhead of all launches (plus the resolve pass over the film)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=2)
for turn in (0.0, 1.2):
    rt.camera.add_y_angle(turn)
    ts = []
    for it in range(5):
        rt.film.clear(); c = rt.render(64); ts.append(c.total_ms)
    print("camera turned by %.2f rad: frame %.3f ms (gpu), primary hits %d, culled %.3f" % (turn, min(ts), c.primary_hits, c.primary_culled / c.primary))
rt.set_flags(pkg.FLAG_TIME_KERNELS)
rt.film.clear(); c = rt.render(64)
print("trace launches of the empty frame: %.3f ms in %d launches" % (c.trace_ms, c.trace_launches))
