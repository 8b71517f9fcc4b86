"""Can two 50-row frames run side by side on the chip?  Two handles on one GPU, their fused kernels in flight together (separate streams and pass buffers)."""
import os, sys, time
os.environ["MI355RT_NO_SPECULATE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
for fchunk in ("32", "64"):
    os.environ["MI355RT_FUSED_CHUNK"] = fchunk
    a = pkg.create_raytracer_from_arrays(sc, 70, 1024, 768, seed=1)
    b = pkg.create_raytracer_from_arrays(sc, 70, 1024, 768, seed=1)
    c = pkg.create_raytracer_from_arrays(sc, 70, 1024, 768, seed=1)
    for n, hs in ((1, [a]), (2, [a, b]), (3, [a, b, c])):
        for h in hs: h.trace_frame_additive(); h.synchronize()
        t0 = time.perf_counter()
        for it in range(300):
            for h in hs: h.trace_frame_additive()
            for h in hs: h.synchronize()
        dt = (time.perf_counter() - t0) / 300
        print("chunk %s: %d frames in flight together: %.3f ms per round = %.3f ms per frame" % (fchunk, n, dt * 1e3, dt * 1e3 / n), flush=True)
    del a, b, c
