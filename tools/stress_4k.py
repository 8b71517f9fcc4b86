"""Scratch: one 3840x2160 x 32 spp frame on one GPU (265 M samples: two passes per slice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib, numpy as np
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
rt = pkg.create_raytracer_from_arrays(sc, 70, 3840, 2160, seed=1)
for it in range(2):
    rt.film.clear(); t = time.time(); c = rt.render(32); dt = time.time() - t
    print("4K x 32 spp: %.1f ms, %.1f Mrays/s, launches %d" % (dt * 1e3, c.as_dict()["total_rays"] / dt / 1e6, c.trace_launches), flush=True)
s, q, n = rt.film.pixel_datas()
print("n ok", bool(np.all(n == 32)), "finite", bool(np.isfinite(s).all()))
