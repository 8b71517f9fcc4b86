import os, sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
for world in (8, 4, 2):
    for sl in (1, 2, 3, 4):
        rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=8, stripe_rank=0, stripe_world=world)
        rt.set_slices(sl)
        best = 1e9
        for it in range(5):
            rt.film.clear(); t = time.time(); c = rt.render(64); best = min(best, time.time() - t)
        print("strong world %d slices %d: %.2f ms" % (world, sl, best * 1e3), flush=True)
        del rt
