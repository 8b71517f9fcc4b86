"""One rank's share of an N-GPU weak-scaling frame on one GPU, a few frames (run it under rocprofv3 --kernel-trace --stats).
usage: share_stats.py <world> [slices]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
world = int(sys.argv[1]); slices = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=8, stripe_rank=0, stripe_world=world)
if slices: rt.set_slices(slices)
best = 1e9
for it in range(4):
    rt.film.clear(); t = time.time(); c = rt.render(64 * world); best = min(best, time.time() - t)
print("world %d slices %d: %.2f ms, %.1f M rays" % (world, rt.get_slices(), best * 1e3, c.as_dict()["total_rays"] / 1e6))
