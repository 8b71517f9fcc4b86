"""Load balance of the row-stripe deal over N ranks: rays and time of every rank's share (one GPU renders them one after the
other, 16 spp), max over mean, for several stripe heights.  usage: stripe_balance.py [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64            # 64: strong-scaling shares; 64 * world: weak-scaling shares
stripes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 4, 8, 16, 32]
for stripe in stripes:
    rays, ms = [], []
    for r in range(world):
        rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=stripe, stripe_rank=r, stripe_world=world)
        best = 1e9
        for it in range(3):
            rt.film.clear(); t = time.time(); c = rt.render(spp); best = min(best, time.time() - t)
        rays.append(c.as_dict()["total_rays"]); ms.append(best * 1e3)
        del rt
    print("stripe %2d rows, %d ranks, %d spp: rays max/mean %.3f, time max/mean %.3f (max %.2f ms, mean %.2f ms)" % (
        stripe, world, spp, max(rays) / (sum(rays) / world), max(ms) / (sum(ms) / world), max(ms), sum(ms) / world), flush=True)
