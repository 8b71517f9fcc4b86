"""Scratch: one rank's share of an N-GPU weak-scaling frame on ONE GPU: stripes of rank 0 of N, 64*N spp
(the same number of samples as the N=1 frame).  Shows what the per-rank frame costs as N grows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
for world in (1, 2, 4, 8):
    rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=8, stripe_rank=0, stripe_world=world)
    for it in range(3):
        rt.film.clear()
        t = time.time(); c = rt.render(64 * world); dt = time.time() - t
    d = c.as_dict()
    print("world %d: rank 0 renders %d rows x %d spp: %.2f ms, %.1f M rays" % (world, rt.owned_rows().size, 64 * world, dt * 1e3, d["total_rays"] / 1e6), flush=True)
    del rt
