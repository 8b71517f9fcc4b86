"""Load balance of the trace launches of rank 0's share of the strong-scaled headline frame (COUNT build + MI355RT_DEBUG_UTIL):
mean wave busy time against the first / last wave out of work, per round.  usage: share_census.py [world] [spp]"""
import os, sys
os.environ["MI355RT_DEBUG_UTIL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))
rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=2, stripe_rank=0, stripe_world=world, flags=pkg.FLAG_COUNT_STEPS)
rt.set_slices(1)
rt.render(spp)
rt.film.clear()
c = rt.render(spp)
print(c.as_dict())
