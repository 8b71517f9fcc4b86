"""Lane census of the trace launches of one 1-slice pass (MI355RT_FLAG_COUNT_STEPS + MI355RT_DEBUG_UTIL): where the lanes are at
every loop iteration, executions of the two sections, refill statistics.  The library prints to stderr.  usage: lane_census.py [scene] [spp]"""
import os, sys
os.environ["MI355RT_DEBUG_UTIL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
scene = sys.argv[1] if len(sys.argv) > 1 else "thai2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = sio.load_scene_file(os.path.join(ge.SCENES, scene + ".scene"))
flags = int(os.environ.get("CENSUS_FLAGS", "0"))
rt = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, flags=flags | pkg.FLAG_COUNT_STEPS)
rt.set_slices(1)
c = rt.render(spp)
d = c.as_dict()
print({k: d[k] for k in ("primary", "bounce", "shadow", "primary_hits", "nodes_visited", "tris_tested", "inner_execs", "leaf_execs")})
