import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
L = pkg.lib()
def scene(name): return sio.load_scene_file(os.path.join(ge.SCENES, name + ".scene"))
def rnd(ntri, seed):
    rng = np.random.default_rng(seed)
    sc = dict(scene("4boxes"))
    c = rng.uniform([-1, -1, 2], [5, 1, 5], (ntri, 3)).astype(np.float32)
    v = (c[:, None, :] + rng.uniform(-0.02, 0.02, (ntri, 3, 3))).astype(np.float32)
    sc["tri_verts"] = v.reshape(ntri, 9); sc["tri_geom"] = np.zeros(ntri, np.uint32)
    return sc
for name, sc in (("thai2", scene("thai2")), ("ico2", scene("ico2")), ("random120k", rnd(120000, 7))):
    for fl in (pkg.FLAG_TRUE_CLOSEST_HIT | pkg.FLAG_DEVICE_LBVH, pkg.FLAG_TRUE_CLOSEST_HIT):
        rt = pkg.create_raytracer_from_arrays(sc, 70, 640, 480, seed=1, flags=fl | pkg.FLAG_COUNT_STEPS)
        msg = (L.mi355rt_last_error(rt._h) or b"").decode()
        c = rt.render(4).as_dict()
        rays = c["primary"] + c["bounce"] + c["shadow"]
        print(name, "flags", fl, rt.bvh_build_info(), rt.accel_stats(), "nodes/ray %.2f tris/ray %.2f" % (c["nodes_visited"] / rays, c["tris_tested"] / rays), "|", msg, flush=True)
