"""Scratch: how much does overlapping two half-frame renders (two handles, two streams, two host threads)
gain over one full-frame render?  Estimates the value of filling the end-of-launch drain windows."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
name = sys.argv[1] if len(sys.argv) > 1 else "thai2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc = sio.load_scene_file(os.path.join(ge.SCENES, name + ".scene"))
full = pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1)
for world in (2, 3, 4):
    parts = [pkg.create_raytracer_from_arrays(sc, 70, 1920, 1080, seed=1, stripe_rows=8, stripe_rank=r, stripe_world=world) for r in range(world)]
    for it in range(3):
        t = time.time(); full.render(spp); t_full = time.time() - t
        t = time.time()
        for p in parts: p.render(spp)
        t_seq = time.time() - t
        th = [threading.Thread(target=p.render, args=(spp,)) for p in parts]
        t = time.time()
        for x in th: x.start()
        for x in th: x.join()
        t_par = time.time() - t
        print("%s spp=%d world=%d full %.2f ms | parts sequential %.2f ms | parts concurrent %.2f ms" % (name, spp, world, t_full * 1e3, t_seq * 1e3, t_par * 1e3), flush=True)
    del parts
