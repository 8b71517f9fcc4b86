"""One rank's share of an N-GPU frame, rendered on ONE GPU (no gather): what each rank of `bench.py --gpus N` does per step.
weak:   thai2 1920x1080, stripes of rank 0 of N (2-row stripes like bench.py; PROBE_STRIPE_ROWS), 64*N spp (the same number of samples as the N=1 frame)
strong: thai2 1920x1080x64 spp and 3840x2160x256 spp (BASELINE config 5) dealt to N ranks: rank 0's rows at the full spp
The slowest rank sets the frame time; rank 0 always owns the first stripe, i.e. the largest share when the stripes do not divide evenly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "thai2.scene"))


def probe(label, w, h, spp_of, worlds):
    base = None
    for world in worlds:
        rt = pkg.create_raytracer_from_arrays(sc, 70, w, h, seed=1, stripe_rows=int(os.environ.get("PROBE_STRIPE_ROWS", "2")), stripe_rank=0, stripe_world=world)
        spp = spp_of(world)
        best = 1e9
        for it in range(4):
            rt.film.clear()
            t = time.time(); c = rt.render(spp); best = min(best, time.time() - t)
        rays = c.as_dict()["total_rays"]
        base = base or best
        print("%s world %d: rank 0 renders %d rows x %d spp: %.2f ms, %.1f M rays, %.0f Mrays/s%s" % (
            label, world, rt.owned_rows().size, spp, best * 1e3, rays / 1e6, rays / best / 1e6,
            "" if label == "weak" else "  -> speed-up x%.2f of %d" % (base / best, world)), flush=True)
        del rt


probe("weak", 1920, 1080, lambda n: 64 * n, (1, 2, 4, 8))
probe("strong 1080p x 64", 1920, 1080, lambda n: 64, (1, 2, 4, 8))
probe("strong c5 4K x 256", 3840, 2160, lambda n: 256, (2, 4, 8))
