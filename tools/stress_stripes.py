"""Stress: striped handles against the full frame, many times, optionally several processes at once on one GPU (a flaky-test hunt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
sio = importlib.import_module("raytracer_rs_amd.scene_io")
sc = sio.load_scene_file(os.path.join(ge.SCENES, "ico2.scene"))
W, H, SPP, STRIPE = 160, 90, 3, 8
tag = sys.argv[1] if len(sys.argv) > 1 else "p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
full = pkg.create_raytracer_from_arrays(sc, 70, W, H, seed=1); full.render(SPP)
ref = full.get_tonemapped_pixels().reshape(H, W).copy()
fs, fq, fn = full.film.pixel_datas()
nbad = 0
for it in range(iters):
    for rank in range(2):
        rt = pkg.create_raytracer_from_arrays(sc, 70, W, H, seed=1, stripe_rows=STRIPE, stripe_rank=rank, stripe_world=2)
        rows = rt.owned_rows()
        rt.render(SPP)
        buf = torch.empty(rows.size * W, dtype=torch.int32, device="cuda")
        rt.tonemap_owned_rows_device(buf.data_ptr(), rows.size * W)
        got = buf.cpu().numpy().view(np.uint32).reshape(rows.size, W)
        s, q, n = rt.film.pixel_datas()
        dpix = (got != ref[rows]).sum()
        dfilm = (s.view(np.uint32).reshape(H, W, 3)[rows] != fs.view(np.uint32).reshape(H, W, 3)[rows]).any(axis=2)
        if dpix or dfilm.any():
            nbad += 1
            rr, cc = np.nonzero(dfilm)
            print(tag, "iter", it, "rank", rank, "pixel diffs", int(dpix), "film diffs", int(dfilm.sum()), "rows", sorted(set(int(rows[r]) for r in rr))[:10], "cols", sorted(set(int(c) for c in cc))[:12], flush=True)
        del rt
print(tag, "done, bad:", nbad, flush=True)
