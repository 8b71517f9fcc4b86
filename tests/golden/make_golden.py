#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

  scenes/*.scene   flat scene containers made by the product's COLLADA ingest
                   (raytracer-rs_amd/bin/dae2scene) from the reference's bundled data files
                   /root/reference/data/*.dae (+ blender_cycles_ico3.png).  Data only: triangle
                   soup, materials, light, camera matrix, texture bytes.
  render_*.npz     outputs of the CPU oracle (oracle/oracle.c — our C restatement; NOT of the Rust
                   reference, which cannot be built here): 64x64, 4 spp, seed 1, for the octree
                   intersector (reference default, 70 tris/leaf) and the brute-force intersector;
                   film sums, sample counts, packed 0xAARRGGBB pixels, ray counters, and a batch of
                   4096 primary rays with their brute-force and octree hit records.

Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF_DATA = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = ["4boxes", "ico2", "thai2", "ico3_tex"]


def main():
    ge.build()
    tool = os.path.join(ge.PKG_DIR, "bin", "dae2scene")
    if os.path.isdir(REF_DATA):
        for s in SCENES:
            subprocess.check_call([tool, os.path.join(REF_DATA, s + ".dae"), os.path.join(HERE, "scenes", s + ".scene")])
    else:
        print("no %s: keeping the committed scene files" % REF_DATA)
    pkg = ge.load_package()
    import importlib
    sio = importlib.import_module("raytracer_rs_amd.scene_io")
    O = ge.load_oracle()
    w = h = 64
    for s in SCENES:
        scene = sio.load_scene_file(os.path.join(HERE, "scenes", s + ".scene"))
        out = {}
        for mode, flags in (("octree", 0), ("brute", O.FLAG_BRUTE_FORCE)):
            orc = O.Oracle(scene, w, h, seed=1, flags=flags)
            c = orc.render(4, nthreads=8)
            fs, fq, fn = orc.film()
            out[mode + "_sum"] = fs; out[mode + "_sumsq"] = fq; out[mode + "_n"] = fn
            out[mode + "_ldr"] = orc.get_tonemapped_pixels()
            out[mode + "_counts"] = np.array([c["primary"], c["bounce"], c["shadow"], c["primary_hits"]], np.uint64)
            if mode == "octree":
                st = orc.octree_stats()
                out["octree_stats"] = np.array([st["nodes"], st["inner"], st["leaves"], st["empty"], st["depth"], st["tri_refs"]], np.uint32)
                rng = np.random.default_rng(123)
                pix = rng.integers(0, w * h, 4096)
                rays = np.stack([orc.primary_ray(int(p), int(i % 7)) for i, p in enumerate(pix)])
                out["rays"] = rays
                t1, p1 = orc.intersect(rays, brute=False)
                t2, p2 = orc.intersect(rays, brute=True)
                out["octree_tuv"] = t1; out["octree_prim"] = p1; out["brute_tuv"] = t2; out["brute_prim"] = p2
        np.savez_compressed(os.path.join(HERE, "render_%s.npz" % s), **out)
        d = (out["octree_ldr"] != out["brute_ldr"]).mean()
        print("%-9s octree-vs-brute differing pixels %.4f  hit records differing %.4f" % (s, d, (out["octree_prim"] != out["brute_prim"]).mean()))


if __name__ == "__main__":
    main()
