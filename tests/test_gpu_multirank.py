"""Two ranks (one process each, gloo rendezvous on 127.0.0.1) render their row stripes on the GPU
and gather the packed stripes exactly as bench.py does for N > 1; the gathered frame must be the
single-handle frame.  (RCCL itself needs one GPU per rank; the GPU box of the test tier has one, so
the collective here is gloo on host copies of the device stripes.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, STRIPE = 160, 90, 3, 8


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    sio = importlib.import_module("raytracer_rs_amd.scene_io")
    st = importlib.import_module("raytracer_rs_amd.stripes")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    scene = sio.load_scene_file(os.path.join(ge.SCENES, "ico2.scene"))
    rt = pkg.create_raytracer_from_arrays(scene, 70, W, H, seed=1, stripe_rows=STRIPE, stripe_rank=rank, stripe_world=world)
    rows = rt.owned_rows()
    assert list(rows) == st.owned_rows(H, STRIPE, rank, world)
    rt.render(SPP)
    fg_dev = st.FrameGather(H, W, STRIPE, world, "cuda")
    stripe = fg_dev.stripe_buffer("cuda")
    rt.tonemap_owned_rows_device(stripe.data_ptr(), rows.size * W)          # the device-pointer entry bench.py uses
    fg = st.FrameGather(H, W, STRIPE, world, "cpu")
    frame = fg.gather(dist, stripe.cpu()).numpy().view(np.uint32).copy()
    ok = True
    if rank == 0:
        full = pkg.create_raytracer_from_arrays(scene, 70, W, H, seed=1)
        full.render(SPP)
        ok = bool(np.array_equal(frame, full.get_tonemapped_pixels()))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_the_single_gpu_frame():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
    assert res == [(0, True), (1, True)]
