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
    # torch fills the buffer with zeros on ITS stream, asynchronously; the library writes it on its own: ordered through torch's stream, as bench.py
    # does (without this the zero fill could land after the pixels — seen once in a loaded suite run)
    rt.tonemap_owned_rows_device(stripe.data_ptr(), rows.size * W, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
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


def _bench(argv, env_extra=None, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_self_launch_with_one_rank():
    """`bench.py --gpus 1 --spawn`: the parent starts torch.distributed.run, which starts the one rank; the line carries the
    three result groups (weak = the line itself, `strong`, `c5`), each with its gather fields, and the drop-in object."""
    d = _bench(["--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "Mrays/s"
    assert d["gather_path"] == "none" and d["rccl_ranks"] == 0 and d["gather_error"] is None
    assert d["hbm_allocated_bytes"] > 0 and d["ms_per_step"] > 0
    for g in ("strong", "c5"):
        assert d[g]["scaling"] == "strong" and d[g]["ms_per_step"] > 0 and d[g]["gather_path"] == "none" and d[g]["gather_error"] is None
    assert "3840x2160 x 256 spp" in d["c5"]["workload"]
    assert d["c5"]["ms_per_step"] > 8 * d["strong"]["ms_per_step"]           # 16x the samples of the 1080p x 64 frame
    assert d["dropin"]["ms_per_step"] > 0 and d["dropin"]["primary_rays_per_s"] > 1e6
    r = d["roofline"]
    assert r["secondary_launches"] > 0 and r["avg_secondary_launch_ms"] > 0 and r["clock_mhz_observed"] and 500 < r["clock_mhz_observed"] < 3000


def test_bench_two_ranks_share_the_gpu_and_report_the_fallback():
    """N = 2 rehearsal on the one GPU of the box (gloo, MI355RT_BENCH_SHARE_GPU): the library's RCCL path is asked for, RCCL
    refuses two ranks on one device, every rank agrees on that, and the line SAYS so: gather_path torch_all_gather with a
    non-null gather_error — in every result group."""
    d = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--spp", "2"], {"MI355RT_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["gather_path"] in ("torch_all_gather", "library_rccl")
    if d["gather_path"] == "torch_all_gather":
        assert d["gather_error"] and d["rccl_ranks"] == 0
    else:
        assert d["gather_error"] is None and d["rccl_ranks"] == 2
    assert d["gather_ms_rank0"] is not None
