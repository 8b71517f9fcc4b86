"""Host-side logic that needs no GPU: stats strings, scene container, stripe decomposition and the
frame gather over two gloo ranks (the N > 1 path of bench.py)."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stats_strings(pkg):
    st = pkg.stats.Stats()
    s = st.stats(96000)                                                    # stats.rs:21-31
    assert re.fullmatch(r"fps: [0-9.e+]+  primary rays/s: \d+", s)
    st.stats(96000)
    assert re.fullmatch(r"mean fps: [0-9.e+]+  mean primary rays/s: [0-9.e+]+", st.mean_stats())
    assert st.num_measurements == 2


def test_scene_container_roundtrip(scene_io, scenes):
    for name, ntri, ngeom in (("4boxes", 48, 4), ("ico2", 608, 5), ("thai2", 20049, 2), ("ico3_tex", 608, 5)):
        sc = scenes(name)
        assert sc["tri_verts"].shape == (ntri, 9) and sc["tri_geom"].shape == (ntri,)
        assert len(sc["mat_kind"]) == ngeom and sc["tri_geom"].max() == ngeom - 1
        assert np.all(np.diff(sc["tri_geom"].astype(np.int64)) >= 0)       # geometry order = visual-scene node order
        assert sc["lights"].shape == (1, 6) and len(sc["cameras"]) == 1
    tex = scenes("ico3_tex")["textures"]
    assert len(tex) == 1 and tex[0].shape == (640, 640, 3) and tex[0].max() < 1.0
    assert np.array_equal(tex[0] * 256.0, np.round(tex[0] * 256.0))        # texels are byte/256 exactly


def test_stripe_partition_properties(pkg):
    import importlib
    st = importlib.import_module("raytracer_rs_amd.stripes")
    for h, sr, world in ((1080, 8, 1), (1080, 8, 2), (1080, 8, 8), (2160, 16, 8), (50, 4, 3), (7, 8, 4)):
        rows = [st.owned_rows(h, sr, r, world) for r in range(world)]
        assert sorted(sum(rows, [])) == list(range(h))                      # a partition of the frame
        assert max(len(r) for r in rows) <= st.max_owned_rows(h, sr, world)
        assert max(len(r) for r in rows) - min(len(r) for r in rows) <= sr   # balanced to one stripe


def _gather_worker(rank, world, port, h, w, sr, q):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    st = importlib.import_module("raytracer_rs_amd.stripes")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    fg = st.FrameGather(h, w, sr, world, "cpu")
    stripe = fg.stripe_buffer("cpu")
    rows = st.owned_rows(h, sr, rank, world)
    vals = (torch.tensor(rows, dtype=torch.int32)[:, None] * 100000 + torch.arange(w, dtype=torch.int32)[None, :])
    stripe.view(fg.max_rows, w)[: len(rows)] = vals
    frame = fg.gather(dist, stripe).view(h, w)
    expect = torch.arange(h, dtype=torch.int32)[:, None] * 100000 + torch.arange(w, dtype=torch.int32)[None, :]
    q.put((rank, bool(torch.equal(frame, expect))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,w,sr", [(2, 50, 16, 8), (3, 37, 5, 4)])
def test_frame_gather_over_gloo_ranks(world, h, w, sr):
    """one process per rank, gloo on 127.0.0.1: every rank ends up with the full frame, rows in place."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, h, w, sr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)]


def test_gather_slot_arithmetic_matches_the_placement_kernel(pkg):
    """place_stripes_kernel (csrc/kernels.hip) and Renderer::slot_rows / rows_of_rank (csrc/renderer.cpp), restated:
    rank r's packed rows land in slot r; pixel (y, x) of the frame comes from slot (y / stripe) % world at local row
    (y / stripe / world) * stripe + y % stripe.  Checked against stripes.owned_rows for 1..8 ranks and ragged heights."""
    import importlib
    stripes = importlib.import_module("raytracer_rs_amd.stripes")
    for height in (1, 7, 8, 50, 131, 1080, 2160):
        for stripe_rows in (1, 4, 8):
            for world in (1, 2, 3, 8):
                nstripes = (height + stripe_rows - 1) // stripe_rows
                slot_rows = ((nstripes + world - 1) // world) * stripe_rows
                assert slot_rows == stripes.max_owned_rows(height, stripe_rows, world)
                seen = set()
                for r in range(world):
                    rows = stripes.owned_rows(height, stripe_rows, r, world)
                    n = sum(min(stripe_rows, height - s * stripe_rows) for s in range(r, nstripes, world))
                    assert n == len(rows) <= slot_rows
                    for local, y in enumerate(rows):
                        s_ = y // stripe_rows
                        assert s_ % world == r and (s_ // world) * stripe_rows + (y - s_ * stripe_rows) == local
                        seen.add(y)
                assert seen == set(range(height))


def test_bench_launches_its_own_ranks_before_touching_torch():
    """`python bench.py --gpus N` without a launcher starts `torch.distributed.run` itself, as a child process, before the
    parent has imported torch or the library (MI355RT_BENCH_DRY_SPAWN prints what would be started)."""
    import json
    import subprocess
    env = dict(os.environ, MI355RT_BENCH_DRY_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    for argv, n in ((["--gpus", "8", "--steps", "5", "--warmup", "2"], 8), (["--gpus", "1", "--spawn"], 1)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        cmd = d["spawn"]
        assert d["torch_imported"] is False and d["library_imported"] is False
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
        assert cmd[cmd.index("--nproc-per-node") + 1] == str(n) and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
        tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
        assert "--spawn" not in tail and tail[:2] == ["--gpus", str(n)]
    # under a launcher (WORLD_SIZE set) it is a rank, not a parent: a mismatch is refused instead of spawning again
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-pmc"], env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


class _FakeRt:
    """stands in for a RayTracer in the set-up protocol of stripes.NativeGather (no GPU, no RCCL)"""

    def __init__(self, available, init_ok, ranks):
        self.available, self.init_ok, self.ranks = available, init_ok, ranks
        self.calls = []

    def comm_available(self):
        self.calls.append("available")
        return self.available, "" if self.available else "cannot load librccl.so"

    def comm_init(self, _id):
        self.calls.append("init")
        if not self.init_ok:
            raise RuntimeError("ncclCommInitRank: unhandled system error")

    def comm_ranks(self):
        return self.ranks

    def comm_destroy(self):
        self.calls.append("destroy")


class _FakePkg:
    @staticmethod
    def comm_unique_id():
        return bytes(range(128))


def _native_worker(rank, world, port, scenario, q):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    st = importlib.import_module("raytracer_rs_amd.stripes")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rt = {"rank1_cannot_load": _FakeRt(rank != 1, True, world), "rank0_init_fails": _FakeRt(True, rank != 0, world),
          "wrong_count": _FakeRt(True, True, 1 if rank == 1 else world), "all_good": _FakeRt(True, True, world)}[scenario]
    err = None
    try:
        ng = st.NativeGather(_FakePkg, rt, dist, rank, world, device="cpu")
        assert ng.ranks == world
    except RuntimeError as e:
        err = str(e)
    q.put((rank, err, rt.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["rank1_cannot_load", "rank0_init_fails", "wrong_count", "all_good"])
def test_native_gather_setup_is_agreed_step_by_step(scenario):
    """No rank enters the collective comm_init unless EVERY rank passed the local pre-check (a rank that cannot load
    librccl.so must not leave the others waiting inside RCCL's bootstrap); a failing init or a communicator of the wrong
    size is raised on every rank together, and ranks that did get a communicator give it back."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    errs = [e for _, e, _ in res]
    calls = [c for _, _, c in res]
    if scenario == "all_good":
        assert errs == [None, None] and calls == [["available", "init"]] * 2
    elif scenario == "rank1_cannot_load":
        assert all(e and "unavailable" in e for e in errs)
        assert all("init" not in c for c in calls)                          # nobody entered the collective
    elif scenario == "rank0_init_fails":
        assert all(e and "comm_init failed" in e for e in errs)
        assert calls[0] == ["available", "init"] and calls[1] == ["available", "init", "destroy"]
    else:
        assert all(e and "expected 2" in e for e in errs)
        assert all(c == ["available", "init", "destroy"] for c in calls)
