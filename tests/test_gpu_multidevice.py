"""GPU tests of the multi-GPU paths INSIDE the library: the device group (one handle, N devices of one process,
stripes gathered with hipMemcpyPeerAsync) — run here with all members on the one GPU of the box
(MI355RT_FLAG_GROUP_SHARES_DEVICE) — and the RCCL communicator path with a world of one."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def make(pkg, scenes, name, w, h, **kw):
    return pkg.create_raytracer_from_arrays(scenes(name), pkg.DEFAULT_TRIANGLES_PER_LEAF, w, h, **kw)


@pytest.mark.parametrize("ndev,w,h", [(2, 96, 70), (3, 64, 50), (8, 200, 131)])
def test_device_group_equals_one_device(pkg, scenes, oracle, ndev, w, h):
    """A handle over N devices renders exactly what a one-device handle renders: films, counters, the gathered
    frame, 50-row frames, camera moves — and both equal the oracle."""
    name, spp = "ico2", 3
    one = make(pkg, scenes, name, w, h, seed=4)
    grp = make(pkg, scenes, name, w, h, seed=4, device_count=ndev, flags=pkg.FLAG_GROUP_SHARES_DEVICE)
    assert grp.device_count == ndev and one.device_count == 1
    assert list(grp.owned_rows()) == list(range(h))
    orc = oracle.Oracle(scenes(name), w, h, seed=4)
    c1 = one.render(spp); cg = grp.render(spp); oc = orc.render(spp, nthreads=4)
    assert (cg.primary, cg.bounce, cg.shadow, cg.primary_hits) == (c1.primary, c1.bounce, c1.shadow, c1.primary_hits) == (oc["primary"], oc["bounce"], oc["shadow"], oc["primary_hits"])
    for a, b in zip(grp.film.pixel_datas(), one.film.pixel_datas()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(grp.get_tonemapped_pixels(), one.get_tonemapped_pixels())
    assert np.array_equal(grp.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(grp.film.get_estimated_variances()), bits(orc.get_estimated_variances()))
    # the reference binary's loop on the group (main.rs:197-207)
    for _ in range(4):
        assert grp.trace_frame_additive() == one.trace_frame_additive() == orc.trace_frame_additive() == 50 * w
        assert np.array_equal(grp.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    cg = grp.last_counts(); c1 = one.last_counts()
    assert (cg.primary, cg.bounce, cg.shadow) == (c1.primary, c1.bounce, c1.shadow)
    assert grp.current_row == one.current_row == orc.current_row
    grp.camera.add_y_angle(0.02); orc.camera_add_y_angle(0.02)
    grp.camera.move_rel(0.0, 0.1, -0.1); orc.camera_move_rel(0.0, 0.1, -0.1)
    grp.film.clear(); orc.film_clear()
    grp.trace_frame_additive(); orc.trace_frame_additive()
    assert np.array_equal(grp.get_tonemapped_pixels(), orc.get_tonemapped_pixels())
    assert np.array_equal(bits(grp.film.pixel_datas()[0]), bits(orc.film()[0]))
    with pytest.raises(RuntimeError, match="device group"):
        grp.tonemap_owned_rows_device(0, 0)
    with pytest.raises(RuntimeError, match="stripe_world"):
        make(pkg, scenes, name, w, h, device_count=2, stripe_world=2, stripe_rank=0, flags=pkg.FLAG_GROUP_SHARES_DEVICE)


def test_device_group_full_size_thai2(pkg, scenes):
    """thai2 1920x1080 on a group of 4: bit-identical to one device, frame gathered through the peer-copy path."""
    w, h, spp = 1920, 1080, 4
    one = make(pkg, scenes, "thai2", w, h, seed=1)
    grp = make(pkg, scenes, "thai2", w, h, seed=1, device_count=4, flags=pkg.FLAG_GROUP_SHARES_DEVICE)
    c1 = one.render(spp); cg = grp.render(spp)
    assert (cg.primary, cg.bounce, cg.shadow, cg.primary_hits) == (c1.primary, c1.bounce, c1.shadow, c1.primary_hits)
    assert np.array_equal(grp.get_tonemapped_pixels(), one.get_tonemapped_pixels())
    assert np.array_equal(bits(grp.film.pixel_datas()[0]), bits(one.film.pixel_datas()[0]))


def test_rccl_communicator_path_world_of_one(pkg, scenes):
    """mi355rt_comm_*: librccl.so is loaded, a communicator of one rank is created, and the gather (here only the
    root's own slot + the placement kernel) returns the frame get_tonemapped_pixels returns.  More ranks need more
    GPUs than this box has: tests/test_host_logic.py covers the slot arithmetic for 2..8 ranks."""
    name, w, h = "ico2", 72, 60
    rt = make(pkg, scenes, name, w, h, seed=2)
    rt.render(2)
    rid = pkg.comm_unique_id()
    assert len(rid) == 128 and any(rid)
    rt.comm_init(rid)
    with pytest.raises(RuntimeError, match="already"):
        rt.comm_init(rid)
    out = np.zeros(w * h, np.uint32)
    rt.comm_gather_frame(0, out)
    assert np.array_equal(out, rt.get_tonemapped_pixels())
    rt.comm_gather_frame(0, None); rt.synchronize()                  # queued-only variant
    with pytest.raises(RuntimeError, match="root"):
        rt.comm_gather_frame(3, out)
    rt.comm_destroy()
    with pytest.raises(RuntimeError, match="no communicator"):
        rt.comm_gather_frame(0, out)


def test_native_gather_wrapper_and_its_check_world_of_one(pkg, scenes):
    """stripes.NativeGather (what bench.py uses for N > 1: RCCL id handed round with torch.distributed, communicator per
    rank) and verify_against (one frame through the library's gather AND torch's all_gather, compared on rank 0) with a
    process group of one rank: the code the N > 1 bench runs before anything is timed."""
    import importlib
    import socket
    import torch
    import torch.distributed as dist
    stripes = importlib.import_module("raytracer_rs_amd.stripes")
    name, w, h = "ico2", 96, 64
    rt = make(pkg, scenes, name, w, h, seed=3, stripe_rows=8)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        native = stripes.NativeGather(pkg, rt, dist, 0, 1)
        fg = stripes.FrameGather(h, w, 8, 1, "cuda")
        stripe = fg.stripe_buffer("cuda")
        rt.render(2)
        assert native.verify_against(dist, 0, fg, stripe)
        native.gather(); rt.synchronize()
        native.close()
    finally:
        dist.destroy_process_group()


def test_cli_gpus_and_png(pkg, scenes, tmp_path):
    """bin/raytracer --gpus 3 (a device group; --share-device puts it on the one GPU here) writes the same picture as
    one GPU, as PPM and as PNG (decoded with Pillow)."""
    from PIL import Image
    exe = os.path.join(ROOT, "raytracer-rs_amd", "bin", "raytracer")
    scene = os.path.join(GOLDEN, "scenes", "ico2.scene")
    w, h = 120, 90
    png = str(tmp_path / "o.png"); ppm = str(tmp_path / "o.ppm")
    base = [exe, "-f", scene, "--width", str(w), "--height", str(h), "--seed", "3", "--spp", "2"]
    r1 = subprocess.run(base + ["--gpus", "3", "--share-device", "--out", png], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stderr
    assert "rendering on 3 GPUs" in r1.stdout and re.search(r"fps: [0-9.e+]+ +primary rays/s: ", r1.stdout)
    r2 = subprocess.run(base + ["--out", ppm], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    img = np.asarray(Image.open(png).convert("RGB")).reshape(-1, 3)
    data = open(ppm, "rb").read()
    header = ("P6\n%d %d\n255\n" % (w, h)).encode()
    rgb = np.frombuffer(data[len(header):], np.uint8).reshape(-1, 3)
    assert img.shape == rgb.shape and np.array_equal(img, rgb)
    rt = make(pkg, scenes, "ico2", w, h, seed=3)
    rt.render(2)
    ldr = rt.get_tonemapped_pixels()
    assert np.array_equal((rgb[:, 0].astype(np.uint32) << 16) | (rgb[:, 1].astype(np.uint32) << 8) | rgb[:, 2], ldr & 0xFFFFFF)
