#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X.

Default mode (`--mode frame`): a "step" is one frame of the render hot path: thai2 (20 049 triangles) at
1920x1080, 64 samples per pixel per GPU, RECURSIONS = 2 / SUB_SPREAD = 1, the reference's own pixel->ray
mapping, seed 1, followed by the tonemapped read-out into device memory and (N > 1) the gather of the row
stripes.  Scene, BVH and film are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 started WITHOUT a launcher (no WORLD_SIZE in the environment) launches itself: the parent — before it imports
torch or touches the GPU — starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free> bench.py <same arguments>` as a child process and exits with its code.  Started under a launcher
(WORLD_SIZE set) it is one of the ranks.

  --scaling weak (default)  every GPU traces 1920*1080*64 primary samples; with N GPUs the frame is the same image
                            at 64*N spp, rows dealt to the ranks in stripes.
  --scaling strong          the frame is FIXED (1920x1080x64 spp, or --config c5: 3840x2160x256 spp, BASELINE config 5)
                            and its rows are dealt to the N ranks.
  --mode dropin             the loop the reference binary runs (main.rs:197-207) at its defaults (thai2, 1024x768):
                            a step is trace_frame_additive() + get_tonemapped_pixels(); metric = primary rays/s,
                            the reference's own stats.rs:27 definition.  N = 1 only.

Rank 0 prints ONE JSON line (see the repo instructions for the contract).  The default run (weak, c4) carries three result
groups: the line itself (weak scaling: what the driver's scaling curve is made of), `strong` (the fixed 1080p x 64 frame
dealt to the N ranks) and `c5` (BASELINE config 5, the fixed 3840x2160 x 256 frame dealt to the N ranks), each with its own
ms_per_step, gather_ms_rank0, gather_path ("none" | "library_rccl" | "torch_all_gather"), rccl_ranks (what ncclCommCount
returned) and gather_error (null unless the library's RCCL path was refused and torch's all_gather stood in).
Extra objects of the line:
  roofline      dominant kernel = trace_kernel (closest-hit traversal), secondary launches.  It is bound by the CU's VECTOR MEMORY
                PIPE (texture addresser / L1 / data return: TA busy 0.91, TD busy 0.95 of its cycles), not by HBM (the 1.5 MB scene
                is cache-resident), not by VALU issue (12 % fewer instructions in the hot block change nothing, one more divergent
                load per step costs 22 %: profiles/r03_notes.md): `bound` = "vmem_gather", unit = G cache-line accesses/s;
                achieved = TCP_TOTAL_CACHE_ACCESSES per secondary ray (a rocprofv3 --pmc child pass of THIS run) x secondary rays
                           per launch / the average duration of a secondary trace launch (HIP events, this process);
                peak     = the rate at which this device's pipe serves the same kind of fetch and nothing else, measured live
                           (mi355rt_debug_gather_rate: 8 waves per SIMD walk random 32-byte nodes of an L2-resident table with the
                           inner step's two 16-byte loads); frac = achieved / peak.
                Next to it: TA / TD busy fractions, the VALU issue rate against the REAL peak (one wave64 instruction per 2 cycles
                per SIMD, MI355X_MICROARCH.md; round 2 divided by 4 cycles), the clock the trace waves were observed to run at,
                the useful-lane fractions of the inner-node and triangle sections (live, MI355RT_FLAG_COUNT_STEPS), the SURVEY.md 8d
                LOGICAL byte rate (labelled logical: it exceeds the HBM peak because node and triangle bytes are cache
                hits) and the measured HBM traffic per launch (`traffic`, PMC FETCH_SIZE x 2 + WRITE_SIZE as
                MI355X_MICROARCH.md prescribes for gfx950).  PMC fields are null when the child passes are skipped
                (--no-pmc, N > 1) or fail.
  cpu_baseline  the CPU oracle (C restatement of the reference: octree, recursive radiance) timed on this box's
                host cores on a bounded sample of the same workload.
  dropin        (N = 1) the loop the reference binary runs, trace_frame_additive() + get_tonemapped_pixels() at 1024x768:
                ms per step, primary rays/s, and the CPU oracle's same loop beside it.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE = "thai2"
WIDTH, HEIGHT, SPP = 1920, 1080, 64
STRIPE_ROWS = 2                 # rows per stripe of the deal over ranks and slices (--stripe-rows); measured: profiles/r02_notes.md
HBM_PEAK_GBS = 8000.0
NUM_SIMDS = 1024            # 256 CUs x 4
NUM_SE = 32                 # 8 XCDs x 4 shader engines: SQ_BUSY_CYCLES is summed over them
NOMINAL_MHZ = 2400.0        # MI355X_MICROARCH.md: max clock


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["frame", "dropin"], default="frame")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--config", choices=["c4", "c5"], default="c4", help="c4: 1920x1080x64 (headline); c5: 3840x2160x256 (strong scaling only)")
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline PMC fields become null)")
    ap.add_argument("--no-groups", action="store_true", help="default run only: skip the `strong` and `c5` result groups")
    ap.add_argument("--no-dropin", action="store_true", help="default N = 1 run only: skip the `dropin` object")
    ap.add_argument("--native-gather", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: gather the stripes with the library's own RCCL path (on), torch.distributed (off), or try native first")
    ap.add_argument("--stripe-rows", type=int, default=0, help="rows per stripe of the deal over ranks (and of the blocks dealt to the frame slices); 0: the default")
    ap.add_argument("--device-lbvh", action="store_true", help="MI355RT_FLAG_DEVICE_LBVH: BVH built on the device (Morton order) instead of the host's SAH build")
    ap.add_argument("--true-closest-hit", action="store_true",
                    help="MI355RT_FLAG_TRUE_CLOSEST_HIT: NoAccelerationIntersector semantics (no octree confirm step) instead of the reference's default intersector")
    ap.add_argument("--fix-row-index", action="store_true",
                    help="v = idx / width instead of the reference's idx / height (SURVEY.md 8d: reported next to the headline, never as it)")
    ap.add_argument("--slices", type=int, default=0, help="concurrent frame slices of the timed frames (0: library default)")
    ap.add_argument("--spawn", action="store_true", help="launch the rank processes from here even for N = 1 (what N > 1 without a launcher does)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.stripe_rows > 0:
        global STRIPE_ROWS
        STRIPE_ROWS = args.stripe_rows
    if args.steps is None:
        args.steps = 10 if args.mode == "frame" else 2000          # 3 warm-up frames, then >= 10 timed
    if args.warmup is None:
        args.warmup = 3 if args.mode == "frame" else 100
    return args


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks from here.  The parent has imported neither torch nor the library and has made
# no HIP call; the ranks are fresh child processes (never an exec of a process that has touched the GPU).
def spawn_command(args, argv):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    child_args = [a for a in argv if a != "--spawn"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + child_args


def self_launch(args, argv):
    cmd = spawn_command(args, argv)
    if os.environ.get("MI355RT_BENCH_DRY_SPAWN") == "1":           # tests: what would be started, and what the parent has loaded
        print(json.dumps({"spawn": cmd, "torch_imported": "torch" in sys.modules, "library_imported": "raytracer_rs_amd" in sys.modules}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: what RCCL needs between processes on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------------------------
# rocprofv3 --pmc child passes (N = 1, rank 0, BEFORE this process touches the GPU): the same frame under counter
# collection, one pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass).
PMC_GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum"],
              ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_VMEM_RD", "TCP_GATE_EN1_sum"],
              ["FETCH_SIZE"], ["WRITE_SIZE"]]
PMC_CHILD_FRAMES = 2
GATHER_TABLE_NODES, GATHER_STEPS = 48000, 2000     # mi355rt_debug_gather_rate: a 1.5 MB table (L2-resident like the scene), 2000 dependent fetches per lane


def pmc_child(args):
    """What runs under rocprofv3: two frames of the benchmark workload with ONE slice (so that kernels do not overlap)."""
    import __graft_entry__ as ge
    import importlib
    pkg = ge.load_package()
    scene_io = importlib.import_module("raytracer_rs_amd.scene_io")
    scene = scene_io.load_scene_file(os.path.join(ge.SCENES, args.scene + ".scene"))
    if args.mode == "dropin":
        rt = pkg.create_raytracer_from_arrays(scene, pkg.DEFAULT_TRIANGLES_PER_LEAF, 1024, 768, seed=1)
        for _ in range(40):
            rt.trace_frame_additive()
        rt.get_tonemapped_pixels()
        return
    rt = pkg.create_raytracer_from_arrays(scene, pkg.DEFAULT_TRIANGLES_PER_LEAF, WIDTH, HEIGHT, seed=1, stripe_rows=STRIPE_ROWS,
                                          flags=(pkg.FLAG_FIX_ROW_INDEX if args.fix_row_index else 0) | (pkg.FLAG_TRUE_CLOSEST_HIT if args.true_closest_hit else 0))
    rt.set_slices(1)
    for _ in range(PMC_CHILD_FRAMES):
        rt.film.clear()
        rt.render(args.spp or SPP)
    rt.debug_gather_rate(GATHER_TABLE_NODES, GATHER_STEPS)         # the roofline's peak kernel under the same counters (3 launches)


def run_pmc_passes(args, kernel_substrs):
    """-> {substr: {counter: (sum over dispatches, dispatches)}} for the kernels whose name contains each substr; ({}, note) on failure."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = {k: {} for k in kernel_substrs}
    base = tempfile.mkdtemp(prefix="mi355rt_pmc_", dir="/tmp")
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    child = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.join(ROOT, "bench.py"), "--pmc-child",
             "--mode", args.mode, "--scene", args.scene] + (["--fix-row-index"] if args.fix_row_index else []) + (["--true-closest-hit"] if args.true_closest_hit else []) \
        + (["--spp", str(args.spp)] if args.spp else []) + (["--stripe-rows", str(args.stripe_rows)] if args.stripe_rows else [])
    note = None
    try:
        for gi, group in enumerate(PMC_GROUPS):
            d = os.path.join(base, "g%d" % gi)
            cmd = [exe, "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                note = "rocprofv3 pass %d timed out" % gi
                break
            if r.returncode != 0:
                note = "rocprofv3 pass %d failed (rc %d): %s" % (gi, r.returncode, (r.stderr or "")[-200:].replace("\n", " "))
                break
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    for sub in kernel_substrs:
                        if sub in row["Kernel_Name"]:
                            acc = out[sub].setdefault(row["Counter_Name"], [0.0, set()])
                            acc[0] += float(row["Counter_Value"]); acc[1].add(row["Dispatch_Id"])
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return {sub: {k: (v[0], len(v[1])) for k, v in d.items()} for sub, d in out.items()}, note


def cpu_model():
    model, phys = "unknown", None
    try:
        cores = set(); pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, phys, usable


def oracle_build_flags(ge):
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                return line.split("=", 1)[1].strip()
    except OSError:
        pass
    return None


# ------------------------------------------------------------------------------------------------------------------
class Ctx:
    """what every result group needs: the package, the scene, the process group"""


class FrameSetup:
    """One handle of a result group + its gather transport.

    gather_path   "none" (1 GPU) | "library_rccl" (mi355rt_comm_*: grouped ncclSend / ncclRecv to rank 0 inside the library,
                  checked against torch's all_gather on one frame before anything is timed) | "torch_all_gather"
    gather_error  null, or why the library's RCCL path was not used although it was asked for (never hidden in a string elsewhere)
    rccl_ranks    ncclCommCount of the library's communicator (0 without one)"""

    def __init__(self, ctx, width, height, spp, flags):
        import torch
        pkg, args = ctx.pkg, ctx.args
        self.ctx, self.width, self.height, self.spp = ctx, width, height, spp
        self.rt = pkg.create_raytracer_from_arrays(ctx.scene, pkg.DEFAULT_TRIANGLES_PER_LEAF, width, height, seed=1, device=ctx.local_rank,
                                                   stripe_rows=STRIPE_ROWS, stripe_rank=ctx.rank, stripe_world=ctx.world, flags=flags)
        self.base_flags = (pkg.FLAG_FIX_ROW_INDEX if args.fix_row_index else 0) | flags
        self.rt.set_flags(self.base_flags)
        if args.slices:
            self.rt.set_slices(args.slices)
        self.slices = self.rt.get_slices()
        self.rows = self.rt.owned_rows()
        assert list(self.rows) == ctx.stripes.owned_rows(height, STRIPE_ROWS, ctx.rank, ctx.world)
        self.fg = ctx.stripes.FrameGather(height, width, STRIPE_ROWS, ctx.world, "cuda")
        self.stripe = self.fg.stripe_buffer("cuda")
        self.cur_stream = torch.cuda.current_stream().cuda_stream
        self.native = None
        self.gather_path, self.gather_error, self.rccl_ranks = ("none" if ctx.world == 1 else "torch_all_gather"), None, 0
        if ctx.world > 1 and args.native_gather != "off":
            try:
                self.native = ctx.stripes.NativeGather(pkg, self.rt, ctx.dist, ctx.rank, ctx.world)
                self.rt.film.clear(); self.rt.render(1)                # one cheap frame through both transports before anything is timed
                if not self.native.verify_against(ctx.dist, ctx.rank, self.fg, self.stripe):
                    self.native.close(); self.native = None
                    raise RuntimeError("the library's RCCL gather did not reproduce the all_gather frame")
                self.gather_path, self.rccl_ranks = "library_rccl", self.native.ranks
            except Exception as e:                                     # noqa: BLE001 — every rank takes this branch together (NativeGather agrees on failures)
                if args.native_gather == "on":
                    raise
                self.native = None
                self.gather_error = str(e)[:300]
        self.gather_ms = []

    def sync(self):
        import torch
        self.rt.synchronize()                                           # the library's own streams (frames, the native gather)
        if self.ctx.dist is not None:
            self.ctx.dist.barrier()
        torch.cuda.synchronize()

    def step(self, time_gather=False, wait=True):
        import torch
        rt = self.rt
        rt.film.clear()
        counts = rt.render(self.spp, wait=wait)                         # wait: returns when the frame is traced; else queued only (mi355rt_render_async)
        t0 = time.perf_counter()
        if self.native is not None:
            self.native.gather()                                        # tonemap + RCCL inside the library, on its own stream
        else:
            rt.tonemap_owned_rows_device(self.stripe.data_ptr(), self.rows.size * self.width, stream=self.cur_stream)   # ordered on torch's stream
            self.fg.gather(self.ctx.dist, self.stripe)
        if time_gather:
            rt.synchronize(); torch.cuda.synchronize()
            self.gather_ms.append((time.perf_counter() - t0) * 1e3)
        return counts

    def timed(self, steps, warmup):
        """`warmup` untimed frames, then EXACTLY `steps` frames between two (barrier + device synchronisation)s;
        -> elapsed seconds (max over ranks), ray sums over ranks, rank-0 per-step host times of the (synchronous) warm-up frames.
        The timed frames are QUEUED (mi355rt_render_async + the gather on its stream): the device runs them back to back and the
        closing synchronisation waits for all of them.  Every frame is the same frame (film cleared, same seed): its ray counters
        are read once, after the loop, checked against a synchronous frame's, and counted `steps` times."""
        import torch
        ref = None
        step_ms = []
        for _ in range(max(warmup, 1)):
            ts = time.perf_counter()
            ref = self.step()
            step_ms.append((time.perf_counter() - ts) * 1e3)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(wait=False)
        self.sync()
        elapsed = time.perf_counter() - t0
        c = self.rt.last_counts()
        same = (c.primary, c.bounce, c.shadow, c.primary_culled, c.shadow_skipped) == (ref.primary, ref.bounce, ref.shadow, ref.primary_culled, ref.shadow_skipped)
        if not same:
            raise SystemExit("bench: a queued frame traced other rays than the synchronous one: %s vs %s" % (c.as_dict(), ref.as_dict()))
        tot = [float(steps * c.primary), float(steps * c.bounce), float(steps * c.shadow), float(steps * c.primary_culled), float(steps * c.shadow_skipped)]
        for _ in range(2):                                              # the gather timed on its own (render excluded)
            self.step(time_gather=True)
        self.sync()
        stats = torch.tensor([elapsed] + tot, dtype=torch.float64, device="cuda")
        dist = self.ctx.dist
        if dist is not None:
            tmax = stats[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sums = stats[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            elapsed = float(tmax[0]); vals = [float(x) for x in sums]
        else:
            vals = [float(x) for x in stats[1:]]
        return elapsed, vals, step_ms

    def summary(self, workload, scaling, steps, warmup, elapsed, vals):
        primary, bounce, shadow, culled, skipped = vals
        total = primary + bounce + shadow
        return {"workload": workload, "scaling": scaling, "steps": steps, "warmup": warmup,
                "ms_per_step": round(elapsed / steps * 1e3, 3), "value": round(total / elapsed / 1e6, 2), "unit": "Mrays/s",
                "traced_mrays_per_s": round((total - culled - skipped) / elapsed / 1e6, 2),
                "gather_ms_rank0": round(sorted(self.gather_ms)[len(self.gather_ms) // 2], 3) if self.gather_ms else None,
                "gather_path": self.gather_path, "rccl_ranks": self.rccl_ranks, "gather_error": self.gather_error,
                "slices": self.slices, "hbm_allocated_bytes_rank0": self.rt.hbm_allocated_bytes()}

    def close(self):
        if self.native is not None:
            self.native.close(); self.native = None
        self.rt = None


def workload_text(args, width, height, spp, base_spp, world, scaling, c5=False):
    if scaling == "strong":
        w = "%s %dx%d x %d spp FIXED frame dealt to %d GPU(s) in row stripes of %d (strong scaling%s)" % (
            args.scene, width, height, spp, world, STRIPE_ROWS, ", BASELINE config 5" if c5 else "")
    else:
        w = "%s %dx%d, %d spp per GPU (frame = %d spp), rows dealt in stripes of %d" % (args.scene, width, height, base_spp, spp, STRIPE_ROWS)
    w += ", recursions 2 / spread 1, " + ("row index FIXED (v = idx / width)" if args.fix_row_index else "reference pixel mapping")
    w += ", intersector semantics: " + ("true closest hit (NoAccelerationIntersector, opt-out flag)" if args.true_closest_hit else
                                        "the reference's default OctTreeIntersector at 70 triangles per leaf (BVH + octree confirm step)")
    return w


def extra_group(ctx, name, sem_flag):
    """the `strong` / `c5` result groups of the default run: the fixed frame dealt to the ranks, a few frames"""
    args = ctx.args
    width, height, spp = (3840, 2160, 256) if name == "c5" else (WIDTH, HEIGHT, SPP)
    steps, warmup = (3, 1) if name == "c5" else (min(args.steps, 5), 1)
    fs = FrameSetup(ctx, width, height, spp, sem_flag)
    elapsed, vals, _ = fs.timed(steps, warmup)
    out = fs.summary(workload_text(args, width, height, spp, spp, ctx.world, "strong", c5=name == "c5"), "strong", steps, warmup, elapsed, vals)
    fs.close()
    return out


# ------------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.pmc_child:
        return pmc_child(args)
    if args.mode == "dropin" and args.gpus != 1:
        raise SystemExit("--mode dropin is a single-GPU loop")
    if args.config == "c5" and args.scaling != "strong":
        raise SystemExit("--config c5 is the fixed 3840x2160x256 frame: use --scaling strong")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(self_launch(args, argv))

    # stdout carries ONE line, the JSON: everything else that writes to file descriptor 1 from here on (RCCL's version banner, gloo's
    # connection messages, a library's printf) goes to stderr; emit() writes the line to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # counter passes first: this process has not initialised the GPU yet, so the children have the device to themselves
    pmc, pmc_gather, pmc_note = {}, {}, ("skipped (--no-pmc)" if args.no_pmc else "skipped (N > 1)")
    if not args.no_pmc and world == 1:
        main_kernel = "fused_pass_kernel" if args.mode == "dropin" else "trace_kernel<false, false"
        both, pmc_note = run_pmc_passes(args, [main_kernel, "gather_rate_kernel"])
        pmc, pmc_gather = both.get(main_kernel, {}), both.get("gather_rate_kernel", {})

    import torch                                                       # before the library: see raytracer-rs_amd/__init__.py lib()
    import __graft_entry__ as ge

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    share_gpu = os.environ.get("MI355RT_BENCH_SHARE_GPU") == "1"       # rehearsal of the N > 1 code path on a one-GPU box: every rank on cuda:0, gloo
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = ge.load_package()
    import importlib
    scene_io = importlib.import_module("raytracer_rs_amd.scene_io")
    scene = scene_io.load_scene_file(os.path.join(ge.SCENES, args.scene + ".scene"))
    if args.mode == "dropin":
        out = dropin_measure(args, ge, pkg, scene, pmc, pmc_note, full=True)
        emit(out)
        return

    ctx = Ctx()
    ctx.args, ctx.pkg, ctx.scene, ctx.dist, ctx.rank, ctx.world, ctx.local_rank = args, pkg, scene, dist, rank, world, local_rank
    ctx.stripes = importlib.import_module("raytracer_rs_amd.stripes")

    width, height = (3840, 2160) if args.config == "c5" else (WIDTH, HEIGHT)
    base_spp = args.spp or (256 if args.config == "c5" else SPP)
    spp = base_spp * world if args.scaling == "weak" else base_spp
    sem_flag = (pkg.FLAG_TRUE_CLOSEST_HIT if args.true_closest_hit else 0) | (pkg.FLAG_DEVICE_LBVH if args.device_lbvh else 0)
    fs = FrameSetup(ctx, width, height, spp, sem_flag)
    rt, base_flags = fs.rt, fs.base_flags

    # instrumented frame (untimed): nodes visited / triangles tested per ray, useful-lane fractions of the two sections,
    # and the clock the trace waves run at (s_memtime against s_memrealtime, stamped by the instrumented build only)
    rt.set_flags(base_flags | pkg.FLAG_COUNT_STEPS)
    rt.film.clear()
    c = rt.render(max(1, min(4, spp)))
    inst_rays = max(1, c.primary - c.primary_culled + c.bounce + c.shadow - c.shadow_skipped)       # traced rays
    nodes_per_ray = c.nodes_visited / inst_rays
    tris_per_ray = c.tris_tested / inst_rays
    lane_inner = c.nodes_visited / (64.0 * c.inner_execs) if c.inner_execs else None
    lane_leaf = c.tris_tested / (64.0 * c.leaf_execs) if c.leaf_execs else None
    observed_mhz = c.shader_clock_mhz or None
    rt.set_flags(base_flags)

    elapsed, vals, step_ms = fs.timed(args.steps, args.warmup)
    primary, bounce, shadow, culled, skipped = vals
    total_rays = primary + bounce + shadow

    # Kernel-timing loop for the roofline: the same frames again with ONE slice and HIP events around every
    # trace launch (on the stream it is launched on).  The frames above run several slices concurrently:
    # their kernels share the chip and have no individual duration, so the per-launch time is taken here.
    rt.set_slices(1)
    rt.set_flags(base_flags | pkg.FLAG_TIME_KERNELS)
    fs.step()
    ksteps = max(1, min(args.steps, 3))
    fs.sync()
    t1 = time.perf_counter()
    k = dict(rays=0, sec_rays=0, trace_ms=0.0, sec_ms=0.0, launches=0, sec_launches=0)
    for _ in range(ksteps):
        c = fs.step()
        k["rays"] += c.primary - c.primary_culled + c.bounce + c.shadow - c.shadow_skipped; k["sec_rays"] += c.bounce + c.shadow - c.shadow_skipped      # rays the launches actually trace
        k["trace_ms"] += c.trace_ms; k["sec_ms"] += c.trace_secondary_ms
        k["launches"] += c.trace_launches; k["sec_launches"] += c.trace_secondary_launches
    fs.sync()
    serial_ms_per_step = (time.perf_counter() - t1) / ksteps * 1e3
    rt.set_flags(base_flags)
    gather = None
    if rank == 0:
        try:
            gather = rt.debug_gather_rate(48000, 2000)              # 1.5 MB table: L2-resident like the scene
        except Exception as e:                                      # noqa: BLE001 — the line then carries a null peak
            gather = None
    main_group = fs.summary(workload_text(args, width, height, spp, base_spp, world, args.scaling, c5=args.config == "c5"), args.scaling, args.steps, args.warmup, elapsed, vals)
    acc = rt.accel_stats()
    build_info = rt.bvh_build_info()
    slices = fs.slices
    fs.close(); del rt

    # the other intersector semantics on the same frame (short: 1 warm + 3 timed frames), for the record in the line
    other = None
    if world == 1:
        oflag = (0 if args.true_closest_hit else pkg.FLAG_TRUE_CLOSEST_HIT) | (pkg.FLAG_DEVICE_LBVH if args.device_lbvh else 0)
        rt2 = pkg.create_raytracer_from_arrays(scene, pkg.DEFAULT_TRIANGLES_PER_LEAF, width, height, seed=1, device=local_rank, stripe_rows=STRIPE_ROWS, flags=oflag)
        rt2.set_flags(oflag | (pkg.FLAG_FIX_ROW_INDEX if args.fix_row_index else 0))
        if args.slices:
            rt2.set_slices(args.slices)
        rt2.film.clear(); rt2.render(spp)
        t2 = time.perf_counter(); r2 = 0
        for _ in range(3):
            rt2.film.clear(); c2 = rt2.render(spp); r2 += c2.primary + c2.bounce + c2.shadow
        dt2 = time.perf_counter() - t2
        other = {"semantics": "reference default (octree)" if args.true_closest_hit else "true closest hit (MI355RT_FLAG_TRUE_CLOSEST_HIT)",
                 "ms_per_step": round(dt2 / 3 * 1e3, 3), "mrays_per_s": round(r2 / dt2 / 1e6, 2)}
        del rt2

    # the other two result groups of the default run (collective: every rank takes part)
    groups = {}
    if args.scaling == "weak" and args.config == "c4" and not args.no_groups and not args.spp:
        for name in ("strong", "c5"):
            groups[name] = extra_group(ctx, name, sem_flag)

    if rank == 0:
        node_bytes = acc["node_bytes"] / max(acc["nodes"], 1)
        bytes_per_ray = node_bytes * nodes_per_ray + 48.0 * tris_per_ray + 96.0
        prim_launches = k["launches"] - k["sec_launches"]
        sec_launch_s = k["sec_ms"] * 1e-3 / max(k["sec_launches"], 1)
        prim_launch_s = (k["trace_ms"] - k["sec_ms"]) * 1e-3 / max(prim_launches, 1)
        logical_gbs = k["rays"] * bytes_per_ray / (k["trace_ms"] * 1e-3) / 1e9 if k["trace_ms"] > 0 else 0.0
        peak_winst = NUM_SIMDS * NOMINAL_MHZ * 1e6 / 2.0 / 1e9                     # G wave64 VALU instructions per second (2-cycle throughput)
        sec_rays_launch = k["sec_rays"] / max(k["sec_launches"], 1)
        roof = {"bound": "vmem_gather", "kernel": "trace_kernel<secondary>", "unit": "G cache-line accesses/s",
                "bound_is": "the CU's vector memory pipe (TA / TCP / TD) serving divergent 16-byte fetches of cache-resident nodes and triangles: the counters, the "
                            "A/B experiments and the microbenchmark are in profiles/r03_notes.md",
                "peak": None,
                "peak_is": "measured on this device in this run (mi355rt_debug_gather_rate): every lane of 8 waves per SIMD walks a dependent chain of random 32-byte "
                           "nodes of a 1.5 MB table with two 16-byte loads per node, grid = the whole chip; its TCP_TOTAL_CACHE_ACCESSES per launch (the same counter, the same "
                           "rocprofv3 pass as `achieved`; the counter reads 2 per lane and 16-byte load) / its duration in this process" + (" (%.3f ms)" % gather["ms"] if gather else ""),
                "achieved": None, "frac": None,
                "achieved_is": "TCP_TOTAL_CACHE_ACCESSES per secondary ray (rocprofv3 --pmc child of this run) x secondary rays per launch / average duration of a secondary trace launch (HIP events, this process)",
                "clock_mhz_nominal": NOMINAL_MHZ, "clock_mhz_observed": round(observed_mhz, 1) if observed_mhz else None,
                "clock_observed_is": "sum of delta s_memtime / sum of delta s_memrealtime x 100 MHz over all waves of the instrumented frame's trace launches",
                "useful_lane_frac_inner": round(lane_inner, 3) if lane_inner else None, "useful_lane_frac_leaf": round(lane_leaf, 3) if lane_leaf else None,
                "secondary_launches": int(k["sec_launches"]), "avg_secondary_launch_ms": round(sec_launch_s * 1e3, 4),
                "primary_launches": int(prim_launches), "avg_primary_launch_ms": round(prim_launch_s * 1e3, 4),
                "secondary_rays_per_launch": round(sec_rays_launch, 1),
                "nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2), "node_bytes": node_bytes,
                "logical": {"what": "SURVEY.md 8d algorithmic bytes: node_bytes x nodes + 48 x triangles + 96 per ray; node and triangle bytes are CACHE hits "
                                    "(1.5 MB scene), so this is not HBM traffic and may exceed the HBM peak",
                            "bytes_per_ray": round(bytes_per_ray, 1), "bytes_per_launch": round(k["rays"] * bytes_per_ray / max(k["launches"], 1), 1),
                            "gbs": round(logical_gbs, 1), "frac_of_hbm_peak": round(logical_gbs / HBM_PEAK_GBS, 4)},
                "traffic": None, "hbm": None,
                "timing": "HIP events around every trace launch in a second loop of %d frames with 1 slice (%.2f ms/frame); the headline frames run %d "
                          "concurrent slices whose kernels overlap" % (ksteps, serial_ms_per_step, slices),
                "pmc": {"source": "rocprofv3 --pmc child passes of this same run (bench.py --pmc-child: %d frames, 1 slice), secondary trace launches; counters are used PER RAY "
                                  "(same frame, same seed, same passes as the timed process)" % PMC_CHILD_FRAMES, "note": pmc_note}}
        child_sec_rays = (bounce + shadow - skipped) / max(args.steps * world, 1) * PMC_CHILD_FRAMES       # secondary rays the child's frames trace (same frame, same seed)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc and child_sec_rays > 0:
            acc_per_ray = pmc["TCP_TOTAL_CACHE_ACCESSES_sum"][0] / child_sec_rays
            roof["pmc"]["cache_line_accesses_per_secondary_ray"] = round(acc_per_ray, 2)
            if "SQ_INSTS_VMEM_RD" in pmc:
                roof["pmc"]["vmem_read_winst_per_secondary_ray"] = round(pmc["SQ_INSTS_VMEM_RD"][0] / child_sec_rays, 3)
                roof["pmc"]["cache_line_accesses_per_vmem_read"] = round(pmc["TCP_TOTAL_CACHE_ACCESSES_sum"][0] / max(pmc["SQ_INSTS_VMEM_RD"][0], 1), 1)
            if "TCP_TCC_READ_REQ_sum" in pmc:
                roof["pmc"]["l1_miss_per_access"] = round(pmc["TCP_TCC_READ_REQ_sum"][0] / max(pmc["TCP_TOTAL_CACHE_ACCESSES_sum"][0], 1), 3)
            if gather and "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc_gather:
                ga, gn = pmc_gather["TCP_TOTAL_CACHE_ACCESSES_sum"]
                roof["peak"] = round(ga / max(gn, 1) / (gather["ms"] * 1e-3) / 1e9, 1)
                roof["pmc"]["gather_kernel"] = {"cache_line_accesses_per_launch": ga / max(gn, 1), "launches": gn,
                                                "accesses_per_lane_and_load": round(ga / max(gn, 1) / (gather["line_accesses_per_s"] * gather["ms"] * 1e-3), 3)}
            if sec_launch_s > 0:
                achieved = acc_per_ray * sec_rays_launch / sec_launch_s / 1e9
                roof["achieved"] = round(achieved, 1)
                if roof["peak"]:
                    roof["frac"] = round(achieved / roof["peak"], 4)
        if "SQ_INSTS_VALU" in pmc and child_sec_rays > 0:
            v, n = pmc["SQ_INSTS_VALU"]
            roof["pmc"]["dispatches"] = n
            roof["pmc"]["valu_winst_per_secondary_ray"] = round(v / child_sec_rays, 2)
            roof["pmc"]["salu_inst_per_secondary_ray"] = round(pmc.get("SQ_INSTS_SALU", (0, 0))[0] / child_sec_rays, 2)
            if sec_launch_s > 0:
                valu_rate = v / child_sec_rays * sec_rays_launch / sec_launch_s / 1e9
                roof["valu"] = {"achieved_g_winst_per_s": round(valu_rate, 1), "peak_g_winst_per_s": round(peak_winst, 1), "frac": round(valu_rate / peak_winst, 4),
                                "peak_is": "%d SIMDs x %.0f MHz / 2 cycles per wave64 VALU instruction (throughput with several waves per SIMD, MI355X_MICROARCH.md; "
                                           "4 cycles is ONE wave's issue cost — the divisor round 2 used, which put the primary trace kernel at 1.05)" % (NUM_SIMDS, NOMINAL_MHZ)}
            if "SQ_BUSY_CYCLES" in pmc and pmc["SQ_BUSY_CYCLES"][0] > 0:
                busy = pmc["SQ_BUSY_CYCLES"][0] / NUM_SE                          # cycles during which the shader engines had waves
                roof["pmc"]["valu_issue_per_simd_cycle"] = round(v / NUM_SIMDS / busy, 4)
                if "TA_TA_BUSY_sum" in pmc:
                    roof["pmc"]["ta_busy_frac"] = round(pmc["TA_TA_BUSY_sum"][0] / 256.0 / busy, 3)
                if "TD_TD_BUSY_sum" in pmc:
                    roof["pmc"]["td_busy_frac"] = round(pmc["TD_TD_BUSY_sum"][0] / 256.0 / busy, 3)
                roof["pmc"]["busy_is"] = "per CU and busy cycle: TA_TA_BUSY_sum (TD_TD_BUSY_sum) / 256 CUs over SQ_BUSY_CYCLES / %d shader engines, same pass" % NUM_SE
            if "SQ_WAIT_ANY" in pmc and pmc.get("SQ_WAVE_CYCLES", (0, 0))[0] > 0:
                roof["pmc"]["wave_wait_frac"] = round(pmc["SQ_WAIT_ANY"][0] / pmc["SQ_WAVE_CYCLES"][0], 3)
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and child_sec_rays > 0:
            f, nf = pmc["FETCH_SIZE"]; w, nw = pmc["WRITE_SIZE"]
            per_ray = (2.0 * f + w) * 1024.0 / child_sec_rays                                # KB -> B; FETCH_SIZE doubled (gfx950 wide reads)
            per_launch = per_ray * sec_rays_launch
            roof["traffic"] = round(per_launch, 1)
            roof["hbm"] = {"bytes_per_launch": round(per_launch, 1), "bytes_per_secondary_ray": round(per_ray, 2), "gbs": round(per_launch / sec_launch_s / 1e9, 1) if sec_launch_s > 0 else None,
                           "frac_of_hbm_peak": round(per_launch / sec_launch_s / 1e9 / HBM_PEAK_GBS, 4) if sec_launch_s > 0 else None,
                           "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / secondary rays of the child's frames x secondary rays per launch"}
        out = {
            "metric": "Mrays/s (whole node) + ms/frame, 1920x1080x64spp thai2.dae",
            "value": main_group["value"],
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_group["ms_per_step"],
            "ms_per_synchronous_step_rank0": round(min(step_ms), 3),
            "pipelining": "the %d timed frames are queued (mi355rt_render_async) between the two synchronisations and run back to back on the device; "
                          "ms_per_synchronous_step_rank0 is the best warm-up frame, rendered with a host wait per frame" % args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": main_group["workload"], "scene": args.scene, "width": width, "height": height, "spp_per_gpu": spp if args.scaling == "strong" else base_spp,
                       "frame_spp": spp, "seed": 1, "slices": slices, "parallelism": "row stripes x%d" % world},
            "gather_path": main_group["gather_path"], "rccl_ranks": main_group["rccl_ranks"], "gather_error": main_group["gather_error"],
            "gather_ms_rank0": main_group["gather_ms_rank0"],
            "hbm_allocated_bytes": main_group["hbm_allocated_bytes_rank0"],
            "primary_mrays_per_s": round(primary / elapsed / 1e6, 2),
            "traced_mrays_per_s": main_group["traced_mrays_per_s"],
            "traced_note": "value counts every ray the reference casts (primary samples by its stats.rs:27 definition, reflection rays, shadow rays); traced_mrays_per_s leaves out "
                           "the primary samples of chunks the culling skipped without tracing (%.1f %% of the primary samples) and the shadow rays the lights' depth maps proved "
                           "free (%.1f %% of the shadow rays): work whose outcome was known" % (100.0 * culled / max(primary, 1), 100.0 * skipped / max(shadow, 1)),
            "rays_per_frame": {"primary": primary / args.steps, "bounce": bounce / args.steps, "shadow": shadow / args.steps, "primary_culled": culled / args.steps, "shadow_skipped": skipped / args.steps},
            "other_semantics": other,
            "roofline": roof,
            "accel": dict(acc, builder=("device LBVH (MI355RT_FLAG_DEVICE_LBVH)" if build_info["on_device"] else "host binned SAH"),
                          device_build_ms=build_info["device_ms"]),
        }
        out.update(groups)
        if world == 1 and not args.no_dropin and args.scaling == "weak" and args.config == "c4":
            d = dropin_measure(args, ge, pkg, scene, {}, "not collected in frame mode", full=False)
            out["dropin"] = d
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ge, scene, width, height, args.fix_row_index)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
def dropin_measure(args, ge, pkg, scene, pmc, pmc_note, full):
    """The reference binary's loop (main.rs:197-207) at its defaults: thai2, 1024x768; one step = trace_frame_additive()
    (50 rows x 1 sample = 51 200 primary rays) + get_tonemapped_pixels() (a fresh copy of the whole frame in host memory).
    full: the whole --mode dropin line (args.steps steps); else the short `dropin` object of the frame-mode line."""
    import numpy as np
    import torch
    w, h = 1024, 768
    steps, warmup = (args.steps, args.warmup) if full else (500, 50)
    extra = {"stripe_rows": args.stripe_rows} if args.stripe_rows > 0 else {}
    rt = pkg.create_raytracer_from_arrays(scene, pkg.DEFAULT_TRIANGLES_PER_LEAF, w, h, seed=1, **extra)
    buf = np.empty(w * h, np.uint32)

    def step():
        n = rt.trace_frame_additive()
        rt.get_tonemapped_pixels(buf)
        return n

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    primary = 0
    for _ in range(steps):
        primary += step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the two halves on their own
    t1 = time.perf_counter()
    for _ in range(steps):
        rt.trace_frame_additive()
    rt.last_counts()                                                   # waits for the device
    trace_only = (time.perf_counter() - t1) / steps
    t2 = time.perf_counter()
    for _ in range(min(steps, 200)):
        rt.get_tonemapped_pixels(buf)                                  # nothing dirty: the host copy alone
    copy_only = (time.perf_counter() - t2) / min(steps, 200)
    # kernel time (HIP events around the fused launch) and rays per call
    rt.set_flags(pkg.FLAG_TIME_KERNELS)
    kms, rays, kcalls = 0.0, 0, 200
    for _ in range(kcalls):
        rt.trace_frame_additive()
        c = rt.last_counts()
        kms += c.trace_ms; rays += c.primary + c.bounce + c.shadow
    rt.set_flags(0)
    avg_kernel_s = kms * 1e-3 / kcalls
    cpu = None
    if not args.no_cpu_baseline:
        O = ge.load_oracle()
        orc = O.Oracle(scene, w, h, seed=1)                              # the reference's default intersector (octree, 70 per leaf)
        t = time.perf_counter(); calls = 0; prim = 0
        budget, cap = (12.0, 400) if full else (4.0, 30)
        while time.perf_counter() - t < budget and calls < cap:
            prim += orc.trace_frame_additive(); orc.get_tonemapped_pixels(); calls += 1
        dt = time.perf_counter() - t
        model, phys, usable = cpu_model()
        cpu = {"value": round(prim / dt, 1), "unit": "primary rays/s", "cores": 1, "kind": "port",
               "sample": "%d calls of oracle_trace_frame_additive + oracle_get_tonemapped on 1 thread (%.1f s): the reference's loop is serial "
                         "(mod.rs:80-117); octree oracle at 70 triangles per leaf" % (calls, dt),
               "ms_per_step": round(dt / calls * 1e3, 2), "cpu_model": model, "physical_cores": phys, "build_flags": oracle_build_flags(ge)}
    if not full:
        return {"what": "the loop the reference binary runs (raytracer/src/main.rs:197-207) at its defaults: thai2 1024x768; one step = trace_frame_additive() "
                        "(50 rows x 1 sample = 51 200 primary rays, ONE launch of fused_pass_kernel) + get_tonemapped_pixels() (whole frame as u32 in host memory)",
                "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4), "primary_rays_per_s": round(primary / elapsed, 1),
                "total_mrays_per_s": round(rays / kcalls / (elapsed / steps) / 1e6, 2), "fps": round(steps / elapsed, 1),
                "ms_trace_frame_additive_only": round(trace_only * 1e3, 4), "ms_get_tonemapped_pixels_clean": round(copy_only * 1e3, 4),
                "fused_kernel_ms": round(avg_kernel_s * 1e3, 4), "rays_per_call": rays / kcalls, "cpu_baseline": cpu}
    # the multi-launch wavefront rounds on the same calls, for the A/B
    os.environ["MI355RT_NO_FUSED"] = "1"
    for _ in range(20):
        step()
    t3 = time.perf_counter()
    for _ in range(200):
        step()
    wavefront_ms = (time.perf_counter() - t3) / 200 * 1e3
    del os.environ["MI355RT_NO_FUSED"]
    # instrumented call: steps per ray for the logical byte figure
    rt.set_flags(pkg.FLAG_COUNT_STEPS)
    rt.trace_frame_additive(); c = rt.last_counts()
    rt.set_flags(0)
    r1 = max(1, c.primary + c.bounce + c.shadow)
    acc = rt.accel_stats()
    node_bytes = acc["node_bytes"] / max(acc["nodes"], 1)
    bytes_per_ray = node_bytes * c.nodes_visited / r1 + 48.0 * c.tris_tested / r1 + 96.0
    logical = rays / kcalls * bytes_per_ray / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    roof = {"bound": "latency", "kernel": "fused_pass_kernel", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(logical, 1), "frac": round(logical / HBM_PEAK_GBS, 4),
            "avg_launch_ms": round(avg_kernel_s * 1e3, 4), "rays_per_call": rays / kcalls, "bytes_per_ray": round(bytes_per_ray, 1),
            "note": "one launch of 1 600 waves (51 200 samples in 32-sample chunks, each wave takes its chunk through all 4 trace and 3 shade phases): "
                    "the launch is a chain of dependent cache-latency-bound phases on a chip that is 1/5 full; achieved is the LOGICAL SURVEY 8d byte rate",
            "traffic": None, "pmc": {"note": pmc_note}}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        f, nf = pmc["FETCH_SIZE"]; wv, nw = pmc["WRITE_SIZE"]
        roof["traffic"] = round((2.0 * f / max(nf, 1) + wv / max(nw, 1)) * 1024.0, 1)
    if "SQ_INSTS_VALU" in pmc:
        roof["pmc"]["valu_winst_per_launch"] = round(pmc["SQ_INSTS_VALU"][0] / max(pmc["SQ_INSTS_VALU"][1], 1), 1)
    out = {
        "metric": "primary rays/s of the reference binary's loop: trace_frame_additive() + get_tonemapped_pixels(), thai2 1024x768 (main.rs:13-15,197-207)",
        "value": round(primary / elapsed, 1), "unit": "primary rays/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s 1024x768, 50 rows x 1 sample per call (51 200 primary rays), whole-frame u32 read-out into host memory per call" % args.scene,
                   "scene": args.scene, "width": w, "height": h, "seed": 1},
        "fps": round(steps / elapsed, 1),
        "ms_trace_frame_additive_only": round(trace_only * 1e3, 4), "ms_get_tonemapped_pixels_clean": round(copy_only * 1e3, 4),
        "ms_per_step_wavefront_rounds": round(wavefront_ms, 4),
        "total_mrays_per_s": round(rays / kcalls / (elapsed / steps) / 1e6, 2),
        "hbm_allocated_bytes": rt.hbm_allocated_bytes(),
        "roofline": roof,
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def cpu_baseline(ge, scene, width, height, fix_row_index=False):
    """The oracle (oracle/oracle.c) on this box's host cores, bounded sample of the same workload.  The oracle's rows are
    handed out dynamically to std threads; more threads than physical cores does not always help (SMT siblings share a
    core), so three thread counts are timed for ~5 s each and the BEST is the baseline (all of them are in the line)."""
    O = ge.load_oracle()
    model, phys, usable = cpu_model()
    orc = O.Oracle(scene, width, height, seed=1, flags=O.FLAG_FIX_ROW_INDEX if fix_row_index else 0)
    tried = {}
    best = None
    for nthreads in sorted({max(1, min(64, usable)), max(1, min(phys or usable, usable)), usable}):
        t0 = time.perf_counter()
        tot = dict(primary=0, bounce=0, shadow=0)
        frames = 0
        while True:
            c = orc.render(1, nthreads=nthreads)                          # whole frames at 1 spp
            frames += 1
            for k in tot:
                tot[k] += c[k]
            if time.perf_counter() - t0 >= 5.0 or frames >= 64:
                break
        dt = time.perf_counter() - t0
        rate = (tot["primary"] + tot["bounce"] + tot["shadow"]) / dt / 1e6
        tried[str(nthreads)] = round(rate, 4)
        if best is None or rate > best[0]:
            best = (rate, nthreads, frames, tot["primary"], dt)
    rate, ncores, frames, nprimary, dt = best
    # single-thread rate on a smaller sample (the reference's real threading model, mod.rs:80-117)
    r0 = height // 2 - 32
    t1 = time.perf_counter()
    c1 = orc.render(1, nthreads=1, rows=(r0, r0 + 64))
    dt1 = time.perf_counter() - t1
    rays1 = c1["primary"] + c1["bounce"] + c1["shadow"]
    return {"value": round(rate, 4), "unit": "Mrays/s", "cores": ncores, "kind": "port",
            "sample": "%d spp of the same %dx%d frame (%d primary samples, %.1f s), octree oracle at 70 tris/leaf (the reference's default intersector), "
                      "%d threads = the best of the thread counts tried; single_thread_value = 64 rows x 1 spp on 1 thread" % (frames, width, height, nprimary, dt, ncores),
            "mrays_per_s_by_threads": tried,
            "cpu_model": model, "physical_cores": phys, "logical_cpus_usable": usable, "build_flags": oracle_build_flags(ge),
            "build_note": "x86-64-v3 (AVX2/FMA-capable ISA level, contraction off) rather than -march=native: liboracle.so is built in the build container and travels to "
                          "the GPU box, whose CPU differs; the oracle is scalar f32 code with fp-contract off, so the ISA level barely matters",
            "primary_mrays_per_s": round(nprimary / dt / 1e6, 4),
            "single_thread_value": round(rays1 / dt1 / 1e6, 4)}


if __name__ == "__main__":
    main()
