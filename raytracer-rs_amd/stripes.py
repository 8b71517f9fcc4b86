"""Row-stripe decomposition of a frame over ranks and the one collective of the render path.

The reference iterates rows independently (raytracer/mod.rs:87-115); every primary sample is
independent given (scene, camera, seed, pixel, sample#).  Rows are dealt to ranks in stripes of
`stripe_rows` rows, round-robin, so the uneven hit fraction of the image load-balances; each rank
renders only its stripes (mi355rt_config.stripe_rows / stripe_rank / stripe_world) and the packed
u32 stripes are exchanged with ONE all_gather per frame (RCCL over xGMI on GPUs; gloo in the CPU
tests).  Nothing else crosses ranks."""
import torch


def owned_rows(height, stripe_rows, rank, world):
    """ascending rows of `rank` — must agree with Renderer::init in csrc/renderer.cpp"""
    return [y for y in range(height) if (y // stripe_rows) % world == rank]


def max_owned_rows(height, stripe_rows, world):
    nstripes = (height + stripe_rows - 1) // stripe_rows
    return ((nstripes + world - 1) // world) * stripe_rows


class FrameGather:
    """all_gather of the ranks' packed stripes + placement of the rows into the full frame."""

    def __init__(self, height, width, stripe_rows, world, device):
        self.height, self.width, self.world = height, width, world
        self.max_rows = max_owned_rows(height, stripe_rows, world)
        self.rows = [torch.tensor(owned_rows(height, stripe_rows, r, world), dtype=torch.int64, device=device) for r in range(world)]
        self.gathered = torch.zeros(world * self.max_rows * width, dtype=torch.int32, device=device)
        self.frame = torch.zeros(height * width, dtype=torch.int32, device=device)

    def stripe_buffer(self, device):
        """per-rank send buffer: max_rows * width packed u32 (rows beyond the owned ones are padding)"""
        return torch.zeros(self.max_rows * self.width, dtype=torch.int32, device=device)

    def gather(self, dist, stripe):
        if self.world == 1:
            n = self.rows[0].numel()
            self.frame.view(self.height, self.width)[self.rows[0]] = stripe.view(self.max_rows, self.width)[:n]
            return self.frame
        dist.all_gather_into_tensor(self.gathered, stripe)
        g = self.gathered.view(self.world, self.max_rows, self.width)
        f = self.frame.view(self.height, self.width)
        for r in range(self.world):
            f[self.rows[r]] = g[r, : self.rows[r].numel()]
        return self.frame


class NativeGather:
    """The gather INSIDE libmi355rt.so (include/mi355rt.h, mi355rt_comm_*): grouped ncclSend / ncclRecv of the packed u32
    stripes to rank 0 on the handle's own stream, placement kernel on the root.  torch.distributed is used ONLY to hand
    the 128-byte RCCL id from rank 0 to the other ranks and to agree on the three verdicts below.

    No rank may enter the collective ncclCommInitRank unless every rank can: a rank that fails earlier would leave the
    others blocked inside RCCL's bootstrap.  So the set-up is three agreed steps, each closed by an all_reduce(MIN):
      1. every rank runs the library's LOCAL pre-check (mi355rt_comm_available: librccl.so loadable with all symbols, device
         bindable, no live communicator); rank 0 also draws the unique id.  Any failure: nobody calls comm_init.
      2. all ranks call comm_init (collective).
      3. every rank reports what RCCL itself says about the communicator (ncclCommCount == world)."""

    def __init__(self, pkg, rt, dist, rank, world, device="cuda"):
        self.rt = rt
        self.error = ""
        self.ranks = 0
        ok, err = rt.comm_available()
        idt = torch.zeros(128, dtype=torch.uint8, device=device)
        if ok and rank == 0:
            try:
                idt.copy_(torch.frombuffer(bytearray(pkg.comm_unique_id()), dtype=torch.uint8))
            except Exception as e:      # noqa: BLE001 — agreed on below
                ok, err = False, str(e)
        if not self._agree(dist, ok, device):
            raise RuntimeError("RCCL gather unavailable on some rank" + (": " + err if not ok else ""))
        dist.broadcast(idt, src=0)
        ok = True
        try:
            rt.comm_init(bytes(idt.cpu().numpy().tobytes()))
        except Exception as e:          # noqa: BLE001 — agreed on below, then raised on every rank
            ok, err = False, str(e)
        if not self._agree(dist, ok, device):
            if ok:
                rt.comm_destroy()
            raise RuntimeError("mi355rt_comm_init failed on some rank" + (": " + err if not ok else ""))
        self.ranks = rt.comm_ranks()
        if not self._agree(dist, self.ranks == world, device):
            rt.comm_destroy()
            raise RuntimeError("RCCL reports %d ranks in the communicator, expected %d" % (self.ranks, world))

    @staticmethod
    def _agree(dist, ok, device):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag[0]) == 1

    def gather(self):
        self.rt.comm_gather_frame(0, None)          # queued on the handle's stream; rt.synchronize() waits for it

    def verify_against(self, dist, rank, frame_gather, stripe):
        """One frame through BOTH transports (collective: every rank calls it): the library's RCCL gather must put on rank 0
        exactly the frame torch.distributed's all_gather assembles.  Returns the verdict agreed on by all ranks."""
        import numpy as np
        rt = self.rt
        out = np.zeros(frame_gather.height * frame_gather.width, np.uint32) if rank == 0 else None
        rt.comm_gather_frame(0, out if rank == 0 else None)
        rt.synchronize()
        rt.tonemap_owned_rows_device(stripe.data_ptr(), len(rt.owned_rows()) * frame_gather.width)
        ref = frame_gather.gather(dist, stripe)
        ok = 1
        if rank == 0:
            ok = int(np.array_equal(out, ref.cpu().numpy().view(np.uint32)))
        flag = torch.tensor([ok], dtype=torch.int32, device=stripe.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag[0]))

    def close(self):
        self.rt.synchronize()
        self.rt.comm_destroy()
