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
