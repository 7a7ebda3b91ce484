"""Multi-GPU plumbing for the batch decode path.

The path shards by independent images: image i of a batch goes to exactly one GPU,
results are concatenated by index, and there is NO data-path collective (no RCCL traffic over
xGMI).  One process per GPU (torch.distributed launcher); the only cross-rank operations are the
benchmark's control plane: a barrier, MAX/SUM reductions and an all-gather of a few scalars.  bench.py
runs them over "gloo" by default (round 3: nothing on the data path needs RCCL, so its bring-up is
not part of the run) and over "nccl" (= RCCL on ROCm) with --control-plane nccl; ControlPlane() without
an explicit backend keeps the old rule (nccl where a GPU is visible, gloo otherwise).
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """Contiguous slice [lo, hi) of n_units owned by `rank`: sizes differ by at most one,
    the union over ranks is exact and ordered (image i -> rank i // ceil-ish)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def owner_of(unit, n_units, world):
    """Inverse of shard_range."""
    base, rem = divmod(n_units, world)
    edge = rem * (base + 1)
    if unit < edge:
        return unit // (base + 1)
    return rem + (unit - edge) // max(base, 1)


class ControlPlane:
    """barrier + scalar reductions across the ranks of one node; a no-op for world size 1."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.backend = None
        self.device = torch.device("cpu")
        self._dist = self.world > 1 or os.environ.get("MIJ_FORCE_DIST") == "1"  # the knob rehearses the RCCL path on one GPU
        if self._dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.backend = backend

    def barrier(self):
        if self._dist:
            if self.backend == "nccl":
                dist.barrier(device_ids=[self.local_rank])
            else:
                dist.barrier()

    def _reduce(self, value, op):
        if not self._dist:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value):
        return self._reduce(value, dist.ReduceOp.MAX)

    def sum(self, value):
        return self._reduce(value, dist.ReduceOp.SUM)

    def gather_floats(self, values):
        """Every rank's list of floats, in rank order (control plane only: a few scalars per rank)."""
        if not self._dist:
            return [[float(v) for v in values]]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [[float(x) for x in o.tolist()] for o in out]

    def close(self):
        if self._dist and dist.is_initialized():
            dist.destroy_process_group()
