"""Multi-GPU plumbing: one process per GPU, rigs (stereo pairs) sharded across ranks.

The path has exactly one exchange step: merging the per-rig obstacle scans into a
robot-level scan, an element-wise MIN over the 90 bins (point_cloud.cpp:264-266 applied across
rigs) plus min/max of the four LaserScan extrema (:255-260).  Everything else is independent per
pair, so there is no other collective.  backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU
tests.
"""
import os


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard(n_items, rank, world):
    """Contiguous block partition of n_items rigs over ranks (first ranks get the remainder)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_scans(bins, meta=None):
    """In-place cross-rank merge of scans: bins [..., nbins] -> element-wise MIN; meta [..., 4] =
    (angle_min, angle_max, range_min, range_max) -> MIN/MAX/MIN/MAX.  Tensors live on the device of
    the backend (GPU for nccl, CPU for gloo).  No-op when not distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bins, meta
    dist.all_reduce(bins, op=dist.ReduceOp.MIN)
    if meta is not None:
        lo = meta[..., 0::2].contiguous()
        hi = meta[..., 1::2].contiguous()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        meta[..., 0::2] = lo
        meta[..., 1::2] = hi
    return bins, meta


class ScanBuffer:
    """Scans of one batch laid out for a single collective: `bins` [batch][nbins] and `meta` [batch][4] are two
    contiguous views of ONE flat float64 tensor, so the cross-rank merge is one MIN all-reduce per batch (the two
    maxima of meta travel negated; negation of a double is exact).  The message is a few tens of KB, i.e. latency
    bound over xGMI: one collective instead of three is what matters."""

    def __init__(self, batch, nbins=90, device="cpu"):
        import torch
        self.batch, self.nbins = batch, nbins
        self.flat = torch.zeros(batch * (nbins + 4), dtype=torch.float64, device=device)
        self.bins = self.flat[:batch * nbins].view(batch, nbins)
        self.meta = self.flat[batch * nbins:].view(batch, 4)

    def merge(self):
        """In place; afterwards every rank holds the robot-level scan.  No-op when not distributed."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self
        self.meta[:, 1::2].neg_()
        dist.all_reduce(self.flat, op=dist.ReduceOp.MIN)
        self.meta[:, 1::2].neg_()
        return self
