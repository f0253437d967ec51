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


class ScanComm:
    """The cross-rig merge through the C-ABI (`jn_comm_*`, `jn_scan_allreduce`): one RCCL communicator per rank,
    one MIN all-reduce of a packed [batch][bins+4] buffer per batch, issued by the library on its own stream.

    `exchange_id(id_bytes_or_None) -> id_bytes` is the side channel that hands rank 0's ncclUniqueId to the other
    ranks (bench.py passes a torch.distributed broadcast; a ROS host would use a parameter or a file)."""

    def __init__(self, rank, world, device, exchange_id):
        import ctypes as C
        from . import _lib
        self._L = _lib.load()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self._L.jn_comm_unique_id(ident), "jn_comm_unique_id")
        raw = exchange_id(bytes(ident) if rank == 0 else None)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        # ncclCommInitRank is a collective with no deadline of its own: a peer that never calls it would hang this rank for ever.  A watchdog
        # thread ends the process (exit code 3, never an exec) when JN_COMM_INIT_TIMEOUT_S (default 120, 0 = none) passes first.
        import sys
        import threading
        done = threading.Event()
        limit = float(os.environ.get("JN_COMM_INIT_TIMEOUT_S", "120"))

        def guard():
            if not done.wait(limit):
                sys.stderr.write("jackal_navigation_amd: jn_comm_create(rank %d of %d, device %d) not complete after %.0f s "
                                 "(a peer rank is missing): exiting 3\n" % (rank, world, device, limit))
                sys.stderr.flush()
                os._exit(3)
        if world > 1 and limit > 0:
            threading.Thread(target=guard, daemon=True).start()
        try:
            _lib.check(self._L.jn_comm_create(ident, rank, world, device, C.byref(h)), "jn_comm_create")
        finally:
            done.set()
        self._h = h

    def info(self):
        """(rank, world, device) as RCCL reports them for this communicator."""
        import ctypes as C
        from . import _lib
        r, w, d = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._L.jn_comm_info(self._h, C.byref(r), C.byref(w), C.byref(d)), "jn_comm_info")
        return r.value, w.value, d.value

    def merge(self, n, bins, dBins, dMeta):
        """In place on device pointers; returns when every rank's buffers hold the merged scans."""
        from . import _lib
        _lib.check(self._L.jn_scan_allreduce(self._h, n, bins, dBins, dMeta), "jn_scan_allreduce")

    def close(self):
        if getattr(self, "_h", None):
            self._L.jn_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host cores per rank ---------------------------------------------------------------------------------------------
# What limits 1 -> 8 GPU scaling of this path is the shared host: every rank runs a Delaunay pool and
# keeps pinned buffers.  Each rank therefore gets its own whole physical cores on the NUMA node its GPU hangs off.

def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11'"""
    cpus = sorted(cpus)
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(parts)


def plan_affinity(allowed, gpu_local_cpus, siblings, rank):
    """CPUs for `rank` of len(gpu_local_cpus) ranks on one node.

    allowed         CPUs this job may use (sched_getaffinity)
    gpu_local_cpus  per rank: the CPUs local to that rank's GPU (its NUMA node), or None if unknown
    siblings        cpu -> tuple of hardware threads sharing its core (may be {})
    Ranks whose GPUs share a NUMA node split that node's physical cores evenly, in rank order; a rank whose node is
    unknown (or has no allowed CPU) shares what is left of `allowed` evenly with the other such ranks."""
    allowed = set(allowed)
    world = len(gpu_local_cpus)
    keys = []
    for cpus in gpu_local_cpus:
        k = frozenset(cpus) & allowed if cpus else frozenset()
        keys.append(k)
    claimed = set().union(*[k for k in keys if k]) if any(keys) else set()
    rest = frozenset(allowed - claimed) or frozenset(allowed)
    keys = [k if k else rest for k in keys]
    mine = keys[rank]
    group = [r for r in range(world) if keys[r] == mine]
    cores, seen = [], set()
    for c in sorted(mine):
        if c in seen:
            continue
        sib = tuple(sorted(x for x in siblings.get(c, (c,)) if x in mine)) or (c,)
        seen.update(sib)
        cores.append(sib)
    lo, hi = shard(len(cores), group.index(rank), len(group))
    share = [c for core in cores[lo:hi] for c in core]
    return sorted(share) if share else sorted(mine)


def gpu_local_cpulist(pci_bdf):
    """CPUs of the NUMA node a PCI device hangs off (sysfs), or None."""
    try:
        with open("/sys/bus/pci/devices/%s/local_cpulist" % pci_bdf) as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except OSError:
        return None


def cpu_siblings(cpus):
    out = {}
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                out[c] = tuple(parse_cpulist(f.read()))
        except OSError:
            out[c] = (c,)
    return out


def pin_rank(rank, world, device_of_rank):
    """Restrict this process (and every thread it starts from now on: the library's slot workers and Delaunay pool) to
    this rank's share of the host.  device_of_rank: HIP device ordinal per rank.  Returns a dict describing the plan."""
    import torch
    allowed = sorted(os.sched_getaffinity(0))
    local = []
    bdfs = []
    for dev in device_of_rank:
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            bdf = None
        bdfs.append(bdf)
        local.append(gpu_local_cpulist(bdf) if bdf else None)
    cpus = plan_affinity(allowed, local, cpu_siblings(allowed), rank)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        cpus = allowed
    numa = None
    try:
        with open("/sys/bus/pci/devices/%s/numa_node" % bdfs[rank]) as f:
            numa = int(f.read())
    except Exception:
        pass
    return {"rank": rank, "device": device_of_rank[rank], "pci": bdfs[rank], "numa_node": numa, "cpus": len(cpus),
            "cpulist": format_cpulist(cpus)}
