"""Multi-GPU plumbing shared by bench.py and the hosts: one process per GPU, replicated tables, tiles dealt
round-robin, ONE start-up collective (table broadcast), none in steady state.  Backend "nccl" is RCCL over
xGMI on the GPU box; the same code runs on "gloo" in the CPU tests (tests/test_dist_cpu.py)."""
import os

import torch
import torch.distributed as td


# BSGS_DIST_FORCE=1: a single process still creates the process group and goes through every collective (a one-rank RCCL communicator): the
# multi-rank code path of bench.py -- RCCL next to the engine in one process, broadcasts into engine-owned memory -- inside a one-GPU lease
FORCE = os.environ.get("BSGS_DIST_FORCE") == "1"


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _alone():
    return not (td.is_available() and td.is_initialized()) or (td.get_world_size() == 1 and not FORCE)


def init(backend, device=None):
    rank, local_rank, world = env_world()
    if world == 1 and FORCE and not td.is_initialized():
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if (world > 1 or FORCE) and not td.is_initialized():
        if backend == "nccl":
            td.init_process_group("nccl", device_id=device)
        else:
            td.init_process_group(backend)
    return rank, local_rank, world


def deal_tiles(items, rank, world):
    """the dispenser sequence (GetJob, 1_9_7File.pb:2077-2092) dealt statically: rank r takes r, r+N, ..."""
    return items[rank::world]


def broadcast_table(img, src=0):
    """start-up broadcast of the htGPU image (a torch tensor on the rank's device); returns seconds spent.  RCCL moves device memory
    directly; on gloo (CPU tests; bench.py --same-device: N ranks on ONE GPU) device tensors are staged through host memory in pieces."""
    import time
    if _alone():
        return 0.0
    if img.is_cuda:
        torch.cuda.synchronize()
    t0 = time.time()
    flat = img.view(-1)
    staged = img.is_cuda and td.get_backend() == "gloo"
    step = (1 << 26) if staged else (1 << 30)        # elements per collective: keeps every call's count far below 2^31
    for s in range(0, flat.numel(), step):
        piece = flat[s:s + step]
        if staged:
            host = piece.cpu() if td.get_rank() == src else torch.empty(piece.shape, dtype=piece.dtype)
            td.broadcast(host, src=src)
            if td.get_rank() != src:
                piece.copy_(host)
        else:
            td.broadcast(piece, src=src)
    if img.is_cuda:
        torch.cuda.synchronize()
    return time.time() - t0


def allgather_slices(full, slice_bytes):
    """in-place all-gather of a uint8 tensor made of world equal slices (this rank's slice, number rank, already there): the line slices of the "1/N each +
    all-gather" start-up.  RCCL: one ncclAllGather (viewed as int64: the element count of a 16 GiB slice stays below 2^31); gloo (CPU tests, --same-device):
    one staged broadcast per slice.  Returns seconds spent."""
    import time
    if _alone():
        return 0.0
    world, rank = td.get_world_size(), td.get_rank()
    assert full.numel() == world * slice_bytes and slice_bytes % 8 == 0
    if full.is_cuda and td.get_backend() != "gloo":
        torch.cuda.synchronize()
        t0 = time.time()
        wide = full.view(torch.int64)
        n = slice_bytes // 8
        td.all_gather_into_tensor(wide, wide[rank * n:(rank + 1) * n])
        torch.cuda.synchronize()
        return time.time() - t0
    t = 0.0
    for r in range(world):
        t += broadcast_table(full[r * slice_bytes:(r + 1) * slice_bytes], src=r)
    return t


class DeviceMemory:
    """a raw device allocation of the engine (pointer, bytes) as something torch can wrap without copying
    (torch.as_tensor(DeviceMemory(...), device=...) -> uint8 tensor over the same memory): the receive buffers of a broadcast table"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def wrap_device_memory(ptr, nbytes, device):
    return torch.as_tensor(DeviceMemory(ptr, nbytes), device=device)


def gather_objects(obj):
    """every rank's picklable object, in rank order, on every rank"""
    if _alone():
        return [obj]
    out = [None] * td.get_world_size()
    td.all_gather_object(out, obj)
    return out


def all_equal(obj):
    """(does every rank hold an equal object?, every rank's object in rank order) -- replica verification: table checksums, the hit list of a
    launch every rank ran (bench.py; the C++ host compares its engines the same way after bsgs_broadcast_tables)"""
    objs = gather_objects(obj)
    return all(o == objs[0] for o in objs), objs


XGMI_LINK_GBPS = 153.0        # one xGMI link of an MI355X (7 per GPU): what a one-to-all broadcast out of rank 0 can use per destination


def barrier(cuda=True):
    if not _alone():
        td.barrier()
    if cuda and torch.cuda.is_available():
        torch.cuda.synchronize()


def reduce_max(values, device="cpu"):
    """max over ranks of a list of floats (wall time, kernel time)"""
    if _alone():
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return [float(x) for x in t]


def reduce_sum_int(value, device="cpu"):
    if _alone():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return int(t[0])
