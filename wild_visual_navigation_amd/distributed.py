"""Multi-GPU plumbing: one process per GPU, frames sharded by rank, RCCL (torch.distributed backend
"nccl" on ROCm) for the only exchange the path has -- the two tiny all-reduces of the MLP step
(traversability_estimator/trainer.py).  Feature extraction / segmentation / pooling need no
communication (frames are independent)."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def _pg_kwargs(backend: str) -> dict:
    """RCCL's kernels on a HIGH-priority stream: the backbone kernels are persistent (one workgroup per CU holding the whole register
    file), so a small kernel on a normal-priority stream is dispatched behind them like any other work, while the hardware
    queue of a high-priority stream is served first at every workgroup boundary -- the same reason the tail of a step runs on
    a high-priority stream (bench.py: TwoStreamPipeline).  c10d's default is a normal-priority stream from its pool."""
    if backend != "nccl":
        return {}
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        return {"pg_options": opts}
    except Exception:  # noqa: BLE001  (a build without the option: the default stream)
        return {}


_FORCE_COLLECTIVES = False   # a process group of ONE rank whose collectives still run (RCCL's kernels on one GPU): bench.py --force-collectives


def init_from_env(backend: str = None, force_collectives: bool = False) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process -- unless ``force_collectives``: then a one-rank process
    group is created and every all-reduce of the path is issued for real (with backend "nccl" each one launches RCCL's reduction
    kernel on this GPU: what a collective costs NEXT TO the persistent backbone kernels can be measured on a one-GPU box)."""
    global _FORCE_COLLECTIVES
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if force_collectives and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=0, world_size=1, **_pg_kwargs(backend))
        _FORCE_COLLECTIVES = True
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **_pg_kwargs(backend))
    return rank, world, local


def is_parallel() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks (no-op for a single process).  Summation order inside RCCL/gloo is
    fixed for a given world size, so every rank receives bit-identical results."""
    if is_parallel():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_ranks_ready(ready: bool, device=None) -> bool:
    """MIN over ranks of a local flag (True for a single process): the collective decision whether a step that contains
    collectives runs at all."""
    if not is_parallel():
        return bool(ready)
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([1 if ready else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def barrier():
    if is_parallel():
        dist.barrier()


def replicas_identical(*tensors) -> bool:
    """Whether the bytes of ``tensors`` (parameters, optimiser moments, losses) are the same on every rank: MIN and MAX over the ranks
    of their int32 image agree everywhere.  The design's claim (trainer.py: every rank applies the same update to the same
    parameters) as a check -- bench.py --gpus N asserts it after the timed steps.  True for a single process."""
    if not is_parallel():
        return True
    ok = True
    for t in tensors:
        t = t.detach().contiguous().reshape(-1)
        img = t.view(torch.int32) if t.element_size() == 4 else t.view(torch.uint8).to(torch.int32)
        if dist.get_backend() != "nccl":
            img = img.cpu()
        lo, hi = img.clone(), img.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok = ok and bool(torch.equal(lo, hi))
    return ok


def max_over_ranks(value: float, device=None) -> float:
    if not is_parallel():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
