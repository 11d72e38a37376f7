"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).

Self-play needs no exchange: game g of the job runs on rank g % world with its own tree, RNG
stream and board. The only collectives are a parameter broadcast at start and ONE
all-reduce of the flattened gradient per training mini-batch: 1.2-3.1 M fp32 values (4.9-12.4 MB),
a latency-bound message on 7 x 153 GB/s xGMI links, so a single bucket and no overlap machinery.
"""
import os

import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; no-op for a single process."""
    import torch.distributed as dist
    n = int(os.environ.get("WORLD_SIZE", "1"))
    if n <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world()


def shard_games(n_games, rank, world_size):
    """Episode indices of this rank: g % world == rank (embarrassingly parallel self-play)."""
    return list(range(rank, n_games, world_size))


def broadcast_parameters(module, src=0):
    """Identical weights and BN buffers on every rank (one flat buffer, one broadcast)."""
    import torch.distributed as dist
    rank, n = world()
    if n == 1:
        return
    tensors = [t for t in module.state_dict().values() if t.is_floating_point()]
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            k = t.numel()
            t.copy_(flat[off:off + k].view_as(t))
            off += k


def allreduce_gradients(module):
    """grad <- mean over ranks, through ONE flattened fp32 buffer (ncclAllReduce(sum) / world)."""
    import torch.distributed as dist
    rank, n = world()
    if n == 1:
        return 0
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(n)
    off = 0
    for g in grads:
        k = g.numel()
        g.copy_(flat[off:off + k].view_as(g))
        off += k
    return flat.numel()
