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


def _collectives_on():
    """Collectives run whenever a process group exists -- also a group of ONE rank (the RCCL smoke test on a 1-GPU
    box, tests/test_gpu_multirank.py): the all-reduce of a single rank is the identity, but it loads librccl and
    runs on the device buffers exactly as at 8 ranks. Without a process group every function below is a no-op."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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
    if not _collectives_on():
        return
    tensors = [t for t in module.state_dict().values() if t.is_floating_point()]
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    buf = _staged(flat)
    dist.broadcast(buf, src=src)
    if buf is not flat:
        flat.copy_(buf)
    off = 0
    with torch.no_grad():
        for t in tensors:
            k = t.numel()
            t.copy_(flat[off:off + k].view_as(t))
            off += k


def _staged(flat):
    """The buffer a collective runs on: gloo works on host memory (two ranks may share one GPU in
    the tests), nccl (= RCCL) on the device buffer itself."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and flat.is_cuda:
        return flat.cpu()
    return flat


def agree(value, op="max", device=None):
    """One integer every rank ends up with: max / min / sum of the ranks' local values. Used wherever a
    later loop issues collectives, so that every rank runs the same number of them."""
    import torch.distributed as dist
    if not _collectives_on():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device or "cpu")
    if dist.get_backend() == "nccl" and not t.is_cuda:
        t = t.cuda()
    dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
    return int(t.item())


def agree_sums(values, device=None):
    """Element-wise sum over the ranks of a list of integers (ONE all-reduce); the list itself without a process group.
    main.train uses it once per pass to learn, for every mini-batch of the pass, how many samples ALL ranks will put
    behind it -- so the per-mini-batch gradient all-reduce needs no host read-back to find its divisor."""
    import torch.distributed as dist
    vals = [int(v) for v in values]
    if not _collectives_on() or not vals:
        return vals
    # (host tensor under gloo -- two test ranks may share one GPU --, device tensor under nccl = RCCL)
    t = torch.tensor(vals, dtype=torch.int64, device="cpu")
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None and torch.device(device).type == "cuda" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def allreduce_gradients(module, contributes=True, weight=1.0, total=None):
    """grad <- weighted mean over the CONTRIBUTING ranks, through ONE flattened fp32 buffer (ncclAllReduce(sum)).

    Every rank calls this once per mini-batch, whether or not it had a batch of its own: a rank without
    one (empty replay shard, fewer samples than the agreed step count) passes contributes=False and adds
    zeros. `weight` is the number of samples behind this rank's gradient (main.train_batch passes len(batch)): the
    local gradient -- a mean over the local batch -- is scaled by it, so the result is the gradient of the mean loss
    over the UNION of the ranks' batches and a rank whose shard ran short (a partial batch) counts for what it holds,
    not for a full share.

    The divisor (the weights' sum over the ranks). `total` given: the caller already knows it (main.train agrees on the
    whole pass's batch sizes up front with ONE agree_sums), the buffer is divided by that number and nothing is read
    back from the device -- the RCCL all-reduce, the division and the copy into .grad are all enqueued and the host runs
    ahead. `total` None: the weights' sum travels as one extra element of the same message and is read back (a host
    synchronisation per call: the convenience form, for callers that do not know their peers' batch sizes).
    Returns (elements reduced, summed weight: the number of contributing ranks when every weight is 1);
    without a process group it is a no-op."""
    import torch.distributed as dist
    params = [p for p in module.parameters() if p.requires_grad]
    weight = float(weight) if contributes else 0.0
    if not _collectives_on():
        return sum(p.numel() for p in params), (int(weight) if weight == int(weight) else weight)
    dev = params[0].device
    if contributes:
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
    else:
        parts = [torch.zeros(p.numel(), dtype=p.dtype, device=dev) for p in params]
    if total is None:
        parts.append(torch.full((1,), 1.0, dtype=parts[0].dtype, device=dev))
    flat = torch.cat(parts)
    if weight != 1.0:
        flat.mul_(weight)                                 # (the trailing 1 becomes the weight)
    buf = _staged(flat)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if buf is not flat:
        flat.copy_(buf)
    if total is None:
        total = float(flat[-1].item())
        n_el = flat.numel() - 1
    else:
        total = float(total)
        n_el = flat.numel()
    count = int(round(total)) if abs(total - round(total)) < 1e-3 else total
    if total > 0:
        flat.div_(total)
    off = 0
    for p in params:
        k = p.numel()
        g = flat[off:off + k].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += k
    return n_el, count


def average_buffers(module, contributes=True):
    """BatchNorm running statistics after a data-parallel training pass: every rank normalised its own
    mini-batches, so running_mean / running_var differ; they are averaged over the ranks that trained
    (one flattened all-reduce) and num_batches_tracked becomes the maximum, which leaves the state_dict
    bit-identical on every rank (the gradients already were)."""
    import torch.distributed as dist
    if not _collectives_on():
        return
    fl = [b for b in module.buffers() if b.is_floating_point()]
    it = [b for b in module.buffers() if not b.is_floating_point()]
    if fl:
        dev = fl[0].device
        parts = [(b.detach() if contributes else torch.zeros_like(b)).reshape(-1).float() for b in fl]
        parts.append(torch.full((1,), 1.0 if contributes else 0.0, device=dev))
        flat = torch.cat(parts)
        buf = _staged(flat)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if buf is not flat:
            flat.copy_(buf)
        # divide on the device by the count that travelled in the message; with no contributor at all every rank
        # keeps what it has (the select below), and nothing is read back to the host
        cnt = flat[-1]
        avg = flat / cnt.clamp_min(1.0)
        off = 0
        with torch.no_grad():
            for b in fl:
                k = b.numel()
                b.copy_(torch.where(cnt > 0, avg[off:off + k].view_as(b).to(b.dtype), b))
                off += k
    if it:
        flat = torch.cat([b.detach().reshape(-1).to(torch.int64) for b in it])
        buf = _staged(flat)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
        if buf is not flat:
            flat.copy_(buf)
        off = 0
        with torch.no_grad():
            for b in it:
                k = b.numel()
                b.copy_(flat[off:off + k].view_as(b))
                off += k
