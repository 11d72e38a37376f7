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


DEFAULT_TIMEOUT_S = 300.0    # collectives of a training run (AO_DIST_TIMEOUT overrides; bench.py asks for 40 s)


def init_from_env(backend=None, timeout_s=None, pin=True):
    """Initialise torch.distributed from torchrun's environment; no-op for a single process.

    timeout_s: the process group's collective timeout (default AO_DIST_TIMEOUT or 300 s). A rank that died or hangs then takes its
    peers down with it instead of leaving them blocked for torch's default 10 minutes (RCCL) / 30 minutes (gloo): under "nccl"
    torch's watchdog aborts the process when a collective exceeds it, under "gloo" the collective raises. pin: bind this process
    (and every thread it starts from here on: the engine's host pool, torch's intra-op threads) to the CPUs of its GPU's NUMA node
    (pin_to_gpu_numa; AO_NO_AFFINITY=1 turns it off)."""
    import datetime
    import torch.distributed as dist
    n = int(os.environ.get("WORLD_SIZE", "1"))
    if n <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if timeout_s is None:
        timeout_s = float(os.environ.get("AO_DIST_TIMEOUT", DEFAULT_TIMEOUT_S))
    timeout = datetime.timedelta(seconds=float(timeout_s))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if pin and torch.cuda.is_available():
        pin_to_gpu_numa(local % max(torch.cuda.device_count(), 1))
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local), timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    return world()


def set_collective_timeout(seconds):
    """Tighten (or relax) the timeout of the default process group's collectives after start-up: rendezvous, RCCL's communicator
    set-up over xGMI and the first page-in of a cold box want minutes, the steady state wants to notice a dead peer in seconds.
    Returns True when the backend took it."""
    import datetime
    import torch.distributed as dist
    if not _collectives_on():
        return False
    try:
        dist.distributed_c10d._set_pg_timeout(datetime.timedelta(seconds=float(seconds)))
        return True
    except Exception:
        return False


# ---- host placement: one process per GPU, each on the NUMA node its GPU hangs off -------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' (the kernel's cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def gpu_numa_cpus(pci_bus_id, sysfs="/sys"):
    """(NUMA node, its CPUs) of the PCI device 'dddd:bb:dd.f', or (None, []) when the platform does not say (numa_node -1 on
    single-node hosts, no sysfs). Looked up under /sys/bus/pci/devices first, then through /sys/class/drm/card*/device."""
    pci_bus_id = pci_bus_id.lower()
    cands = [os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id)]
    drm = os.path.join(sysfs, "class", "drm")
    try:
        for card in sorted(os.listdir(drm)):
            dev = os.path.join(drm, card, "device")
            if card.startswith("card") and "-" not in card and os.path.basename(os.path.realpath(dev)).lower() == pci_bus_id:
                cands.append(dev)
    except OSError:
        pass
    for dev in cands:
        try:
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            continue
        if node < 0:
            return None, []
        try:
            with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
                return node, parse_cpulist(f.read())
        except (OSError, ValueError):
            return node, []
    return None, []


def device_pci_bus_id(local):
    p = torch.cuda.get_device_properties(local)
    return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)


last_affinity = None     # what pin_to_gpu_numa did last: dict(numa_node, cpus, pci) or dict(skipped=reason)


def pin_to_gpu_numa(local, sysfs="/sys", pci_bus_id=None):
    """sched_setaffinity of the calling thread to the CPUs of GPU `local`'s NUMA node (intersected with what the process may use
    already). Threads started afterwards inherit it: call before the first Engine is created (its host pool draws the Dirichlet
    noise of thousands of games per move and stages 10 MB of MT19937 state per move through pinned memory: with eight ranks on a
    two-socket host, half of them would otherwise do that across the socket link). Never fails: returns what it did."""
    global last_affinity
    if os.environ.get("AO_NO_AFFINITY"):
        last_affinity = dict(skipped="AO_NO_AFFINITY")
        return last_affinity
    try:
        pci = pci_bus_id or device_pci_bus_id(local)
        node, cpus = gpu_numa_cpus(pci, sysfs)
        if node is None or not cpus:
            last_affinity = dict(skipped="no NUMA node reported for %s" % pci)
            return last_affinity
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            last_affinity = dict(skipped="NUMA node %d has no CPU this process may use" % node)
            return last_affinity
        os.sched_setaffinity(0, allowed)
        last_affinity = dict(numa_node=node, cpus=len(allowed), pci=pci)
        # the engine's host pool (csrc/host_rng.hpp) budgets hardware threads / LOCAL_WORLD_SIZE: with the ranks bound to their
        # nodes the share is this node's CPUs / the ranks that live on it
        try:
            n_local = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
            if n_local > 1 and "AO_HOST_THREADS" not in os.environ and pci_bus_id is None:
                ndev = max(torch.cuda.device_count(), 1)       # (rank r drives device r % ndev: ranks may share a device in the tests)
                node_of = [gpu_numa_cpus(device_pci_bus_id(d), sysfs)[0] for d in range(ndev)]
                same = sum(1 for r_ in range(n_local) if node_of[r_ % ndev] == node)
                os.environ["AO_HOST_THREADS"] = str(max(1, min(32, len(allowed) // max(same, 1))))
                last_affinity["host_threads"] = int(os.environ["AO_HOST_THREADS"])
        except Exception:
            pass
    except Exception as e:                                # (placement is an optimisation: never a reason to fail a run)
        last_affinity = dict(skipped=repr(e))
    return last_affinity


# ---- a rank that stops making progress takes itself down (and, through the collective timeout, its peers) -------------------
class Watchdog:
    """Armed around the regions that issue collectives (bench.py's timed loop, main's training pass): beat() after every unit of
    work; when no beat arrives for `stall_s` seconds a daemon thread prints what the rank was doing and ends the process with
    exit code 86 -- a hung GPU queue or a peer that never shows up becomes a failed job in about a minute, not a node that sits
    there until somebody looks. AO_WATCHDOG=0 disables it."""

    def __init__(self, stall_s, what=""):
        import threading
        self.stall_s, self.what = float(stall_s), what
        self._last = None
        self._note = what
        self._stop = threading.Event()
        self._thread = None
        self._lock = threading.Lock()

    def beat(self, note=None):
        import time
        with self._lock:
            self._last = time.monotonic()
            if note is not None:
                self._note = note

    def __enter__(self):
        import threading
        if os.environ.get("AO_WATCHDOG", "1") != "0" and self.stall_s > 0:
            self.beat()
            self._stop.clear()
            self._thread = threading.Thread(target=self._run, name="alpha_omok_amd.watchdog", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
            self._thread = None
        return False

    def _run(self):
        import sys
        import time
        while not self._stop.wait(min(1.0, self.stall_s / 4.0)):
            with self._lock:
                idle = time.monotonic() - self._last
                note = self._note
            if idle > self.stall_s:
                r, w = world()
                sys.stderr.write("alpha_omok_amd watchdog: rank %d of %d made no progress for %.0f s (%s) -- exiting with code 86 so that the "
                                 "job fails instead of hanging\n" % (r, w, idle, note))
                sys.stderr.flush()
                os._exit(86)


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
