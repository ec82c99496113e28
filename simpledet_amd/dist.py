"""One-process-per-GPU data parallelism for the hot path.

Every op on the path is per image (RoIAlign batch index = roi / R, ProposalTarget and NMS loop over
images, soft-NMS is per (image, class)), so images -- and at test time (image, class) problems --
shard across ranks with NO data-path collective.  The only exchange of a training step is the
gradient all-reduce, which the reference does with KVStore 'nccl' (detection_train.py:42-43,
gradients pre-scaled by 1/num_device :266).  Here: torch.distributed, backend "nccl" (= RCCL over
xGMI on MI355X) or "gloo" (CPU tests), gradients coalesced into a few large flat buckets (xGMI is
point to point: ring all-reduce is per-link bound, so few large messages beat many small ones) and
reduced on a side stream so the reduction overlaps the rest of the backward.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items independent units (images) for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_round_robin(n_items, rank, world):
    """Round-robin shard of (image, class) soft-NMS problems (their cost is ragged)."""
    return list(range(rank, n_items, world))


class GradBucketReducer:
    """Flat-bucket gradient all-reduce (sum, then scale by 1/world as detection_train.py:266).

    bucket_mb: target bucket size.  With 7 xGMI links x ~153 GB/s per GPU a ring all-reduce of the
    ~165 MB of R50-FPN fp32 gradients is bandwidth bound only for buckets of tens of MB.
    """

    def __init__(self, params, bucket_mb=64.0, average=True):
        self.params = [p for p in params]
        self.average = average
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.buckets = []
        cur, cur_bytes = [], 0
        limit = int(bucket_mb * 1e6)
        for p in self.params:
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > limit or cur[0].dtype != p.dtype):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self.stream = torch.cuda.Stream() if (self.params and self.params[0].is_cuda) else None

    def reduce(self):
        """All-reduce .grad of every parameter in place; returns after the reduction is ordered
        after the current stream's work and before anything issued later on it."""
        if self.world == 1:
            return
        handles = []
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            for b in self.buckets:
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b]
                flat = torch.cat([g.reshape(-1) for g in grads])
                h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
                handles.append((h, flat, b))
            for h, flat, b in handles:
                h.wait()
                if self.average:
                    flat.div_(self.world)
                off = 0
                for p in b:
                    n = p.numel()
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                    p.grad.copy_(flat[off:off + n].view_as(p))
                    off += n
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def gather_ragged(local_list, world=None):
    """all_gather of per-rank variable-length result lists (soft-NMS detections at test time)."""
    if not dist.is_initialized():
        return [local_list]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_list)
    return out
