"""One-process-per-GPU data parallelism for the hot path.

Every op on the path is per image (RoIAlign batch index = roi / R, ProposalTarget and NMS loop over
images, soft-NMS is per (image, class)), so images -- and at test time (image, class) problems --
shard across ranks with NO data-path collective.  The only exchange of a training step is the
gradient all-reduce, which the reference does with KVStore 'nccl' (detection_train.py:42-43,
gradients pre-scaled by 1/num_device :266).  Here: torch.distributed, backend "nccl" (= RCCL over
xGMI on MI355X) or "gloo" (CPU tests), gradients coalesced into a few large flat buckets (xGMI is
point to point: ring all-reduce is per-link bound, so few large messages beat many small ones) and
reduced on a side stream, each bucket as soon as the backward has produced its last gradient.
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
    """Gradient all-reduce (sum, then 1/world as detection_train.py:266) in a few large flat buckets,
    overlapped with the backward pass.

    * every bucket is ONE persistent flat tensor; each parameter's .grad is a view into it, so
      nothing is concatenated or copied per step and autograd accumulates straight into the bucket;
    * a post-accumulate hook on every parameter counts the bucket down; when the last gradient of a
      bucket has been written, its all-reduce is launched at once (async, on a side stream ordered
      after the producing stream) while autograd keeps computing the earlier layers -- buckets are
      filled in reverse parameter order, the order the backward produces them;
    * finish() (call it after loss.backward()) joins the side stream and re-arms the counters.
    reduce() is the manual form for gradients that were written without autograd (it launches every
    bucket, then finish()).  A parameter that receives no gradient in a step takes part with zeros
    (its slot of the bucket), like an unused parameter under DistributedDataParallel.

    bucket_mb: target bucket size.  xGMI is point to point (7 links x ~153 GB/s per GPU): a ring
    all-reduce of the ~165 MB of R50-FPN fp32 gradients is bandwidth bound only for messages of
    tens of MB, so the default is few, large buckets.
    """

    def __init__(self, params, bucket_mb=64.0, average=True, hooks=True):
        self.params = [p for p in params]
        self.average = average
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        limit = int(bucket_mb * 1e6)
        groups, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):  # the backward produces the last layers' gradients first
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > limit or cur[0].dtype != p.dtype or cur[0].device != p.device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        self.buckets = []
        self._bucket_of = {}
        for g in groups:
            flat = torch.zeros(sum(p.numel() for p in g), dtype=g[0].dtype, device=g[0].device)
            off = 0
            for p in g:
                n = p.numel()
                view = flat[off:off + n].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                off += n
                self._bucket_of[id(p)] = len(self.buckets)
            # only parameters that receive gradients count a bucket down (a frozen BN / stem
            # parameter has no hook: with it in the count the bucket would never launch early)
            live = sum(1 for p in g if p.requires_grad)
            self.buckets.append({"flat": flat, "params": g, "pending": live, "live": live, "handle": None,
                                 "seen": set()})
        cuda = bool(self.params) and self.params[0].is_cuda
        self.stream = torch.cuda.Stream() if cuda else None
        self._hooks = []
        if hooks and self.world > 1:
            for p in self.params:
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def zero_grad(self):
        """Zero every bucket in place (keeps the .grad views)."""
        for b in self.buckets:
            b["flat"].zero_()

    def _launch(self, b, producer=None):
        """all-reduce bucket b on the side stream, ordered after `producer` (the stream the
        gradients were written on: the CURRENT stream at the time of the call unless given)."""
        if self.stream is not None:
            self.stream.wait_stream(producer if producer is not None else torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        b = self.buckets[self._bucket_of[id(p)]]
        if p.grad is not None and p.grad.data_ptr() != b["flat"].data_ptr() + self._offset(b, p):
            # autograd replaced the view (first accumulation into a None grad): fold it back
            self._view(b, p).copy_(p.grad)
            p.grad = self._view(b, p)
        b["seen"].add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _offset(self, b, p):
        off = 0
        for q in b["params"]:
            if q is p:
                return off * p.element_size()
            off += q.numel()
        raise KeyError("parameter not in bucket")

    def _view(self, b, p):
        off = self._offset(b, p) // p.element_size()
        return b["flat"][off:off + p.numel()].view_as(p)

    def finish(self):
        """Wait for every launched bucket, scale by 1/world, order the result before later work on
        the current stream, re-arm the per-bucket counters."""
        if self.world == 1:
            return
        # the compute stream, captured BEFORE entering the side-stream context: inside it the
        # "current stream" is the side stream itself and waiting on it would order nothing
        producer = torch.cuda.current_stream() if self.stream is not None else None
        for b in self.buckets:
            if b["handle"] is None:   # a bucket some gradient never arrived for
                if self._hooks:
                    # a parameter that got no gradient this step takes part with zeros, not with the
                    # averaged value its slot still holds from the previous step
                    for p in b["params"]:
                        if p.requires_grad and id(p) not in b["seen"]:
                            self._view(b, p).zero_()
                            if p.grad is None:
                                p.grad = self._view(b, p)
                self._launch(b, producer)
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            for b in self.buckets:
                b["handle"].wait()
                if self.average:
                    b["flat"].div_(self.world)
                b["handle"] = None
                b["pending"] = b["live"]
                b["seen"] = set()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def reduce(self):
        """All-reduce gradients that were written into .grad without autograd."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b["handle"] is None:
                self._launch(b)
        self.finish()


class OverlappedAllReduce:
    """One persistent flat buffer all-reduced per step on a side stream: start() after the step's
    producer work has been enqueued, finish() before the consumer (the optimizer).  bench.py uses it
    to put the reference's only inter-GPU exchange (165 MB of fp32 gradients for R50-FPN) next to
    the RoIAlign step."""

    def __init__(self, nbytes, device=None, average=True):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.buf = torch.ones(max(1, int(nbytes) // 4), dtype=torch.float32, device=device)
        self.stream = torch.cuda.Stream() if self.buf.is_cuda else None
        self.average = average
        self.handle = None

    def start(self):
        if self.world == 1:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            self.handle = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self.world == 1 or self.handle is None:
            return
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            self.handle.wait()
            if self.average:
                self.buf.div_(self.world)
        self.handle = None
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
