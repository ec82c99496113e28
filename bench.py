#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X.

One "step" = one pass of the hot path over one per-GPU batch of synthetic input:
  fused FPN RoIAlign forward + backward, P2-P5, 256 ch, 800x1333, N=2 images, 512 RoIs/img, 7x7 bins
  x 4 samples (BASELINE.json configs[1]).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment re-launches itself as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
(one rank per GPU, RCCL); started by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

Prints ONE JSON line on rank 0.  `value` = images/s over all ranks of THE PATH ITSELF (weak scaling:
every rank processes its own images; the path has no data-path collective and no parameters -- RoIs /
images are independent): the timed loop of K steps contains the hot path only, on every
rank, bracketed by barrier + synchronize, max over ranks.  The reference's one inter-GPU exchange --
the KVStore `nccl` all-reduce of the model's gradients (detection_train.py:42-43,266; 165 MB of
fp32 for R50-FPN) -- is NOT part of this path; for N > 1 it is measured in the same invocation as
separate legs and reported under `grad_allreduce` (alone: bus bandwidth; composite: every step
starts it behind the forward and joins it at the end of the step, the way a data-parallel backward
overlaps it -- a 0.2 ms step cannot hide a 165 MB all-reduce, a real step hides it under ~100 ms of
backbone backward).

roofline: ALGORITHMIC bytes of one forward launch (N*S_F + 16*N*R + 3*N*S_O
= 335.9 MB at N=2) / the forward kernel's average duration measured with HIP events on the launch
stream inside the timed region; peak = 8 TB/s HBM3E.  The backward (one fused launch for all levels,
same algorithmic bytes) is reported next to it.
cpu_baseline: the CPU oracle (oracle/, a restatement of operator_cxx's arithmetic that is pinned
bit for bit to the reference's own compiled operators, tests/test_ref_pins.py) timed on the host
cores on the same workload: all cores and one core.  Reported baseline, not the product.
Verification (not timed): the forward output AND the gradients of the last timed step are compared
with the oracle (forward bit-exact, backward |err| <= 1e-4 elementwise).
Clock state: before the W warm-up steps the same step runs untimed for ~0.1 s (--preheat-ms) so
that short K/W settings measure the steady state a training loop sees instead of the clock ramp.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0
R50_FPN_GRAD_MB = 165.0  # fp32 gradients of faster_r50v1_fpn (reference: KVStore 'nccl' all-reduce)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--images", type=int, default=2, help="images per GPU (reference: 2)")
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="untimed clock pre-heat before the --warmup steps (0 disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU oracle (baseline timing AND the verification of the timed step)")
    ap.add_argument("--cpu-passes", type=int, default=3)
    ap.add_argument("--grad-allreduce", type=float, default=-1.0,
                    help="MB of fp32 gradients all-reduced (RCCL) per step; -1 = 165 when N > 1, 0 = off")
    ap.add_argument("--tuning", action="append", default=[], help="key=value kernel-variant knob (A/B)")
    ap.add_argument("--float-argmax", action="store_true",
                    help="keep the arg-max between forward and backward as two fp32 planes (the "
                         "reference op's outputs) instead of one byte per output")
    ap.add_argument("--no-plan", action="store_true",
                    help="A/B: the forward and the backward each launch their own rois-only pre-pass (4 "
                         "launches per step) instead of one merged pre-pass in the forward (3 launches)")
    ap.add_argument("--calibrate", action="store_true",
                    help="also run the known-size HBM stream copies (measured peak + PMC calibration)")
    ap.add_argument("--no-ops", action="store_true",
                    help="skip the per-operator secondary measurements (bench_ops.py)")
    ap.add_argument("--no-extra", action="store_true", help="skip the un-fused 4-op graph path")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP events (forward / backward split for the roofline) are recorded in every "
                         "n-th step of the timed region: an event record is a marker packet with a release "
                         "fence between two launches (+3-4 us each, measured: tools/event_cost_bench.py), "
                         "three per step inflate the very step they time by ~6 %%; 1 = every step")
    return ap.parse_args()


def kernel_source_sha():
    """sha256 over the sources of the kernels the roofline is about: a PMC profile is only quoted for them"""
    import hashlib
    h = hashlib.sha256()
    for f in ("roi_align_common.h", "roi_align_lists.h", "roi_align_fwd.hip", "roi_align_bwd.hip", "roi_align_prep.hip",
              "common.h", "runtime.hip"):
        h.update(open(os.path.join(ROOT, "simpledet_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def nms_l2_hit_rates():
    """north_star: "rocprof reports ... L2-hit for NMS".  PMC counters cannot be read inside this run; the
    newest committed profile of the secondary ops (profiles/*_ops_pmc_summary.json, tools/prof_ops.sh:
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum over bench_ops.run) is quoted -- only when it was taken with
    the nms.hip / soft_nms.hip / common.h of this tree (per-file sha256 recorded by
    tools/summarize_profile.py).  -> ({kernel: l2_hit_rate}, source) or (None, why not)."""
    import glob
    import hashlib
    import re
    need = {}
    for f in ("nms.hip", "soft_nms.hip", "common.h"):
        need[f] = hashlib.sha256(open(os.path.join(ROOT, "simpledet_amd", "csrc", f), "rb").read()).hexdigest()
    cands = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*_ops_pmc_summary.json"))
             if re.match(r"^r\d+[a-z]?_ops_pmc_summary\.json$", os.path.basename(f))]
    for path in sorted(cands)[::-1]:
        try:
            prof = json.load(open(path))
            sha = prof.get("source_sha256") or {}
            if any(sha.get(f) != h for f, h in need.items()):
                continue
            out = {}
            for name, d in prof["kernels"].items():
                m = re.search(r"sd::(nms_\w+|soft_nms_kernel)", name)
                if m and "l2_hit_rate" in d:
                    out[m.group(1)] = round(d["l2_hit_rate"], 4)
            if out:
                return out, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, "no committed *_ops_pmc_summary.json was taken with the current nms.hip / soft_nms.hip / common.h"


def algorithmic_bytes(n_img, n_roi, channels, shapes, pooled=49):
    s_f = 4 * channels * sum(h * w for h, w in shapes)      # every level read (fwd) / written (bwd)
    s_o = 4 * n_roi * channels * pooled                      # one (R,C,7,7) fp32 tensor per image
    return n_img * s_f + n_img * n_roi * 16 + 3 * n_img * s_o


def self_launch(n):
    """python bench.py --gpus N, no torchrun environment: become the launcher."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def leg_single_rank(step, barrier, args, rank, device):
    """rank 0 alone, the other ranks idle: the N = 1 figure of this box in this invocation (seconds
    for K steps, the same on every rank)"""
    import torch
    import torch.distributed as dist
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    barrier()
    t0 = time.perf_counter()
    if rank == 0:
        for _ in range(args.steps):
            step()
        sync()
    ts = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.broadcast(ts, 0)
    return float(ts.item())


def legs_grad_allreduce(step, reducer, barrier, max_over_ranks, args, world, grad_mb, t_single, t_ops):
    """The reference's data-parallel exchange next to the path, K steps each, every rank:
      alone      the gradient all-reduce by itself          -> its bus bandwidth over xGMI
      composite  hot path + the all-reduce started behind the forward, joined at the end of the step
    t_single / t_ops: seconds of K steps of rank 0 alone / of the timed loop (`value`)."""
    for _ in range(2):
        reducer.start()
        reducer.finish()
    nar = max(3, min(args.steps, 10))
    barrier()
    t0 = time.perf_counter()
    for _ in range(nar):
        reducer.start()
        reducer.finish()
    barrier()
    t_ar, _ = max_over_ranks(time.perf_counter() - t0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(reducer=reducer)
    barrier()
    t_both, _ = max_over_ranks(time.perf_counter() - t0)
    ms_ops, ms_both = t_ops * 1e3 / args.steps, t_both * 1e3 / args.steps
    return {"mb_per_step": grad_mb,
            "alone_ms": t_ar * 1e3 / nar,
            "busbw_GBs": (2.0 * (world - 1) / world) * grad_mb * 1e6 / (t_ar / nar) / 1e9,
            "ops_plus_allreduce_ms_per_step": ms_both,
            "value_ops_plus_allreduce": args.images * world * args.steps / t_both,
            "exposed_ms": max(0.0, ms_both - ms_ops),
            "weak_scaling_eff_with_allreduce": t_single / t_both,
            "what": "NOT in `value`: the reference's KVStore all-reduce of the model gradients, started behind "
                    "the forward and joined at the end of the step; a 0.2 ms step cannot hide it"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    backend = os.environ.get("SD_BENCH_BACKEND", "nccl")  # "gloo": launcher-path test on CPU boxes
    # SD_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, collectives, the all-reduce legs) with ONE rank --
    # how the RCCL side of this file is exercised on a single-GPU box (`torchrun --nproc-per-node 1 bench.py --gpus 1`)
    multi = world > 1 or os.environ.get("SD_BENCH_FORCE_DIST") == "1"
    use_cuda = backend == "nccl"
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if use_cuda else {}
        dist.init_process_group(backend, **kw)
    if not use_cuda:
        return launcher_selftest(args, rank, world)

    from simpledet_amd import dist as sdd
    from simpledet_amd import ops, synth
    from simpledet_amd._lib import lib

    for kv in args.tuning:
        k, v = kv.split("=")
        lib().set_tuning(k, int(v))

    strides = list(synth.FPN_STRIDES)
    shapes = synth.FPN_SHAPES
    # disjoint synthetic images per rank
    feats_np = synth.feature_maps(args.seed + 17 * rank, args.images, args.channels, shapes)
    rois_np = synth.random_rois(args.seed + 17 * rank, args.images, args.rois)
    feats = [torch.from_numpy(f).cuda() for f in feats_np]
    rois = torch.from_numpy(rois_np).cuda()
    out_shape = (args.images, args.rois, args.channels, 7, 7)
    dy = torch.randn(out_shape, device="cuda")
    d_feats = [torch.empty_like(f) for f in feats]
    grad_mb = args.grad_allreduce if args.grad_allreduce >= 0 else (R50_FPN_GRAD_MB if multi else 0.0)
    reducer = None
    rccl_ranks = 1
    if multi:
        # a real collective before anything is timed: every rank contributes 1
        one = torch.ones(1, device="cuda")
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        if grad_mb > 0:
            reducer = sdd.OverlappedAllReduce(grad_mb * 1e6, device="cuda")

    state = {}

    def step(ev=None, reducer=None):
        if ev:
            ev[0].record()
        if args.float_argmax:
            out, ax, ay = ops.fpn_roi_align_forward(feats, rois, strides, (7, 7))
        else:
            out, am = ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7), plan=not args.no_plan)
        if ev:
            ev[1].record()
            state["fwd_dispatch"] = (lib().cdll.sd_last_dispatch() or b"").decode()
        if reducer is not None:
            # the gradients of the layers behind the RoI head exist by now: their all-reduce runs
            # under the RoIAlign backward, as in a data-parallel training step
            reducer.start()
        if args.float_argmax:
            ops.fpn_roi_align_backward(dy, rois, ax, ay, None, strides, d_feats=d_feats)
        else:
            ops.fpn_roi_align_backward_packed(dy, rois, am, None, strides, d_feats=d_feats)
        if ev:
            ev[2].record()
            state["bwd_dispatch"] = (lib().cdll.sd_last_dispatch() or b"").decode()
        if reducer is not None:
            reducer.finish()  # the optimizer needs the reduced gradients: the step ends here
        state["out"] = out
        state["arg"] = (ax, ay) if args.float_argmax else am

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # bring the GPU out of its idle clock / power state first (untimed, ~0.1 s of the same step):
    # with a short --warmup the first timed steps otherwise run while the clocks are still ramping
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preheat_ms * 1e-3:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()

    def max_over_ranks(x):
        if not multi:
            return x, [x]
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        vals = [float(v.item()) for v in allt]
        return max(vals), vals

    # ---- THE timed region: exactly K steps of the path on every rank, nothing else in it ----
    # HIP events inside it split forward / backward for the roofline -- on every `event_stride`-th step
    # only: a record between two launches is a marker packet + release fence that
    # costs the step it times 3-4 us (profiles/r05b_event_cost.txt), and with three of them in every step
    # the timed region measured its own instrumentation (0.1928 against 0.1817 ms per step, same box).
    stride = max(1, int(args.event_stride))
    # (steps s/2, s/2 + s, ...: not step 0, whose first launch follows the barrier's synchronise into an empty
    # queue -- its event interval would time the launch latency, not the kernel)
    event_steps = [i for i in range(args.steps) if i % stride == stride // 2] or [args.steps - 1]
    events = {i: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for i in event_steps}
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events.get(i))
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed, vals = max_over_ranks(elapsed_local)
    per_rank_ms = [v * 1e3 / args.steps for v in vals]
    scaling = {}
    if multi:
        # afterwards, outside the timed region: rank 0 alone (the N = 1 figure of this box in this
        # invocation), then the reference's gradient all-reduce beside the path
        t_single = leg_single_rank(step, barrier, args, rank, torch.device("cuda"))
        scaling = {"n1_ms_per_step_same_invocation": t_single * 1e3 / args.steps,
                   "weak_scaling_eff_same_invocation": t_single / elapsed}
        if reducer is not None:
            scaling["grad_allreduce"] = legs_grad_allreduce(step, reducer, barrier, max_over_ranks, args, world,
                                                            grad_mb, t_single, elapsed)

    fwd_samples = [events[i][0].elapsed_time(events[i][1]) for i in event_steps]
    bwd_samples = [events[i][1].elapsed_time(events[i][2]) for i in event_steps]
    # beside the line's value (not part of it): the same K steps with NO event in them, and with events in
    # EVERY step (round 4's timed region), so that what the instrumentation costs is measured in this very run
    instr = {}
    if world == 1:
        def leg(evs):
            barrier()
            t = time.perf_counter()
            for i in range(args.steps):
                step(evs[i] if evs else None)
            barrier()
            return (time.perf_counter() - t) * 1e3 / args.steps
        every = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        instr = {"ms_per_step_no_events": leg(None), "ms_per_step_events_in_every_step": leg(every),
                 "ms_per_step_no_events_again": leg(None)}
    fwd_ms = float(np.mean(fwd_samples))
    bwd_ms = float(np.mean(bwd_samples))
    ms_per_step = elapsed * 1e3 / args.steps
    total_images = args.images * world * args.steps
    value = total_images / elapsed

    extra = {}
    if not args.no_extra and rank == 0:
        # what an UNCHANGED reference graph executes through the drop-in per-level operators
        # (models/FPN/builder.py:588-605): assign -> 4 x ROIAlign_v2 (3 full-size outputs each) -> add_n
        def graph_step():
            per, _ = ops.fpn_roi_assign(rois, strides)
            total = None
            for f, p, s in zip(feats, per, strides):
                o, _, _ = ops.roi_align_v2_forward(f, p, (7, 7), 1.0 / s)
                total = o if total is None else total + o
            return total
        for _ in range(3):
            graph_step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graph_step()
        e1.record()
        torch.cuda.synchronize()
        extra["unfused_4op_fwd_ms"] = e0.elapsed_time(e1) / 10
        extra["fused_vs_unfused_fwd"] = extra["unfused_4op_fwd_ms"] / fwd_ms

    if args.calibrate and rank == 0:
        import ctypes
        nbytes = 1 << 30  # 1 GiB src + 1 GiB dst: far past the 256 MiB Infinity Cache
        src = torch.empty(nbytes // 4, device="cuda", dtype=torch.float32).normal_()
        dst = torch.empty_like(src)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cal = {}
        for width in (16, 8, 4):
            for _ in range(2):
                lib().call("sd_hbm_stream_copy", ctypes.c_void_p(src.data_ptr()),
                           ctypes.c_void_p(dst.data_ptr()), nbytes, width, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib().call("sd_hbm_stream_copy", ctypes.c_void_p(src.data_ptr()),
                           ctypes.c_void_p(dst.data_ptr()), nbytes, width, st)
            e1.record()
            torch.cuda.synchronize()
            cal["copy%dB_GBs" % width] = 2 * nbytes / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9
        cal["bytes_read_per_launch"] = nbytes
        cal["bytes_written_per_launch"] = nbytes
        extra["hbm_stream_copy"] = cal
        del src, dst

    if rank != 0:
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    alg = algorithmic_bytes(args.images, args.rois, args.channels, shapes)
    # HBM-side bytes per forward launch: NOT measured in this run (PMC counters need rocprofv3); the
    # newest committed PMC profile of this same command is quoted and named.  2 x FETCH_SIZE (gfx950
    # correction, confirmed on the known-size hbm_stream_copy in the same profile) + WRITE_SIZE
    traffic, traffic_source, bwd_traffic = None, None, None
    import glob
    import re
    # profiles of THIS command only (rNN<letter>_pmc_summary.json); *_ops_* are bench_ops profiles and
    # stale_* are superseded ones
    cands = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))
             if re.match(r"^r\d+[a-z]?_pmc_summary\.json$", os.path.basename(f))]
    src_sha = kernel_source_sha()
    stale = []
    for pmc_path in sorted(cands)[::-1]:
        try:
            prof = json.load(open(pmc_path))
            # a profile is quoted only if it was taken with THESE kernel sources (tools/summarize_profile.py
            # records the hash of simpledet_amd/csrc/{roi_align_*.{h,hip},common.h,runtime.hip} next to the counters)
            if prof.get("kernel_source_sha256") != src_sha:
                stale.append(os.path.basename(pmc_path))
                continue
            ks = prof["kernels"]
            # the step's kernel is the most-dispatched forward / backward kernel of the profile
            # (the per-level kernels of the --extra leg run a handful of times)
            for pat in ("roi_align_fwd", "roi_align_bwd"):
                cand = [(d.get("n_dispatch", 0), d) for name, d in ks.items()
                        if pat in name and "bwd_lists" not in name and "fetch_bytes_x2_gfx950" in d]
                if cand:
                    d = max(cand, key=lambda t: t[0])[1]
                    tot = d["fetch_bytes_x2_gfx950"] + d.get("write_bytes", 0.0)
                    pre = ("bwd_lists",) if pat == "roi_align_bwd" else ("roi_fwd_prep", "roi_prep_merged")
                    tot += sum(x["fetch_bytes_x2_gfx950"] + x.get("write_bytes", 0.0)
                               for name, x in ks.items()
                               if any(q in name for q in pre) and "fetch_bytes_x2_gfx950" in x)
                    if pat == "roi_align_fwd":
                        traffic = tot
                    else:
                        bwd_traffic = tot
            if traffic is not None:
                traffic_source = ("%s (rocprofv3 --pmc passes of this command with these kernel sources, sha256 %s; "
                                  "not re-measured in this run)" % (os.path.relpath(pmc_path, ROOT), src_sha[:12]))
                break
        except Exception:
            traffic = None
    if traffic is None:
        traffic_source = ("none: no committed PMC profile was taken with the current kernel sources (sha256 %s)%s"
                          % (src_sha[:12], "; stale: " + ", ".join(stale[:4]) if stale else ""))
    fwd_kernel = state.get("fwd_dispatch") or "unknown"
    bwd_kernel = state.get("bwd_dispatch") or "unknown"
    roofline = {
        "kernel": fwd_kernel,
        "bound": "hbm",
        "achieved": alg / (fwd_ms * 1e-3) / 1e9,
        "peak": PEAK_HBM_GBS,
        "unit": "GB/s",
        "frac": alg / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "algorithmic_bytes": alg,
        "avg_launch_ms": fwd_ms,
        "kernel_note": "names reported by the library for the dispatch it took (sd_last_dispatch); avg_launch_ms "
                       "spans every launch of the forward op incl. its rois-only pre-pass",
        "event_steps": event_steps,
        "instrumentation": instr,
        "event_note": "HIP events on the torch stream the kernels are launched on, inside the timed region, on "
                      "every %d-th step (a record between launches costs 3-4 us of the step it times)" % stride,
        "fwd_ms_min_max": [float(min(fwd_samples)), float(max(fwd_samples))],
        "bwd_ms_min_max": [float(min(bwd_samples)), float(max(bwd_samples))],
        "backward": {
            "kernel": bwd_kernel,
            "achieved": alg / (bwd_ms * 1e-3) / 1e9,
            "frac": alg / (bwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "avg_ms": bwd_ms,
            "traffic": bwd_traffic,
        },
        "fwd_bwd_frac": 2 * alg / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS,
    }

    cpu_baseline = None
    if not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        dy_np = dy.cpu().numpy()
        fshapes = [f.shape for f in feats_np]
        # warm (page in) once; the multi-thread figure uses the thread count that is FASTEST on this
        # host (more OpenMP threads than memory channels can be slower than one: measured 0.67 img/s
        # on 256 threads against 1.48 on one), found with one forward pass per candidate
        o = orc.fpn_roi_align_fwd(feats_np, rois_np, strides, (7, 7), nthreads=min(avail, 16))
        cores, best = 1, None
        for cand in sorted({avail, max(1, avail // 4), min(avail, 32), min(avail, 8)}):
            t0 = time.perf_counter()
            orc.fpn_roi_align_fwd(feats_np, rois_np, strides, (7, 7), nthreads=cand)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                cores, best = cand, dt
        wd = None
        if world == 1:
            t0 = time.perf_counter()
            for _ in range(args.cpu_passes):
                o = orc.fpn_roi_align_fwd(feats_np, rois_np, strides, (7, 7), nthreads=cores)
                wd = orc.fpn_roi_align_bwd(dy_np, rois_np, o[1], o[2], fshapes, strides, nthreads=cores)
            cpu_t = time.perf_counter() - t0
            t0 = time.perf_counter()
            o1 = orc.fpn_roi_align_fwd(feats_np, rois_np, strides, (7, 7), nthreads=1)
            orc.fpn_roi_align_bwd(dy_np, rois_np, o1[1], o1[2], fshapes, strides, nthreads=1)
            cpu_t1 = time.perf_counter() - t0
            cpu_baseline = {
                "value": args.images * args.cpu_passes / cpu_t,
                "unit": "images/s",
                "cores": cores,
                "cores_available": avail,
                "kind": "port",
                "sample": "%d passes of the same fwd+bwd workload (N=%d, %d RoIs/img) through "
                          "oracle/liboracle.so on %d threads (OpenMP over outputs / planes, glue "
                          "loops included); single_thread: 1 pass on 1 thread"
                          % (args.cpu_passes, args.images, args.rois, cores),
                "ms_per_step": cpu_t * 1e3 / args.cpu_passes,
                "single_thread": {"value": args.images / cpu_t1, "unit": "images/s", "cores": 1,
                                  "ms_per_step": cpu_t1 * 1e3},
            }
        else:
            wd = orc.fpn_roi_align_bwd(dy_np, rois_np, o[1], o[2], fshapes, strides, nthreads=cores)
        # the reference's OWN compiled CPU forward beside the port: operator_cxx/contrib/roi_align_v2.cc
        # built where it lies into oracle/_ref/libref_roi_align_v2.so (oracle/build_ref_cxx.py), run as
        # the graph the reference builds (assign -> 4 x ROIAlign_v2 -> add_n); serial, as the shim's
        # Kernel::Launch is (MXNet's CPU Launch uses OpenMP: the multi-thread port above stands for that)
        if cpu_baseline is not None:
            try:
                from oracle import refmx
                if refmx.available("roi_align_v2"):
                    per, _ = orc.fpn_roi_assign(rois_np, strides)
                    t0 = time.perf_counter()
                    ref_out = None
                    for l, st_ in enumerate(strides):
                        rop = refmx.RefOp("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=(7, 7),
                                          spatial_scale=1.0 / st_)
                        r_ = rop.forward([feats_np[l], per[l]], ctx="cpu")[0]
                        ref_out = r_ if ref_out is None else ref_out + r_
                    t_ref = time.perf_counter() - t0
                    cpu_baseline["reference"] = {
                        "value": args.images / t_ref, "unit": "images/s (forward only)", "cores": 1,
                        "kind": "reference", "ms_forward": t_ref * 1e3,
                        "sample": "1 pass of the forward graph (fpn_roi_assign -> 4 x ROIAlign_v2 -> add_n, "
                                  "roi_align_v2-inl.h:157-195) through oracle/_ref/libref_roi_align_v2.so = "
                                  "the reference's roi_align_v2.cc compiled against oracle/mxshim",
                        "equals_gpu_forward_bit_for_bit": bool(np.array_equal(ref_out, state["out"].cpu().numpy())),
                    }
                    # ... and its compiled BACKWARD (round 6).  The operator's semantics are the GPU scatter kernel
                    # (roi_align_v2.cu:35-84, what this library's backward reproduces), run here from the reference's
                    # own source on one host core (oracle/mxshim/cuemu.h: the CUDA grid as a serial loop, atomicAdd
                    # as +=), level by level as the un-fused graph does; its CPU file's gather kernel
                    # (roi_align_v2.cc:35-106) is O(pixels x RoIs) -- timed on P5 only, where it takes seconds.
                    ref_fw = []
                    for l, st_ in enumerate(strides):
                        rop = refmx.RefOp("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=(7, 7),
                                          spatial_scale=1.0 / st_)
                        ref_fw.append((rop, rop.forward([feats_np[l], per[l]], ctx="cpu")))
                    t0 = time.perf_counter()
                    ref_dx = [rop.backward([dy_np], [feats_np[l], per[l]], fw, ctx="gpu")[0]
                              for l, (rop, fw) in enumerate(ref_fw)]
                    t_bwd = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    rop5, fw5 = ref_fw[-1]
                    rop5.backward([dy_np], [feats_np[-1], per[-1]], fw5, ctx="cpu")
                    t_gather5 = time.perf_counter() - t0
                    gd = [g.cpu().numpy() for g in d_feats]
                    cpu_baseline["reference"].update({
                        "ms_backward": t_bwd * 1e3,
                        "value_fwd_bwd": args.images / (t_ref + t_bwd), "unit_fwd_bwd": "images/s",
                        "backward_sample": "1 pass of ROIAlignBackwardKernelGPU_v2 (roi_align_v2.cu, the operator's "
                                           "semantics) over the four levels, emulated serially on one core",
                        "backward_max_abs_diff_to_gpu": max(float(np.abs(a_ - b_).max()) for a_, b_ in zip(ref_dx, gd)),
                        "cpu_gather_backward_p5_only_ms": t_gather5 * 1e3,
                    })
            except Exception as e:  # the reference library is optional on the GPU box
                cpu_baseline["reference"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the GPU results of the LAST TIMED step against the oracle (not timed): forward values bit
        # for bit, gradients elementwise within 1e-4
        fwd_ok = bool(np.array_equal(state["out"].cpu().numpy(), o[0]))
        bwd_err = max(float(np.abs(g.cpu().numpy() - w).max()) for g, w in zip(d_feats, wd))
        extra["matches_oracle"] = bool(fwd_ok and bwd_err <= 1e-4)
        extra["verify"] = {"forward_bit_exact": fwd_ok, "backward_max_abs_err": bwd_err,
                           "backward_tol": 1e-4, "what": "outputs of the last timed step vs oracle/"}

    if world == 1 and not args.no_ops:
        import bench_ops
        del feats, d_feats, dy
        torch.cuda.empty_cache()
        extra["ops"] = bench_ops.run(args.seed, cpu=not args.no_cpu_baseline)
        l2, l2_src = nms_l2_hit_rates()
        for key, kernels in (("nms", ("nms_sort_kernel", "nms_mask_kernel", "nms_scan_kernel")),
                             ("soft_nms", ("soft_nms_kernel",))):
            if key in extra["ops"]:
                extra["ops"][key]["l2_hit"] = ({k: l2[k] for k in kernels if k in l2} if l2 else None)
                extra["ops"][key]["l2_hit_source"] = l2_src
        if cpu_baseline is not None and "cpu_ms" in extra["ops"]["nms"]:
            # north_star: GPU vs CPU throughput of RoIAlign fwd+bwd + NMS on the same inputs
            gpu_ms = ms_per_step + extra["ops"]["nms"]["ms"]
            cpu_ms = cpu_baseline["ms_per_step"] + extra["ops"]["nms"]["cpu_ms"]
            extra["roialign_plus_nms_speedup_vs_cpu"] = cpu_ms / gpu_ms

    line = {
        "metric": "images/sec at 800x1333 FPN 512-RoI RoIAlign fwd+bwd",
        "value": value,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "fused FPN RoIAlign_v2 fwd+bwd, P2-P5 %dch 800x1333, N=%d img/GPU, %d RoIs/img, "
                        "7x7x4 samples (BASELINE configs[1])" % (args.channels, args.images, args.rois),
            "images_per_gpu": args.images,
            "rois_per_image": args.rois,
            "argmax_state": "fp32 x,y planes" if args.float_argmax else "packed u8 (decoded in backward)",
            "sharding": "images across ranks, no data-path collective",
            "collectives_in_value": "none (the path has no exchange step; see grad_allreduce for the "
                                    "reference's gradient all-reduce measured beside it)",
        },
        "rccl_ranks": rccl_ranks,
        "per_rank_ms_per_step": per_rank_ms,
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
    }
    line.update(scaling)
    line.update(extra)
    print(json.dumps(line))
    sys.stdout.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def launcher_selftest(args, rank, world):
    """SD_BENCH_BACKEND=gloo: exercise the launcher path (self-spawn, rendezvous, per-step overlapped
    gradient all-reduce, max-over-ranks timing, one JSON line from rank 0) on a box without GPUs.
    No kernel runs and the line says so; tests/test_dist.py drives this with --gpus 2."""
    import torch
    import torch.distributed as dist
    from simpledet_amd import dist as sdd
    grad_mb = args.grad_allreduce if args.grad_allreduce >= 0 else (R50_FPN_GRAD_MB if world > 1 else 0.0)
    one = torch.ones(1)
    if world > 1:
        dist.all_reduce(one)
    reducer = sdd.OverlappedAllReduce(grad_mb * 1e6) if world > 1 and grad_mb > 0 else None
    work = torch.randn(256, 256)

    def ops_only_step():  # stand-in for the hot path: the legs and their bookkeeping are what is tested
        (work @ work).sum().item()

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x, [x]
        t = torch.tensor([x], dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        vals = [float(v.item()) for v in allt]
        return max(vals), vals

    def step(reducer=None):
        ops_only_step()
        if reducer is not None:
            reducer.buf.fill_(float(rank + 1))
            reducer.start()
            reducer.finish()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed, vals = max_over_ranks(time.perf_counter() - t0)
    scaling = {}
    if world > 1:
        t_single = leg_single_rank(step, barrier, args, rank, torch.device("cpu"))
        scaling = {"n1_ms_per_step_same_invocation": t_single * 1e3 / args.steps,
                   "weak_scaling_eff_same_invocation": t_single / elapsed}
        if reducer is not None:
            scaling["grad_allreduce"] = legs_grad_allreduce(step, reducer, barrier, max_over_ranks, args, world,
                                                            grad_mb, t_single, elapsed)
    ok = reducer is None or bool(torch.allclose(reducer.buf, torch.full_like(reducer.buf, (world + 1) / 2.0)))
    if rank == 0:
        line = {"metric": "launcher self-test (no GPU kernels)", "value": None, "n_gpus": world,
                "steps": args.steps, "rccl_ranks": int(one.item()), "backend": "gloo",
                "ms_per_step": elapsed * 1e3 / max(1, args.steps),
                "allreduce_correct": ok,
                "per_rank_ms_per_step": [v * 1e3 / max(1, args.steps) for v in vals]}
        line.update(scaling)
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
