#!/usr/bin/env python3
"""bench.py -- keypoints+descriptors/s of the MI355X ORB front-end (BASELINE.json metric).

A "step" = one pass of the whole extractor hot path (pyramid -> FAST -> quadtree -> pack ->
orientation+blur+descriptor with host-libm-exact trig) over one batch of synthetic frames that
is ALREADY RESIDENT in HBM; outputs stay in HBM.  Workload = BASELINE.json configs[1]: 752x480,
8 levels, scale 1.2, nFeatures 1000, FAST 20/7 -- as a batch of --batch frames per GPU per step.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank extracts its own
shard of frames (weak scaling) and ONE RCCL all-gather per step exchanges the descriptor slabs for
cross-camera matching.  Timing: W warm-up steps, then exactly K steps between barrier +
torch.cuda.synchronize(), MAX over ranks; rank 0 prints one JSON line.

Extra objects in the line:
  roofline     -- dominant kernel (by hipEvent time on the extractor's stream, measured live over
                  the timed steps): algorithmic bytes per launch / average launch time vs 8 TB/s HBM.
  pipelined    -- not `value`: the same batches alternating between two extractor contexts on two streams.
  single_frame -- not `value`: configs[1] read literally, ONE resident frame per call (latency-bound).
  cpu_baseline -- the CPU oracle (a port of the reference path; the reference itself cannot be
                  built without OpenCV) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def level_sizes(rows, cols, nlevels=8, scale=1.2):
    sf = np.float32(1.0)
    out = []
    for _ in range(nlevels):
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(cols) * inv)), int(np.rint(np.float32(rows) * inv))))
        sf = np.float32(np.float64(sf) * np.float64(np.float32(scale)))
    return out


def algorithmic_bytes_per_frame(rows, cols, n_kp, nlevels=8):
    """Per-stage split of SURVEY.md section 8d's B_frame = 5*SumP - P_7 + 2390*N."""
    P = [w * h for (w, h) in level_sizes(rows, cols, nlevels)]
    sp = sum(P)
    return {
        "pyramid": 2 * sp - P[-1],          # SumP_{l<7} read + SumP written
        "fast": sp,                          # every level read once
        "octree": 0,                         # candidate lists: second order, excluded by 8d
        "pack": 28 * n_kp,                   # KeyPoint records written
        "desc": 2 * sp + (31 * 31 + 37 * 37) * n_kp + 32 * n_kp,  # blur r/w + patches + descriptors
        "trigfix": 0,                        # host libm check + fix-up of a handful of keypoints
    }


def usable_cores():
    """CPUs this process may actually use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(rows, cols, nfeatures, seconds=12.0):
    """Oracle extractor (C++ threads, one extractor per thread) over independent frames, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle_py as O
    from orb_slam3_detailed_comments_kor_amd import synth
    O.build()
    cores = usable_cores()
    frames = np.stack([synth.make_frame(rows, cols, 1234 + i) for i in range(4)])
    # single-thread rate first (mono protocol, reference src/Frame.cc:306)
    n1, t1 = O.extract_many(frames, 1, 4, nfeatures, lap=(0, 1000))
    reps1 = max(4, int(0.2 * seconds / (t1 / 4)))
    n1, t1 = O.extract_many(frames, 1, reps1, nfeatures, lap=(0, 1000))
    per_frame = t1 / reps1
    # all cores: calibrate with 2 frames per thread, then size the run for the remaining budget
    nc, tc = O.extract_many(frames, cores, 2, nfeatures, lap=(0, 1000))
    reps = int(min(max(2, 0.6 * seconds / (tc / 2)), 200))
    n, dt = O.extract_many(frames, cores, reps, nfeatures, lap=(0, 1000))
    return {
        "value": n / dt,
        "unit": "keypoints/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d threads x %d frames of %dx%d (nF=%d) in %.1f s; 1 thread: %.0f keypoints/s (%.1f ms/frame)"
                  % (cores, reps, cols, rows, nfeatures, dt, n1 / t1, 1e3 * per_frame),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=752)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--trig", choices=["libm", "cr", "hostcheck"], default="libm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=6,
                    help="record the per-stage hipEvents on every N-th timed step (7 event records cost ~25 us)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-context measurement")
    ap.add_argument("--contexts", type=int, default=1,
                    help="experiment: consecutive steps alternate between this many extractor contexts, each with "
                         "its own stream and output buffers (like the reference's left/right extractor threads)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orb_slam3_detailed_comments_kor_amd as pkg
    from orb_slam3_detailed_comments_kor_amd.multicam import PipelinedExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("ORBFE_BENCH_FORCE_DIST"):  # the latter: rehearse the N>1 code path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    B, H, W = args.batch, args.rows, args.cols
    # distinct frames per rank: a few generated frames, horizontally rolled to fill the batch
    nuniq = min(B, 8)
    base = [pkg.synth.make_frame(H, W, 1234 + rank * 1000 + i) for i in range(nuniq)]
    imgs = np.stack([np.roll(base[i % nuniq], 23 * (i // nuniq), axis=1) for i in range(B)])
    d_img = torch.from_numpy(imgs).to(dev)

    ex = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank,
                          trig={"libm": pkg.binding.TRIG_LIBM, "cr": pkg.binding.TRIG_CR,
                                "hostcheck": pkg.binding.TRIG_LIBM_HOSTCHECK}[args.trig])
    # One explicit stream for everything: the extractor's kernels, torch's ops and the RCCL collective
    # (which orders itself after the current stream) -- the legacy null stream would not order a
    # non-blocking stream.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    cap = ex.max_keypoints(H, W)
    # two slab pairs: the all-gather of batch i (process group's stream) overlaps the extraction of batch i+1
    pipe = PipelinedExchange(B, cap, dev, world, rank)
    d_desc = pipe.x[0].desc_view()
    d_n = pipe.x[0].count_view()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
    lap = (0, 1000)  # mono protocol, src/Frame.cc:306

    # --contexts N (experiment, single GPU): further extractor contexts with their own streams and outputs
    extra = []
    if args.contexts > 1 and world == 1:
        for _ in range(args.contexts - 1):
            e2 = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank, trig=ex.trig)
            s2 = torch.cuda.Stream(device=dev)
            e2.set_stream(s2.cuda_stream)
            extra.append((e2, s2, torch.zeros((B, cap, 7), dtype=torch.float32, device=dev),
                          torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                          torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    counter = [0]

    def step():
        k = counter[0] % (1 + len(extra))
        counter[0] += 1
        if k > 0:
            e2, _, k2, de2, n2, m2 = extra[k - 1]
            e2.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, k2.data_ptr(), de2.data_ptr(), cap,
                                    n2.data_ptr(), m2.data_ptr())
            return
        x = pipe.begin()  # waits (on the stream) for the collective that last read this slab
        ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, d_kps.data_ptr(), x.desc_view().data_ptr(), cap,
                                x.count_view().data_ptr(), d_mono.data_ptr())
        if dist.is_initialized():
            pipe.submit()  # one RCCL all-gather of descriptor slabs per batch, asynchronous

    def barrier():
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ex.profile(args.event_every)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # every event_every-th call records one set of stage events on the stream (no extra sync)
    barrier()
    dt = time.perf_counter() - t0
    stage_ms = ex.stage_ms()  # hipEvent times averaged over the timed steps
    ex.profile(False)

    # Not part of `value`: the same batches alternating between TWO extractor contexts on two streams (how
    # a multi-camera rig drives one extractor per camera, reference src/Frame.cc:119-122).  Consecutive
    # batches then overlap on the GPU, which hides the latency-bound stages (octree, pack) and launch gaps.
    pipelined = None
    if world == 1 and not extra and not args.no_pipelined:
        e2 = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank, trig=ex.trig)
        s2 = torch.cuda.Stream(device=dev)
        e2.set_stream(s2.cuda_stream)
        k2 = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        de2 = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        n2 = torch.zeros(B, dtype=torch.int32, device=dev)
        m2 = torch.zeros(B, dtype=torch.int32, device=dev)

        def step2(i):
            if i & 1:
                e2.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, k2.data_ptr(), de2.data_ptr(), cap,
                                        n2.data_ptr(), m2.data_ptr())
            else:
                ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, d_kps.data_ptr(), d_desc.data_ptr(),
                                        cap, d_n.data_ptr(), d_mono.data_ptr())
        for i in range(4):
            step2(i)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        assert torch.equal(n2, d_n)
        pipelined = {"contexts": 2, "ms_per_step": 1e3 * tp / args.steps,
                     "value": float(d_n.sum().item()) * args.steps / tp, "unit": "keypoints/s"}
        e2.close()

    # Also not part of `value`: BASELINE configs[1] read literally -- ONE resident frame per call (what a monocular
    # tracker issues), the latency-bound end of the same pipeline.
    single = None
    if world == 1 and not extra and not args.no_pipelined and B > 1:
        k1 = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
        de1 = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
        n1 = torch.zeros(1, dtype=torch.int32, device=dev)
        m1 = torch.zeros(1, dtype=torch.int32, device=dev)

        def step1():
            ex.extract_batch_device(d_img.data_ptr(), 1, H, W, W, H * W, lap, k1.data_ptr(), de1.data_ptr(), cap,
                                    n1.data_ptr(), m1.data_ptr())
        for _ in range(20):
            step1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step1()
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t1
        single = {"frames_per_call": 1, "ms_per_frame": 1e3 * t1 / args.steps,
                  "value": float(n1.item()) * args.steps / t1, "unit": "keypoints/s"}

    n_local = int(d_n.sum().item())
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    cnt = torch.tensor([n_local], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    kp_per_step = float(cnt.item())

    if rank == 0:
        dom = max(stage_ms, key=stage_ms.get)
        abytes = algorithmic_bytes_per_frame(H, W, n_local / B)
        launch_bytes = abytes[dom] * B
        achieved = launch_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        # HBM bytes per launch from the rocprofv3 counter passes (tools/collect_profiles.sh, collected in
        # separate --pmc runs and corrected with the FETCH_SIZE calibration), if they match this workload
        traffic = None
        valu_issue = None
        per_kernel = {}
        kernel_of = {"pyramid": "k_pyr_fused", "fast": "k_fast_cells", "octree": "k_octree", "pack": "k_pack",
                     "desc": "k_orient_blur_desc<0", "trigfix": "k_orient_blur_desc<1"}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("workload") == {"batch": B, "rows": H, "cols": W, "nfeatures": args.nfeatures}:
                    for st, kn in kernel_of.items():
                        for k, e in j.get("kernels", {}).items():
                            if k.startswith(kn) and "hbm_bytes_per_launch" in e:
                                per_kernel[st] = e["hbm_bytes_per_launch"]
                    for k, e in j.get("kernels", {}).items():
                        if k.startswith(kernel_of[dom]) and "hbm_bytes_per_launch" in e:
                            traffic = e["hbm_bytes_per_launch"]
                            # share of the chip's VALU issue slots this kernel's wave-instructions occupy
                            # (1024 SIMDs, 4 cycles per wave64 VALU instruction, 2.4 GHz): the bound that
                            # actually binds these integer kernels
                            if "SQ_INSTS_VALU" in e and e.get("avg_duration_us"):
                                valu_issue = e["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * e["avg_duration_us"] * 1e-6)
            except Exception:
                traffic = None
        out = {
            "metric": "keypoints+descriptors/sec on 752x480x8-level pyramid",
            "value": kp_per_step * args.steps / dt,
            "unit": "keypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d grayscale, 8-level pyramid, nFeatures=%d, FAST 20/7, %d frames/GPU/step "
                            "resident in HBM (BASELINE configs[1] batched)" % (W, H, args.nfeatures, B),
                "frames_per_step": B * world,
                "keypoints_per_step": kp_per_step,
                "trig": args.trig,
                "contexts": 1 + len(extra),
                "exchange": ("1 all-gather of descriptor slabs per step, overlapped with the next step's extraction"
                             if dist.is_initialized() else "none"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_of[dom],
                "stage": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "valu_issue_frac": valu_issue,
                "algorithmic_bytes_per_launch": launch_bytes,
                "event_sampling": "stage hipEvents on every %d-th of the timed steps" % max(args.event_every, 1),
                "avg_launch_ms": stage_ms[dom],
                "stage_ms": stage_ms,
                # the same figures for every kernel of the step (the two largest are within a few per cent of
                # each other, so which one is "dominant" can change from run to run) and for the whole step
                "kernels": {st: {"kernel": kernel_of[st], "avg_launch_ms": ms,
                                 "algorithmic_bytes_per_launch": abytes[st] * B,
                                 "achieved": abytes[st] * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                 "frac": (abytes[st] * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ms > 0 else 0.0,
                                 "traffic": per_kernel.get(st)}
                            for st, ms in stage_ms.items() if st != "trigfix"},
                "whole_step": {"algorithmic_bytes": sum(abytes.values()) * B,
                               "achieved": sum(abytes.values()) * B / dt * args.steps / 1e9,
                               "frac": sum(abytes.values()) * B / dt * args.steps / 1e9 / HBM_PEAK_GBPS},
            },
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if single is not None:
            out["single_frame"] = single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, W, args.nfeatures)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
