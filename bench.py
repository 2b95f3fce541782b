#!/usr/bin/env python3
"""bench.py -- keypoints+descriptors/s of the MI355X ORB front-end (BASELINE.json metric).

A "step" = one pass of the whole extractor hot path (pyramid -> FAST -> quadtree -> pack ->
orientation+blur+descriptor with host-libm-exact trig) over one batch of synthetic frames that
is ALREADY RESIDENT in HBM; outputs stay in HBM.  Workload = BASELINE.json configs[1]: 752x480,
8 levels, scale 1.2, nFeatures 1000, FAST 20/7 -- as a batch of --batch frames per GPU per step.
`value` is that resident rate because the bench contract of this build says so in as many words ("value is
whole-job throughput with inputs already resident in HBM when the timed region starts; if the boundary hands over
host buffers, note the PCIe-inclusive rate ... it is never value").  The rate SURVEY.md 8(d) defines for the
drop-in boundary -- host pointers in, host arrays out, H2D and D2H inside the clock -- is measured in the same run and
printed next to it: `boundary_value` (= pcie_inclusive.batch_pipelined), with `vs_cpu` ratios for both.
--config c4 switches to BASELINE configs[3]: 64 frames of 1280x720 IN TOTAL, sharded over the ranks (strong scaling).
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank extracts its own
shard of frames (weak scaling) and ONE RCCL all-gather per step exchanges the descriptor slabs for
cross-camera matching.  Timing: W warm-up steps, then exactly K steps between barrier +
torch.cuda.synchronize(), MAX over ranks; rank 0 prints one JSON line.

Clock settling: after the W warm-up steps the same step keeps running, untimed, until at least
--settle seconds (default 0.5) have passed, whatever W is -- with few warm-up steps the GPU clocks have
not ramped up and the timed steps read several per cent slow.  Then exactly K steps are timed.

Extra objects in the line (none of them is `value`):
  roofline       -- dominant kernel (by hipEvent time on the extractor's stream, measured live over
                    the timed steps): algorithmic bytes per launch / average launch time vs 8 TB/s HBM.
                    `traffic` (HBM bytes per launch) and the wave-instruction count behind `issue_frac` are MEASURED for
                    the tree that runs: two `rocprofv3 --pmc` child passes of this script (--pmc-child: the same
                    batches, one lane) before the timed region, corrected by a known-size copy in the same pass
                    (--no-pmc skips them; the fields are then null).  `issue_frac` has ONE definition, in
                    tools/isa/issue_table.py.  `step` prices the whole step by SURVEY 8(d)'s bytes.
  pcie_inclusive -- the metric as SURVEY.md 8(d) defines it for the drop-in boundary: wall time of
                    orbfe_extract* with HOST pointers, H2D of the images and D2H of keypoints +
                    descriptors included; measured by the C++ caller tools/hostbench (child process):
                    single frame (pageable / pinned, p50 / p99), batch of 64 (pageable / pinned /
                    two batches in flight), the bare-copy PCIe floor of the same bytes, and the
                    reference's stereo protocol (2 extractors, 2 threads, + ComputeStereoMatches).
  cross_camera   -- the step plus the consumer of the exchanged descriptors: knn-2 of every local frame
                    against the next camera of the ring, read from the gathered buffer in place (one launch).
  pipelined      -- the same resident batches alternating between two extractor contexts on two streams.
  single_frame   -- configs[1] read literally, ONE resident frame per call (latency-bound).
  first_call_ms  -- the first extraction of the process (libm trig table build + upload, allocations).
  cpu_baseline   -- the CPU oracle (a port of the reference path; the reference itself cannot be
                    built without OpenCV), rebuilt on this host with BASELINE.md's flags and timed on a
                    bounded sample: 1 thread (mono protocol), 2 threads / 2 extractors (the reference's
                    stereo protocol, src/Frame.cc:119-122), and all usable cores.
"""
import argparse
import gc
import json
import os
import sys
import subprocess
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_LANES = None  # by batch size: 3 lanes below 16 frames per step (a chain of latency-bound kernels per batch), else 2
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
        "pack": 0,                           # (K-PACK only runs for fisheye rays since round 3; the records are K-DESC's)
        "desc": 2 * sp + (31 * 31 + 37 * 37) * n_kp + (32 + 28) * n_kp,  # blur r/w + patches + descriptors + KeyPoint records
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


def cpu_baseline(rows, cols, nfeatures, seconds=16.0):
    """The CPU path timed on this host's cores, bounded sample.  Two builds of the SAME source (oracle/orb_oracle.cpp, g++ -O3
    -march=native): the scalar port that is the parity oracle, and its timing-only fast path -- SIMD rejection test in FAST, a blur
    and a resize vertical pass the compiler vectorises, source rows of the resize reused as OpenCV reuses them -- whose results are
    asserted equal to the scalar path's on the timed frames before anything is timed (OpenCV's own FAST / blur /
    resize are SIMD code; a baseline timed on a scalar port flatters the GPU).  `value` is the FASTER of the two, all cores.
    Protocols of SURVEY.md 8(d): (i) 1 thread, (ii) 2 threads / 2 extractors, (iii) all usable cores; per-stage CPU ms from (i)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle_py as O
    from orb_slam3_detailed_comments_kor_amd import synth
    O.build()
    native = O.lib_native() is not None  # -O3 -march=native build made on THIS host; portable -O2 build otherwise
    flags = O.NATIVE_FLAGS if native else O.PORTABLE_FLAGS
    cores = usable_cores()
    frames = np.stack([synth.make_frame(rows, cols, 1234 + i) for i in range(4)])
    # the fast path computes what the scalar path computes: checked on the frames that are timed
    simd = False
    for f in frames:
        a = O.Extractor(nfeatures, 1.2, 8, 20, 7, native=native)
        b = O.Extractor(nfeatures, 1.2, 8, 20, 7, native=native)
        simd = b.set_fastpath(True)
        ra, rb = a.extract(f, (0, 1000)), b.extract(f, (0, 1000))
        if not (ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])):
            raise SystemExit("cpu_baseline: the timing-only fast path differs from the scalar oracle")

    def run(threads, budget, lap, fast, nf=nfeatures):
        n, t, _ = O.extract_many_stages(frames, threads, 2, nf, lap=lap, native=native, fastpath=fast)  # calibrate
        reps = int(min(max(2, budget / (t / 2)), 400))
        n, t, st = O.extract_many_stages(frames, threads, reps, nf, lap=lap, native=native, fastpath=fast)
        return n, t, reps, st

    def leg(fast, share):
        n1, t1, r1, st1 = run(1, 0.2 * share, (0, 1000), fast)       # (i) mono protocol, reference src/Frame.cc:306
        n2, t2, r2, _ = run(2, 0.2 * share, (0, 0), fast, 1200)      # (ii) stereo protocol: left + right extractor threads, nF 1200
        n, dt, reps, _ = run(cores, 0.5 * share, (0, 1000), fast)    # (iii) best-case CPU throughput over independent frames
        return {"value": n / dt, "unit": "keypoints/s", "cores": cores,
                "one_thread": {"value": n1 / t1, "ms_per_frame": 1e3 * t1 / r1,
                               "stage_ms": {k: 1e3 * v / r1 for k, v in st1.items()}},
                "two_threads_stereo": {"value": n2 / t2, "ms_per_pair": 1e3 * t2 / r2, "nfeatures": 1200,
                                       "note": "2 threads x 1 extractor each, one frame per thread per pair (src/Frame.cc:119-122, "
                                               "Examples/Stereo/EuRoC.yaml nFeatures 1200)"},
                "sample": "%d threads x %d frames of %dx%d (nF=%d) in %.1f s; 1 thread: %.0f keypoints/s (%.1f ms/frame, %d frames); "
                          "2 threads: %.1f ms per stereo pair (%d pairs)"
                          % (cores, reps, cols, rows, nfeatures, dt, n1 / t1, 1e3 * t1 / r1, r1, 1e3 * t2 / r2, r2)}

    vec = leg(True, 0.6 * seconds)
    sca = leg(False, 0.4 * seconds)
    out = dict(vec)
    out.update({
        "kind": "port",
        "variant": ("vectorised timing build of the port: " + ("AVX2 rejection test in FAST, " if simd else "scalar FAST (no AVX2 on this host), ")
                    + "blur and resize written for the compiler's vectoriser, resize rows reused; results asserted equal to the scalar "
                    "oracle on the timed frames"),
        "flags": "g++ " + flags,
        "scalar_port": sca,
        "sample": "oracle built with `g++ %s`, timing-only fast path; %s" % (flags, vec["sample"]),
    })
    return out


def bench_frames(rows, cols, batch, rank=0):
    """The synthetic batch: a few generated frames, horizontally rolled to fill the batch (distinct per rank)."""
    from orb_slam3_detailed_comments_kor_amd import synth
    nuniq = min(batch, 8)
    base = [synth.make_frame(rows, cols, 1234 + rank * 1000 + i) for i in range(nuniq)]
    return np.stack([np.roll(base[i % nuniq], 23 * (i // nuniq), axis=1) for i in range(batch)])


def startup_cost(rows, cols, nfeatures, device=0):
    """create + first call of a fresh process (tools/hostbench ... first) in three states of the libm trig table: built from
    libm with no cache file, full table from the cache file (expanded on the device), compact table from the cache file."""
    exe = os.path.join(ROOT, "tools", "hostbench")
    with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as f:
        f.write(bench_frames(rows, cols, 1).tobytes())
        path = f.name
    out = {}
    try:
        for tag, env in (("full_table_no_cache", {"ORBFE_TRIG_CACHE": "0", "ORBFE_TRIG_TABLE": "2"}),
                         ("full_table_cached", {"ORBFE_TRIG_TABLE": "2"}),
                         ("compact_table_cached", {"ORBFE_TRIG_TABLE": "1"})):
            e = dict(os.environ)
            e.update(env)
            r = subprocess.run([exe, path, str(rows), str(cols), "1", str(nfeatures), str(device), "first"], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=120, env=e)
            out[tag] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-200:]}
    finally:
        os.unlink(path)
    out["note"] = ("libm trig table of the process: 65 MB of codes cached in /dev/shm after the first build; the 1.03 GB table is "
                   "expanded from them on the device; ranks of a multi-process job default to the compact form")
    return out


def hostbench_mode(mode, rows, cols, batch, nfeatures, device=0):
    """tools/hostbench in one of its extra modes ("c5": fisheye stereo frame, "matcher": the matcher entry points from C++)."""
    exe = os.path.join(ROOT, "tools", "hostbench")
    if not os.path.exists(exe):
        raise SystemExit("tools/hostbench is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as f:
        f.write(bench_frames(rows, cols, batch).tobytes())
        path = f.name
    try:
        out = subprocess.run([exe, path, str(rows), str(cols), str(batch), str(nfeatures), str(device), mode],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    finally:
        os.unlink(path)
    if out.returncode != 0:
        raise SystemExit("tools/hostbench %s failed (%d): %s" % (mode, out.returncode, out.stderr[-500:]))
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_stereo_baseline(rows, cols, nfeatures, fisheye, seconds=10.0):
    """The CPU path of a stereo frame, timed on this host: the oracle's two extractors on two threads (src/Frame.cc:119-122),
    then the consumer of the pair -- Frame::ComputeStereoMatches (rectified, config 3) or the knn-2 brute force of
    Frame::ComputeStereoFishEyeMatches (config 5) -- on one thread, as the reference runs it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle_py as O
    from orb_slam3_detailed_comments_kor_amd import synth
    O.build()
    native = O.lib_native() is not None
    left, right = synth.make_stereo_pair(rows, cols, 51, shift=40 if fisheye else 12)
    frames = np.stack([left, right])
    lap = (0, cols - 1) if fisheye else (0, 0)
    # (the extraction through the timing-only fast path of the oracle, as in cpu_baseline; equality with the scalar path checked first)
    for f in frames:
        a, b = O.Extractor(nfeatures, 1.2, 8, 20, 7, native=native), O.Extractor(nfeatures, 1.2, 8, 20, 7, native=native)
        b.set_fastpath(True)
        ra, rb = a.extract(f, lap, cap=nfeatures * 2 + 256), b.extract(f, lap, cap=nfeatures * 2 + 256)
        if not (ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])):
            raise SystemExit("cpu_stereo_baseline: the timing-only fast path differs from the scalar oracle")
    n, t, _ = O.extract_many_stages(frames, 2, 2, nfeatures, lap=lap, native=native, fastpath=True)
    reps = int(min(max(2, 0.6 * seconds / (t / 2)), 200))
    n, t, st = O.extract_many_stages(frames, 2, reps, nfeatures, lap=lap, native=native, fastpath=True)
    ms_extract = 1e3 * t / reps
    n0, t0s, _ = O.extract_many_stages(frames, 2, max(2, reps // 3), nfeatures, lap=lap, native=native, fastpath=False)
    ms_extract_scalar = 1e3 * t0s / max(2, reps // 3)
    exL, exR = O.Extractor(nfeatures, 1.2, 8, 20, 7), O.Extractor(nfeatures, 1.2, 8, 20, 7)
    mL, kL, dL = exL.extract(left, lap)
    mR, kR, dR = exR.extract(right, lap)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 0.3 * seconds or k < 2:
        if fisheye:
            O.bfknn2(dL[mL:], dR[mR:])
        else:
            O.compute_stereo_matches(exL, exR, kL, dL, kR, dR, 47.90639384423901 / 435.2046959714599, 47.90639384423901)
        k += 1
    ms_match = 1e3 * (time.perf_counter() - t0) / k
    return {"value": (len(kL) + len(kR)) / ((ms_extract + ms_match) * 1e-3), "unit": "keypoints/s", "cores": 2, "kind": "port",
            "flags": "g++ " + (O.NATIVE_FLAGS if native else O.PORTABLE_FLAGS),
            "ms_per_pair": ms_extract + ms_match, "extract_ms_per_pair": ms_extract, "match_ms_per_pair": ms_match,
            "variant": "vectorised timing build of the port (cpu_baseline of the default run says what that is)",
            "extract_stage_ms_per_frame": {k: 1e3 * v / (2 * reps) for k, v in st.items()},
            "scalar_port": {"extract_ms_per_pair": ms_extract_scalar, "ms_per_pair": ms_extract_scalar + ms_match},
            "sample": "%d stereo pairs of %dx%d (nF=%d) on 2 threads + %d runs of %s on 1 thread" %
                      (reps, cols, rows, nfeatures, k, "the knn-2 brute force of ComputeStereoFishEyeMatches (python wrapper "
                       "around the C++ oracle)" if fisheye else "ComputeStereoMatches")}


def per_call_config(args, real_stdout):
    """BASELINE configs[2] (`--config c3`: EuRoC stereo pair per call, nFeatures 1200, rectified matching on the GPU) and
    configs[4] (`--config c5`: 1024 x 1024 fisheye stereo frame, nFeatures 1500, KB8 rays in the extractor, knn-2 matching +
    triangulation): per-CALL configurations at the drop-in boundary, measured by the C++ caller tools/hostbench with host
    images in and host arrays out.  `value` = keypoints (with descriptors) per second through the whole stereo frame."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.config == "c3":
        rows, cols, nf = 480, 752, 1200
        hb = pcie_inclusive(rows, cols, 8, nf, 0)
        fused = hb["stereo_pair_fused"]
        kp_pair = hb["stereo_pair"]["keypoints_per_s"] * hb["stereo_pair"]["ms_per_pair_mean"] * 1e-3
        ms = fused["pinned"]["ms_per_pair_p50"]
        protocols = {"one_call_pinned (orbfe_extract_stereo_pair)": fused["pinned"],
                     "one_call_pageable (orbfe_extract_stereo_pair)": fused["pageable"],
                     "two_threads_two_contexts (reference protocol, src/Frame.cc:119-122)":
                         {k: hb["stereo_pair"][k] for k in ("ms_per_pair_mean", "ms_per_pair_p50", "ms_per_pair_p99", "extract_ms_p50")},
                     "one_batched_call_then_resident_matching": {k: hb["stereo_pair_one_call"][k] for k in
                                                                 ("ms_per_pair_mean", "ms_per_pair_p50", "ms_per_pair_p99")}}
        try:  # round 5: the same frames as a STREAM with 1..4 frames in flight on one context (orbfe_extract_stereo_pair_submit)
            st = hostbench_mode("stream", rows, cols, 8, nf, 0)
            protocols["frames_in_flight (orbfe_extract_stereo_pair_submit / _wait, sustained ms per frame)"] = {
                "pinned": st["pinned"], "pageable": st["pageable"]}
        except (SystemExit, Exception) as e:  # noqa: BLE001
            protocols["frames_in_flight"] = {"error": str(e)}
        workload = ("EuRoC stereo pair per call: 2 x 752x480, nFeatures 1200, both extractions + Frame::ComputeStereoMatches on the "
                    "GPU, host images in, host keypoints / descriptors / mvuRight / mvDepth out (BASELINE configs[2])")
        matches = hb["stereo_pair"]["matches_per_pair"]
        fisheye = False
    else:
        rows, cols, nf = 1024, 1024, 1500
        hb = hostbench_mode("c5", rows, cols, 8, nf, 0)
        kp_pair = hb["keypoints_per_pair"]
        ms = hb["one_call"]["ms_per_pair_p50"]
        protocols = {"one_batched_call + orbfe_stereo_fisheye_matches": hb["one_call"],
                     "two_threads_two_contexts (reference protocol) + orbfe_stereo_fisheye_matches": hb["two_threads"]}
        if hb.get("one_call_pinned", {}).get("ms_per_pair_p50"):
            protocols["one_batched_call, page-locked images (orbfe_host_register) + orbfe_stereo_fisheye_matches"] = hb["one_call_pinned"]
        if "one_call_resident_matching" in hb:
            protocols["one_batched_call + orbfe_stereo_fisheye_matches on the resident descriptors"] = hb["one_call_resident_matching"]
        workload = ("TUM-VI-like fisheye stereo frame per call: 2 x 1024x1024, nFeatures 1500, KannalaBrandt8 bearing rays fused "
                    "into the extractor (orbfe_set_kb8), every keypoint in the lapping area, then "
                    "Frame::ComputeStereoFishEyeMatches (knn-2 + ratio + triangulation) on the GPU (BASELINE configs[4])")
        matches = hb["matches_per_pair"]
        fisheye = True
    out = {"metric": "keypoints+descriptors/sec through one stereo frame per call (%s)" % args.config, "value": kp_pair / (ms * 1e-3),
           "unit": "keypoints/s", "n_gpus": 1, "steps": hb.get("pairs", 200), "warmup": 10, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": workload, "keypoints_per_pair": kp_pair, "stereo_matches_per_pair": matches,
                      "timing": "p50 of the wall time per stereo frame at the C ABI, PCIe inside the clock (tools/hostbench)"},
           "protocols_ms": protocols,
           "roofline": None,
           "roofline_note": "a per-call latency configuration: the frame is launch- and PCIe-latency-bound, no kernel runs long "
                            "enough to be priced against a roofline; the batched headline (`python bench.py`) carries it"}
    if not args.no_cpu_baseline:
        try:
            cb = cpu_stereo_baseline(rows, cols, nf, fisheye)
            out["cpu_baseline"] = cb
            out["vs_cpu"] = {"stereo_frame_speedup": cb["ms_per_pair"] / ms, "cpu_protocol": "2 threads (reference)"}
        except (SystemExit, Exception) as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": str(e)}
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


def pcie_inclusive(rows, cols, batch, nfeatures, device=0, as_text=False):
    """Runs tools/hostbench (C++ caller of the C ABI, built by __graft_entry__.build()) on the bench frames and
    returns its JSON object: the PCIe-inclusive rates of the drop-in boundary."""
    exe = os.path.join(ROOT, "tools", "hostbench")
    if not os.path.exists(exe):
        raise SystemExit("tools/hostbench is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as f:
        f.write(bench_frames(rows, cols, batch).tobytes())
        path = f.name
    try:
        out = subprocess.run([exe, path, str(rows), str(cols), str(batch), str(nfeatures), str(device)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    finally:
        os.unlink(path)
    if out.returncode != 0:
        raise SystemExit("tools/hostbench failed (%d): %s" % (out.returncode, out.stderr[-500:]))
    line = out.stdout.strip().splitlines()[-1]
    return line if as_text else json.loads(line)


def settle_together(seconds, body, world, dev):
    """Call body() until `seconds` have passed -- on EVERY rank the same number of times: body() holds a collective
    (the all-gather of the step), so the ranks must not decide by their own clocks.  Returns the number of calls."""
    import torch
    import torch.distributed as dist
    calls = 0
    ts = time.perf_counter()
    while True:
        more = time.perf_counter() - ts < seconds
        if world > 1:
            flag = torch.tensor([1 if more else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item())
        if not more:
            return calls
        body()
        calls += 1


def knn2_roofline(ndist, knn_ms, cap):
    """The brute force's roofs (VERDICT r04 #4a).  Matrix-pipe form (k_bfknn2_frames_mfma, frames of <= 2048 keypoints): per
    distance 256 + 32 i8 MACs (the descriptor's bits + the index block) against the dense i8 MFMA peak, 1024 SIMDs x 32x32x32
    MACs per 32 cycles at 2.4 GHz; vector-pipe form (k_bfknn2_frames, ORBFE_KNN2_MFMA=0 / larger frames): 8 v_xor + 8 v_bcnt per
    64 distances per SIMD at the issue costs of profiles/r01_valu_rate.txt (2.7 / 4.45 cycles)."""
    mfma_on = os.environ.get("ORBFE_KNN2_MFMA", "1") != "0" and cap <= 2048
    i8_peak = 1024 * (32 * 32 * 32 * 2 / 32.0) * 2.4e9 / 1e12  # TOP/s
    vec_peak = 1024 * 64 / (8 * 2.7 + 8 * 4.45) * 2.4e9        # distances/s
    rate = ndist / (knn_ms * 1e-3)
    if mfma_on:
        tops = ndist * 2 * 288 / (knn_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "k_bfknn2_frames_mfma", "achieved": tops, "peak": i8_peak, "unit": "TOP/s",
                "frac": tops / i8_peak, "traffic": None, "algorithmic_ops_per_launch": ndist * 2 * 288,
                "note": "exact: the i32 accumulator of v_mfma_i32_32x32x32_i8 is the sequential scan's key (DESIGN.md 7.5); "
                        "2 x (256 + 32) i8 operations per distance",
                "vector_pipe_peak_distances_per_s": vec_peak, "distances_per_s_over_vector_pipe_peak": rate / vec_peak}, vec_peak
    return {"bound": "valu", "kernel": "k_bfknn2_frames<8>", "achieved": rate, "peak": vec_peak, "unit": "distances/s",
            "frac": rate / vec_peak, "traffic": None}, vec_peak


def pmc_child(args):
    """--pmc-child: what the counter passes profile -- the default step (one lane, rotating resident batches) a few times, then a
    256-MiB device copy whose bytes are known (the calibration of FETCH_SIZE / WRITE_SIZE the guide asks for, in the same pass
    and the same process).  Prints nothing."""
    os.environ.setdefault("ORBFE_TRIG_TABLE", "2")
    import torch
    import orb_slam3_detailed_comments_kor_amd as pkg
    B, H, W = args.batch or 64, args.rows, args.cols
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(bench_frames(H, W, B, 0)).pin_memory().to(dev)
    ex = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    ex.set_lanes(1)
    cap = ex.max_keypoints(H, W)
    R = max(1, min(args.rotate, 12))
    d_rot = [d_img] + [torch.roll(d_img, shifts=(7 * j, 13 * j), dims=(1, 2)).contiguous() for j in range(1, R)]
    o = (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
         torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
    for k in range(2 + 2 * R):  # (the first call builds tables; every batch twice after that)
        ex.extract_batch_device(d_rot[k % R].data_ptr(), B, H, W, W, H * W, (0, 1000), o[0].data_ptr(), o[1].data_ptr(), cap,
                                o[2].data_ptr(), o[3].data_ptr())
    ex.sync()
    x = torch.empty(PMC_CALIB_BYTES // 16, 4, dtype=torch.int32, device=dev)
    x.fill_(3)
    for _ in range(3):
        y = x.clone()  # a vectorised copy kernel: PMC_CALIB_BYTES read, PMC_CALIB_BYTES written
    torch.cuda.synchronize()
    del y
    ex.close()


PMC_CALIB_BYTES = 256 << 20
KERNEL_OF = {"pyramid": "k_pyr_fused", "fast": "k_fast_cells", "octree": "k_octree", "pack": "k_pack",
             "desc": "k_orient_blur_desc<0", "trigfix": "k_orient_blur_desc<1"}


def pmc_measure(args, batch):
    """Two rocprofv3 --pmc passes over `bench.py --pmc-child` (the program itself behind `--`; counters only, beside
    --kernel-trace): FETCH_SIZE + SQ_INSTS_VALU, then WRITE_SIZE.  Returns {kernel name prefix: {hbm_bytes_per_launch,
    SQ_INSTS_VALU, ...}} + the correction factors, or {"error": ...}."""
    import csv
    import glob
    import shutil
    roc = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(roc):
        return {"error": "rocprofv3 not found"}
    # already inside a profiler (somebody ran `rocprofv3 ... -- python3 bench.py`): a second tool library in the children would
    # fight the first one over the counters -- leave the fields empty and say why
    outer = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCTRACER_")) or
             (k == "LD_PRELOAD" and "rocprof" in os.environ[k])]
    if outer:
        return {"error": "running under a profiler (%s): counter passes skipped" % ", ".join(sorted(outer)[:3])}
    tmp = tempfile.mkdtemp(prefix="orbfe_pmc_", dir="/tmp")
    acc, calib = {}, {}
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for tag, ctrs in (("a", ["FETCH_SIZE", "SQ_INSTS_VALU", "SQ_WAVES"]), ("b", ["WRITE_SIZE", "SQ_INSTS_SALU", "SQ_INSTS_LDS"])):
            cmd = [roc, "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", os.path.join(tmp, tag), "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--batch", str(batch), "--rows", str(args.rows), "--cols", str(args.cols),
                   "--nfeatures", str(args.nfeatures), "--rotate", str(args.rotate)]
            # (a pass takes ~2 s; bounded, and the whole process group goes when the bound is hit: the profiled child holds the GPU)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                so, se = pr.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)  # (the group this call started, nothing else)
                except OSError:
                    pass
                pr.communicate()
                return {"error": "rocprofv3 pass %s did not finish in 150 s" % tag}
            if pr.returncode != 0:
                return {"error": "rocprofv3 pass %s failed (rc %d): %s" % (tag, pr.returncode, (se or so)[-300:])}
            rows = {}
            for f in glob.glob(os.path.join(tmp, tag, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    rows.setdefault((name, row["Counter_Name"]), []).append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
            for (name, ctr), v in rows.items():
                v.sort()
                vals = [x for _, x in v]
                if name.startswith("k_"):
                    use = vals[2:] if len(vals) > 4 else vals  # (not the first calls of the process)
                    acc.setdefault(name, {})[ctr] = sum(use) / len(use)
                    acc[name]["dispatches"] = len(use)
                elif ctr in ("FETCH_SIZE", "WRITE_SIZE") and vals and max(vals) * 1024 > 0.3 * PMC_CALIB_BYTES:
                    # the known-size copy: the largest mover of the pass that is not ours (last repetition: warm)
                    big = [x for x in vals if x * 1024 > 0.3 * PMC_CALIB_BYTES]
                    calib[ctr] = PMC_CALIB_BYTES / (1024.0 * big[-1])
        out = {"calibration": {"bytes": PMC_CALIB_BYTES, "fetch_bytes_per_FETCH_SIZE_KB": calib.get("FETCH_SIZE"),
                               "write_bytes_per_WRITE_SIZE_KB": calib.get("WRITE_SIZE")}, "kernels": {}}
        fc, wc = calib.get("FETCH_SIZE"), calib.get("WRITE_SIZE")
        if not fc or not wc or not (0.8 < fc < 2.6) or not (0.8 < wc < 2.6):
            out["error"] = "calibration copy not found or implausible: %r" % (calib,)
            return out
        for name, e in acc.items():
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * fc + e["WRITE_SIZE"] * 1024 * wc
            out["kernels"][name] = e
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        import shutil as _sh
        _sh.rmtree(tmp, ignore_errors=True)


def plan_only(args, world, rank, local_rank, batch_given, real_stdout):
    """--plan: the N-rank launch path without a GPU.  What is rehearsed is exactly what a first run on eight GPUs has never
    executed: RANK / LOCAL_RANK / WORLD_SIZE -> device and shard, the ranks' agreement on the exchange transport (a MIN
    all-reduce: every rank must be able to make its orbfe_mc handle, else all fall back together), one line from rank 0."""
    import torch
    import torch.distributed as dist
    import orb_slam3_detailed_comments_kor_amd as pkg
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames_total = args.batch * world
    first, count = pkg.binding.mc_shard(frames_total, world, rank)  # the library's own shard arithmetic (orbfe_mc_shard)
    # can THIS rank use the C-ABI exchange?  (the library loads, knows the entry points, and finds librccl)
    can = 0
    try:
        L = pkg.lib()
        import ctypes
        can = 1 if (hasattr(L, "orbfe_mc_create") and args.exchange == "cabi") else 0
        if can:
            try:
                ctypes.CDLL("librccl.so.1")
            except OSError:
                try:
                    ctypes.CDLL("/opt/rocm/lib/librccl.so.1")
                except OSError:
                    can = 0
    except Exception:  # noqa: BLE001
        can = 0
    mine = {"rank": rank, "local_rank": local_rank, "device": "cuda:%d" % local_rank, "first_frame": first, "frames": count, "cabi": can}
    if world > 1:
        flag = torch.tensor([can], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        agreed = int(flag.item())
        plans = [None] * world
        dist.all_gather_object(plans, mine)
    else:
        agreed, plans = can, [mine]
    if rank == 0:
        H, W = args.rows, args.cols
        lanes = args.lanes if args.lanes is not None else (3 if args.batch < 16 else 2)
        cap_guess = None
        out = {"metric": "keypoints+descriptors/sec on %dx%dx8-level pyramid" % (W, H), "plan": True, "value": None, "n_gpus": world,
               "scaling": "strong" if args.config == "c4" and not batch_given else "weak",
               "config": {"workload": "%dx%d, nFeatures=%d: %d frames in total, %d per rank%s" % (
                              W, H, args.nfeatures, frames_total, args.batch,
                              " (BASELINE configs[3] sharded over the ranks)" if args.config == "c4" else " (BASELINE configs[1] batched)"),
                          "frames_per_step": frames_total, "frames_per_rank": args.batch, "lanes": lanes,
                          "exchange": ("cabi: one ncclAllGather of %d-frame descriptor slabs per step issued by liborbfe.so" % args.batch)
                                      if agreed and world > 1 else ("torch.distributed all-gather (some rank cannot make the C-ABI handle)"
                                                                  if world > 1 else "none"),
                          "ranks": plans}}
        del cap_guess
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner at communicator
    # creation, for one): until the line is ready, file descriptor 1 points at stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # No collector pauses inside timed loops: with torch loaded one full collection of the interpreter's heap takes 35-65 ms,
    # and it fell once into the cross-camera leg of every 300-step run (0.196 ms per step read as 0.34-0.42; found with
    # ORBFE_BENCH_CROSS_TRACE=1).  The loops below allocate a few small objects per step; nothing here builds cycles.
    gc.disable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 64; with --config c4: 64 / N, "
                    "e.g. --config c4 --batch 8 = the shard ONE of eight ranks runs, timed on one GPU)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=752)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--trig", choices=["libm", "cr", "hostcheck"], default="libm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cross", action="store_true", help="skip the cross-camera matching leg")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive child process (tools/hostbench)")
    ap.add_argument("--settle", type=float, default=0.5,
                    help="keep running untimed steps after the warm-up until this many seconds have passed")
    ap.add_argument("--event-every", type=int, default=0,
                    help="record the per-stage hipEvents on every N-th timed step (a set of records costs ~20 us of stream "
                         "time); 0 = max(6, steps / 20): twenty samples over the default 300 steps")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-context measurement")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2: BASELINE configs[1] batched (752x480, --batch frames per GPU, weak scaling); c4: configs[3], "
                         "64 frames of 1280x720 in total, 64 / N per GPU (strong scaling); c3 / c5: the per-call stereo "
                         "configurations (EuRoC rectified pair nF 1200 / 1024^2 fisheye pair nF 1500) through tools/hostbench")
    ap.add_argument("--exchange", choices=["cabi", "torch"], default="cabi",
                    help="N > 1: the all-gather through the C ABI (orbfe_mc_*: ncclAllGather issued by liborbfe.so on its own "
                         "stream) or through torch.distributed (c10d's process group)")
    ap.add_argument("--lanes", type=int, choices=[1, 2, 3, 4], default=DEFAULT_LANES,
                    help="orbfe_set_lanes for the TIMED region: up to this many batches in flight on streams of the ONE extractor "
                         "context (the latency-bound quadtree kernel and the kernel tails of one batch beside the throughput-bound "
                         "kernels of its neighbours).  The per-kernel times of `roofline` are always measured with one lane, in a "
                         "second region of the same run: overlapped kernels have no per-launch duration")
    ap.add_argument("--rotate", type=int, default=12,
                    help="the timed region rotates through this many DISTINCT resident input batches (and as many output sets as "
                         "there are lanes), so that nothing a step reads or writes is still in the 256-MiB Infinity Cache from the "
                         "previous use (VERDICT r04 weak #5); 1 = the same batch every step, which is also measured and reported "
                         "as `same_batch`")
    ap.add_argument("--input-guard", type=int, choices=[0, 1], default=0,
                    help="orbfe_set_lane_input_guard: 1 = after every call the context's stream waits for the lane's pyramid "
                         "kernel so that the caller may refill the SAME image buffer in stream order (the library's default); "
                         "0 = the images of calls in flight are never rewritten -- true of this bench, whose inputs are resident "
                         "and rotate through --rotate buffers")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process (0 = leave the runtime's default of 4 per priority)")
    ap.add_argument("--plan", action="store_true",
                    help="no GPU work: every rank resolves what it WOULD run (device of its LOCAL_RANK, its shard of the frames, the "
                         "exchange transport all ranks agree on through a MIN all-reduce -- over gloo, so the launcher path of an N-GPU "
                         "run can be rehearsed on a CPU) and rank 0 prints ONE JSON line with the plan (tests/test_bench_plan.py)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes (roofline.traffic / issue_frac: null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--contexts", type=int, default=1,
                    help="experiment: consecutive steps alternate between this many extractor contexts, each with "
                         "its own stream and output buffers (like the reference's left/right extractor threads)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.config in ("c3", "c5"):
        os.environ.setdefault("ORBFE_TRIG_TABLE", "2")
        return per_call_config(args, real_stdout)
    if args.event_every <= 0:
        args.event_every = max(6, args.steps // 20)
    # The same table form at every N (the scaling curve compares like with like): libm's values themselves, 1.03 GB per
    # GPU, expanded on the device from the 65 MB of codes every rank maps from /dev/shm.  The LIBRARY's default for a rank
    # of a multi-process job is the compact form (65 MB, K-DESC 5 us slower per 64 frames); ORBFE_TRIG_TABLE=1 measures it.
    os.environ.setdefault("ORBFE_TRIG_TABLE", "2")
    if args.hw_queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)  # (read by the HIP runtime when it starts: before torch is imported)

    import torch
    import torch.distributed as dist
    import orb_slam3_detailed_comments_kor_amd as pkg
    from orb_slam3_detailed_comments_kor_amd.multicam import CrossCameraMatcher, PipelinedExchange, ring_pairs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 64
    if args.config == "c4":
        args.rows, args.cols = 720, 1280
        if not batch_given:
            if 64 % max(world, 1):
                raise SystemExit("--config c4 shares 64 frames among the ranks: the world size must divide 64 (ADVICE r03)")
            args.batch = max(64 // max(world, 1), 1)
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N`" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.plan:
        return plan_only(args, world, rank, local_rank, batch_given, real_stdout)
    # HBM traffic and wave-instruction counts of THIS tree, by two counter passes over a child of this script -- before this
    # process touches the GPU (the child has the chip to itself; nothing here forks after HIP is initialised)
    pmc_live = None
    if world == 1 and not args.no_pmc:
        tp = time.perf_counter()
        pmc_live = pmc_measure(args, args.batch)
        pmc_live["seconds"] = time.perf_counter() - tp
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("ORBFE_BENCH_FORCE_DIST"):  # the latter: rehearse the N>1 code path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    if args.lanes is None:
        args.lanes = 3 if args.batch < 16 else 2
    B, H, W = args.batch, args.rows, args.cols
    imgs = bench_frames(H, W, B, rank)  # distinct frames per rank
    d_img = torch.from_numpy(imgs).pin_memory().to(dev)

    ex = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank,
                          trig={"libm": pkg.binding.TRIG_LIBM, "cr": pkg.binding.TRIG_CR,
                                "hostcheck": pkg.binding.TRIG_LIBM_HOSTCHECK}[args.trig])
    # One explicit stream for everything: the extractor's kernels, torch's ops and the RCCL collective
    # (which orders itself after the current stream) -- the legacy null stream would not order a
    # non-blocking stream.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    lane_mode = pkg.binding.LANES_BATCH
    ex.set_lanes(args.lanes, lane_mode)
    ex.set_lane_input_guard(args.input_guard)
    cap = ex.max_keypoints(H, W)
    # Rotating inputs: R distinct resident batches -- batch j is batch 0 with every frame shifted cyclically
    # by (7 j, 13 j) pixels, made on the device: other bytes at other addresses, the same corner statistics -- so that a step's
    # images, pyramids and outputs have been pushed out of the 256-MiB Infinity Cache by the steps in between.
    R = max(1, args.rotate)
    d_rot = [d_img] + [torch.roll(d_img, shifts=(7 * j, 13 * j), dims=(1, 2)).contiguous() for j in range(1, R)]
    rot_on = [True]
    # two slab pairs: the all-gather of batch i (process group's stream) overlaps the extraction of batch i+1
    pipe = PipelinedExchange(B, cap, dev, world, rank)
    d_desc = pipe.x[0].desc_view()
    d_n = pipe.x[0].count_view()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
    lap = (0, 1000)  # mono protocol, src/Frame.cc:306
    # batch lanes: calls in flight together need output sets of their own (include/orbfe.h) -- a ring of `lanes` sets
    nring = max(args.lanes, 1)
    ring = [(torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
             torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(nring)]

    # N > 1 (or the one-GPU rehearsal): the sharded extraction + exchange behind the C ABI (include/orbfe_mc.h).  The id of
    # the RCCL communicator travels through torch.distributed's store; if the handle cannot be made on EVERY rank the run
    # falls back to the c10d all-gather (and says so in the line).
    mc = None
    mc_note = None
    if dist.is_initialized() and args.exchange == "cabi":
        try:
            box = [pkg.binding.mc_unique_id(pkg.binding.MC_RCCL) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            mc = pkg.binding.MultiCam(ex, box[0], rank, world, B, cap, pkg.binding.MC_RCCL)
        except Exception as e:  # noqa: BLE001
            mc_note = "%s: %s" % (type(e).__name__, e)
            mc = None
        flag = torch.tensor([1 if mc is not None else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and mc is not None:
            mc.close()
            mc = None
            mc_note = "another rank could not create its handle"
    mc_inflight = [0]
    mc_last = [None]

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

    used = []  # the input batch of every step since the last reset (the keypoints of a region = sum of its batches' counts)

    def step():
        k = counter[0] % (1 + len(extra))
        j = (counter[0] % R) if rot_on[0] else 0
        d_img = d_rot[j]
        used.append(j)
        counter[0] += 1
        if k > 0:
            e2, _, k2, de2, n2, m2 = extra[k - 1]
            e2.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, k2.data_ptr(), de2.data_ptr(), cap,
                                    n2.data_ptr(), m2.data_ptr())
            return
        if not dist.is_initialized():  # one GPU, no exchange: the ring of output sets (one per lane)
            o = ring[counter[0] % nring]
            ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, o[0].data_ptr(), o[1].data_ptr(), cap,
                                    o[2].data_ptr(), o[3].data_ptr())
            return
        if mc is not None:
            # C ABI: extraction into the next slab + ncclAllGather on the library's side stream; two batches in flight
            # (the host waits for batch i-1's collective while batch i is already queued)
            if mc_inflight[0] == pkg.binding.MC_MAX_IN_FLIGHT:  # (three batches in flight: include/orbfe_mc.h)
                mc_last[0] = mc.wait()
                mc_inflight[0] -= 1
            mc.submit(d_img.data_ptr(), H, W, W, H * W, lap)
            mc_inflight[0] += 1
            return
        x = pipe.begin()  # waits (on the stream) for the collective that last read this slab
        ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, d_kps.data_ptr(), x.desc_view().data_ptr(), cap,
                                x.count_view().data_ptr(), d_mono.data_ptr())
        if dist.is_initialized():
            ex.lanes_join()  # (the collective orders itself after THIS stream: the second lane's half must be behind it)
            pipe.submit()  # one RCCL all-gather of descriptor slabs per batch, asynchronous

    def barrier():
        while mc_inflight[0] > 0:
            mc_last[0] = mc.wait()
            mc_inflight[0] -= 1
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the first extraction of the process builds and uploads the libm trig table and allocates every buffer
    tf = time.perf_counter()
    step()
    barrier()
    first_call_ms = 1e3 * (time.perf_counter() - tf)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    barrier()
    # clock settling by TIME, not by step count: keep stepping (untimed) until --settle seconds have passed
    def eight_steps():
        for _ in range(8):
            step()
        torch.cuda.synchronize()

    # keypoints of every input batch (one plain extraction each, outside every clock and BEFORE the settling: the host
    # round trips of this loop let the clocks drop)
    kp_of, n_of = [], []
    for j in range(R):
        ex.extract_batch_device(d_rot[j].data_ptr(), B, H, W, W, H * W, lap, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_n.data_ptr(), d_mono.data_ptr())
        ex.sync()
        kp_of.append(int(d_n.sum().item()))
        n_of.append(d_n.to(torch.float64).clone())
    barrier()
    # `same_batch` (round 4's measurement, reported beside `value`): the same step on ONE batch, cache-resident between steps
    same_batch = None
    settle_steps = 0
    if R > 1:
        rot_on[0] = False
        settle_steps += 8 * settle_together(args.settle, eight_steps, world, dev)
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        ts = time.perf_counter() - ts
        same_batch = {"ms_per_step": 1e3 * ts / args.steps, "value_this_rank": kp_of[0] * args.steps / ts, "unit": "keypoints/s",
                      "note": "every step re-reads ONE resident batch (images + pyramids + outputs < 256 MiB: mostly Infinity "
                              "Cache hits); `value` rotates through %d batches" % R}
        rot_on[0] = True
    settle_steps += 8 * settle_together(args.settle, eight_steps, world, dev)
    barrier()
    del used[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kp_timed = sum(kp_of[j] for j in used)
    # Per-kernel launch durations for `roofline`: hipEvents between the kernels on the stream they run on, in a SECOND
    # region of the same run with ONE lane (orbfe_set_lanes(1)), at least 60 steps, every event_every-th of them recording
    # one set of stage events.  With two lanes the kernels of the two half-batches overlap on the GPU, so a launch has a
    # wall-clock duration but no throughput of its own; the one-lane durations are what `rocprofv3 --kernel-trace --stats`
    # of `bench.py --lanes 1` reports (profiles/).
    ex.set_lanes(1)
    barrier()
    roof_steps = max(args.steps, 60)
    ex.profile(args.event_every)
    t1l = time.perf_counter()
    for _ in range(roof_steps):
        step()  # every event_every-th call records one set of stage events on the stream (no extra sync)
    barrier()
    one_lane_ms = 1e3 * (time.perf_counter() - t1l) / roof_steps
    stage_ms = ex.stage_ms()  # hipEvent times averaged over the sampled steps
    ex.profile(False)
    ex.set_lanes(args.lanes, lane_mode)
    barrier()
    # Not part of `value`: the same step K more times with one event per step boundary on the stream (an event record
    # costs ~3.5 us, which is why the timed region above carries none): the distribution a single average hides.
    step_dist = None
    if not extra and args.lanes == 1:  # (with lanes the context's stream carries only the ordering: an event there marks nothing)
        nd = min(max(args.steps, 20), 400)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nd + 1)]
        evs[0].record(stream)
        for i in range(nd):
            step()
            evs[i + 1].record(stream)
        barrier()
        per = np.sort(np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(nd)]))
        step_dist = {"steps": nd, "min_ms": float(per[0]), "p50_ms": float(per[nd // 2]), "p90_ms": float(per[(9 * nd) // 10]),
                     "max_ms": float(per[-1]), "note": "hipEvent between consecutive steps on the extractor's stream (one lane); each "
                     "step carries one event record (~3.5 us), so these read slightly above ms_per_step"}

    # Not part of `value`: the step followed by the consumer of the exchanged descriptors -- cross-camera matching,
    # sharded by query frame: knn-2 of each of this rank's frames against the next camera of the ring
    # (global frame g+1; the last local frame's partner lives on the next rank), read from the gathered buffer in
    # place, one launch per batch.  The match of batch i-1 is queued behind the extraction of batch i, so it overlaps
    # batch i's all-gather.  Runs at every N (at N=1 the "gather" is the local slab copy).
    cross = None
    # World 1 without a process group: the same leg through the C ABI (orbfe_mc_* with a world of one: the "gather" is a device
    # copy on the handle's side stream), so that the matching rides on the batch lanes like the N > 1 path does (round 4 ran
    # it through the torch classes, whose local copy needs the lanes joined every step)
    mcx, mcl = mc, None
    if mcx is None and world == 1 and not dist.is_initialized() and not extra and not args.no_cross:
        try:
            mcl = pkg.binding.MultiCam(ex, pkg.binding.mc_unique_id(pkg.binding.MC_HOST), 0, 1, B, cap, pkg.binding.MC_HOST)
            mcx = mcl
        except Exception:  # noqa: BLE001
            mcx = mcl = None
    if mcx is not None and not args.no_cross:
        try:
            hops = (1,)
            fifo = []      # input batch of every submit in flight
            last = [None]  # (view, input batch) of the newest completed exchange

            def step_cross_mc():
                j = (counter[0] % R) if rot_on[0] else 0
                counter[0] += 1
                v = None
                t0 = time.perf_counter()
                if mc_inflight[0] == pkg.binding.MC_MAX_IN_FLIGHT:
                    v = mcx.wait()
                    mc_inflight[0] -= 1
                    last[0] = (v, fifo.pop(0))
                t1 = time.perf_counter()
                mcx.submit(d_rot[j].data_ptr(), H, W, W, H * W, lap)
                t2 = time.perf_counter()
                fifo.append(j)
                mc_inflight[0] += 1
                # the matching of the batch that has just arrived goes behind the next submit: the lane of that submit starts at
                # the point of the call on this stream, so it does not queue behind a matching kernel that shares the CUs with
                # two extractions (70-210 us per launch there against 18-22 alone)
                if v is not None:
                    mcx.match_ring_async(v.batch, hops)
                t3 = time.perf_counter()
                worst[0], worst[1], worst[2] = max(worst[0], t1 - t0), max(worst[1], t2 - t1), max(worst[2], t3 - t2)

            worst = [0.0, 0.0, 0.0]  # (diagnostic: the longest wait / submit / match call of the leg)

            def drain():
                while mc_inflight[0] > 0:
                    last[0] = (mcx.wait(), fifo.pop(0))
                    mc_inflight[0] -= 1
                barrier()

            for _ in range(10):
                step_cross_mc()
            drain()
            stamps = [] if os.environ.get("ORBFE_BENCH_CROSS_TRACE") else None
            worst[:] = [0.0, 0.0, 0.0]
            tc = time.perf_counter()
            for _ in range(args.steps):
                step_cross_mc()
                if stamps is not None:
                    stamps.append(time.perf_counter())
            drain()
            tc = time.perf_counter() - tc
            if stamps is not None:
                print("cross worst calls ms: wait %.3f submit %.3f match %.3f" % tuple(1e3 * w for w in worst), file=sys.stderr)
            if stamps is not None:  # (diagnostic: host time of every 10 steps of the leg, to stderr)
                print("cross trace (ms per step over 10 steps):", " ".join("%.3f" % (1e2 * (stamps[i + 10] - stamps[i]))
                                                                            for i in range(0, len(stamps) - 10, 10)), file=sys.stderr)
            cross = {"jobs_per_step": B * world, "pairing": "every frame against the next camera of the ring (global frame g+1), "
                     "train frames read from the gathered buffer in place (orbfe_mc_match_ring_async)",
                     "ms_per_step": 1e3 * tc / args.steps}
            if world == 1 and last[0] is not None:
                # the matching launch alone, by events on the stream it runs on (the handle's buffers of the last batch)
                v, jb = last[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                mcx.match_ring_async(v.batch, hops)
                e0.record()
                for _ in range(20):
                    mcx.match_ring_async(v.batch, hops)
                e1.record()
                torch.cuda.synchronize()
                knn_ms = e0.elapsed_time(e1) / 20
                cnt = n_of[jb]
                ndist = float((cnt * torch.roll(cnt, -1)).sum().item())
                cross.update({"knn2_launch_ms": knn_ms, "distances_per_launch": ndist, "distances_per_s": ndist / (knn_ms * 1e-3)})
                cross["roofline"], vec_peak = knn2_roofline(ndist, knn_ms, cap)
                # ... and the single-problem kernel at BASELINE configs[4]'s size (orbfe_bfknn2 inside
                # Frame::ComputeStereoFishEyeMatches: 1500 x 1500, one wavefront per query)
                nb = 1500
                dq = torch.randint(0, 256, (nb, 32), dtype=torch.uint8, device=dev)
                dt2 = torch.randint(0, 256, (nb, 32), dtype=torch.uint8, device=dev)
                di = torch.zeros((nb, 2), dtype=torch.int32, device=dev)
                dd = torch.zeros((nb, 2), dtype=torch.int32, device=dev)
                for _ in range(3):
                    pkg.binding.bfknn2_device(dq.data_ptr(), nb, dt2.data_ptr(), nb, di.data_ptr(), dd.data_ptr(), stream=stream.cuda_stream)
                e0.record()
                for _ in range(20):
                    pkg.binding.bfknn2_device(dq.data_ptr(), nb, dt2.data_ptr(), nb, di.data_ptr(), dd.data_ptr(), stream=stream.cuda_stream)
                e1.record()
                torch.cuda.synchronize()
                ms1 = e0.elapsed_time(e1) / 20
                cross["bfknn2_1500x1500"] = {"kernel": "k_bfknn2", "launch_ms": ms1, "distances_per_s": nb * nb / (ms1 * 1e-3),
                                             "roofline": {"bound": "valu", "achieved": nb * nb / (ms1 * 1e-3), "peak": vec_peak,
                                                          "unit": "distances/s", "frac": nb * nb / (ms1 * 1e-3) / vec_peak,
                                                          "note": "2.25 M distances are 0.8 us of popcount issue: the launch is its "
                                                                  "own latency (1500 wavefronts of 24 dependent iterations)"}}
        except Exception as e:  # noqa: BLE001
            cross = {"error": "%s: %s" % (type(e).__name__, e)}
        if mcl is not None:
            try:
                barrier()
                mcl.close()
            except Exception:  # noqa: BLE001
                pass
    elif not extra and not args.no_cross:
        try:  # (a secondary leg: if it fails, the line still carries `value` and says why this object is missing)
            cm = CrossCameraMatcher(pipe.x, ring_pairs(world, B, rank), dev)
            prev = [None]

            def step_cross():
                x = pipe.begin()
                ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, d_kps.data_ptr(), x.desc_view().data_ptr(),
                                        cap, x.count_view().data_ptr(), d_mono.data_ptr())
                k = pipe.i % len(pipe.x)
                ex.lanes_join()  # (the copy below runs on the context's stream: behind the lane that holds this batch)
                pipe.submit()  # world 1 without a process group: the local copy into the gathered buffer
                if prev[0] is not None:
                    kp, xp = prev[0]
                    if pipe.pending[kp] is not None:
                        pipe.pending[kp].wait()  # the stream waits for batch i-1's collective; begin() clears the handle
                    cm.match(xp)
                prev[0] = (k, x)

            for _ in range(10):
                step_cross()
            barrier()
            tc = time.perf_counter()
            for _ in range(args.steps):
                step_cross()
            barrier()
            tc = time.perf_counter() - tc
            # the matching launch alone, by events on the stream it runs on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            xl = pipe.completed()
            cm.match(xl)
            e0.record()
            for _ in range(20):
                cm.match(xl)
            e1.record()
            torch.cuda.synchronize()
            knn_ms = e0.elapsed_time(e1) / 20
            cnt = xl.count_view().to(torch.float64)
            tcount = torch.stack([xl.unpack(g // B)[0][g % B] for _, g in cm.pairs]).to(torch.float64)
            ndist = float((cnt * tcount).sum().item())
            good = cm.dist[:, :, 0].to(torch.float64) < 0.7 * cm.dist[:, :, 1].to(torch.float64)
            valid = torch.arange(cap, device=dev)[None, :] < xl.count_view()[:, None]
            cross = {"jobs_per_step": cm.njobs * world, "pairing": "every frame against the next camera of the ring (global "
                     "frame g+1), train frames read from the gathered buffer in place",
                     "ms_per_step": 1e3 * tc / args.steps, "knn2_launch_ms": knn_ms,
                     "distances_per_launch": ndist, "distances_per_s": ndist / (knn_ms * 1e-3),
                     "ratio_test_survivors_per_frame": float((good & valid).sum().item()) / max(cm.njobs, 1)}
            cross["roofline"], vec_peak = knn2_roofline(ndist, knn_ms, cap)
            # ... and the single-problem kernel at BASELINE configs[4]'s size (orbfe_bfknn2 inside
            # Frame::ComputeStereoFishEyeMatches: 1500 x 1500, one wavefront per query)
            try:
                nb = 1500
                dq = torch.randint(0, 256, (nb, 32), dtype=torch.uint8, device=dev)
                dt2 = torch.randint(0, 256, (nb, 32), dtype=torch.uint8, device=dev)
                di = torch.zeros((nb, 2), dtype=torch.int32, device=dev)
                dd = torch.zeros((nb, 2), dtype=torch.int32, device=dev)
                for _ in range(3):
                    pkg.binding.bfknn2_device(dq.data_ptr(), nb, dt2.data_ptr(), nb, di.data_ptr(), dd.data_ptr(), stream=stream.cuda_stream)
                e0.record()
                for _ in range(20):
                    pkg.binding.bfknn2_device(dq.data_ptr(), nb, dt2.data_ptr(), nb, di.data_ptr(), dd.data_ptr(), stream=stream.cuda_stream)
                e1.record()
                torch.cuda.synchronize()
                ms1 = e0.elapsed_time(e1) / 20
                cross["bfknn2_1500x1500"] = {"kernel": "k_bfknn2", "launch_ms": ms1, "distances_per_s": nb * nb / (ms1 * 1e-3),
                                             "roofline": {"bound": "valu", "achieved": nb * nb / (ms1 * 1e-3), "peak": vec_peak,
                                                          "unit": "distances/s", "frac": nb * nb / (ms1 * 1e-3) / vec_peak,
                                                          "note": "2.25 M distances are 0.8 us of popcount issue: the launch is its "
                                                                  "own latency (1500 wavefronts of 24 dependent iterations)"}}
            except Exception as e:  # noqa: BLE001
                cross["bfknn2_1500x1500"] = {"error": "%s: %s" % (type(e).__name__, e)}
            prev[0] = None
        except Exception as e:  # noqa: BLE001
            cross = {"error": "%s: %s" % (type(e).__name__, e)}

    # Not part of `value`: the same batches alternating between TWO extractor contexts on two streams (how
    # a multi-camera rig drives one extractor per camera, reference src/Frame.cc:119-122).  Consecutive
    # batches then overlap on the GPU, which hides the latency-bound stages (octree, pack) and launch gaps.
    pipelined = None
    if world == 1 and not extra and not args.no_pipelined:
        e2 = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank, trig=ex.trig)
        s2 = torch.cuda.Stream(device=dev)
        e2.set_stream(s2.cuda_stream)
        ex.set_lanes(1)  # (this leg: two contexts with one lane each)
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
        pipelined = {"contexts": 2, "lanes_per_context": 1, "ms_per_step": 1e3 * tp / args.steps,
                     "value": float(d_n.sum().item()) * args.steps / tp, "unit": "keypoints/s"}
        e2.close()
        ex.set_lanes(args.lanes, lane_mode)

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

        def timed1():
            for _ in range(20):
                step1()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step1()
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        ex.set_lanes(1)  # ms_per_frame is the one-stream figure (calls back to back on one stream: the chain of four kernels)
        t1 = timed1()
        single = {"frames_per_call": 1, "ms_per_frame": 1e3 * t1 / args.steps,
                  "value": float(n1.item()) * args.steps / t1, "unit": "keypoints/s"}
        if args.lanes > 1:  # ... and the same calls dealt to the lanes (frames of several cameras / trackers in flight together)
            ex.set_lanes(args.lanes, lane_mode)
            tl = timed1()
            single["ms_per_frame_lanes"] = 1e3 * tl / args.steps
            single["lanes"] = args.lanes

    n_local = kp_timed / max(args.steps, 1)  # keypoints of this rank's average step
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
        # HBM bytes per launch and vector wave-instructions per launch: measured for this tree by pmc_measure() above
        traffic = None
        per_kernel = {}
        kernel_of = KERNEL_OF
        issue = None
        traffic_source = None
        pk = (pmc_live or {}).get("kernels") or {}
        if pmc_live is not None and "error" not in pmc_live:
            traffic_source = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_WAVES / --pmc WRITE_SIZE ... over "
                              "`bench.py --pmc-child` (same batches, one lane), bytes per FETCH_SIZE / WRITE_SIZE unit from a %d-MiB "
                              "device copy in the same passes (%.3f / %.3f KB); %.0f s"
                              % (PMC_CALIB_BYTES >> 20, pmc_live["calibration"]["fetch_bytes_per_FETCH_SIZE_KB"],
                                 pmc_live["calibration"]["write_bytes_per_WRITE_SIZE_KB"], pmc_live.get("seconds", 0.0)))
            for st, kn in kernel_of.items():
                for k, e in pk.items():
                    if k.startswith(kn) and "hbm_bytes_per_launch" in e:
                        per_kernel[st] = e["hbm_bytes_per_launch"]
            traffic = per_kernel.get(dom)
        elif pmc_live is not None:
            traffic_source = "not measured: " + str(pmc_live.get("error"))
        # the vector-issue roof, ONE definition (tools/isa/issue_table.py): wave-instructions counted in this run x the static-mix
        # mean of the per-opcode issue costs measured on this chip / (1024 SIMDs x 2.4 GHz x this run's launch duration)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools", "isa"))
            import issue_table as IT
            try:
                table = IT.issue_table()
                table_src = "llvm-objdump of the library this run loaded"
            except Exception:  # noqa: BLE001 (no llvm-objdump on this host: the committed table of the same sources)
                table = json.load(open(os.path.join(ROOT, "profiles", "r06_issue_table.json")))
                table_src = "profiles/r06_issue_table.json"
            issue_all = {}
            for st, kn in kernel_of.items():
                ms = stage_ms.get(st, 0.0)
                ek = next((e for k, e in pk.items() if k.startswith(kn) and "SQ_INSTS_VALU" in e), None)
                tk = next((t for k, t in table["kernels"].items() if k.startswith(kn) and (ek is None or k in pk)), None) or \
                    next((t for k, t in table["kernels"].items() if k.startswith(kn)), None)
                if ek is None or tk is None or ms <= 0:
                    continue
                cyc = ek["SQ_INSTS_VALU"] * tk["cycles_per_instruction"]
                issue_all[st] = {"wave_instructions": ek["SQ_INSTS_VALU"], "cycles_per_instruction": tk["cycles_per_instruction"],
                                 "issue_us": cyc / (1024 * 2.4e3), "issue_frac": cyc / (1024 * 2.4e3) / (ms * 1e3)}
            if dom in issue_all:
                issue = dict(issue_all[dom], table=table_src, definition=table["definition"], kernels=issue_all)
        except Exception as e:  # noqa: BLE001
            issue = {"error": "%s: %s" % (type(e).__name__, e)}
        step_bytes = sum(abytes.values()) * B
        out = {
            "metric": "keypoints+descriptors/sec on %dx%dx8-level pyramid" % (W, H),
            "value": kp_per_step * args.steps / dt,
            "unit": "keypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.config == "c4" and not batch_given else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d grayscale, 8-level pyramid, nFeatures=%d, FAST 20/7, %d frames/GPU/step "
                            "resident in HBM, rotating %d distinct input batches (%.0f MiB of images, %.0f MiB of pyramids and "
                            "outputs per lane touched between two uses of a batch), outputs left in HBM (%s); the "
                            "host-pointer rate of the same workload (H2D + D2H inside the clock) is `boundary_value`"
                            % (W, H, args.nfeatures, B, R, R * B * H * W / 2.0 ** 20,
                               B * (3.096 * 1.4 * H * W + cap * 60) / 2.0 ** 20, ("BASELINE configs[3]: 64 frames in total, sharded over the ranks" if not batch_given
                                                         else "BASELINE configs[3]'s frame size, %d frames per GPU: the shard one of %d "
                                                         "ranks runs" % (B, max(64 // B, 1)))
                               if args.config == "c4" else "BASELINE configs[1] batched"),
                "frames_per_step": B * world,
                "keypoints_per_step": kp_per_step,
                "trig": args.trig,
                "trig_table": {"2": "full (1.03 GB per GPU, expanded on the device from the shared 65 MB of codes)",
                               "1": "compact (65 MB of codes)", "0": "none (host check)"}.get(
                                   os.environ.get("ORBFE_TRIG_TABLE", ""), os.environ.get("ORBFE_TRIG_TABLE", "")),
                "contexts": 1 + len(extra),
                "lanes": args.lanes,
                "rotate": R,
                "input_guard": bool(args.input_guard),
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "lanes_note": ("whole batches dealt round-robin to %d streams of the one extractor context (orbfe_set_lanes), a ring of %d "
                               "output sets: identical outputs, complete when the timed region's closing synchronisation returns"
                               % (args.lanes, nring) if args.lanes >= 2 else "one stream"),
                "exchange": (("1 ncclAllGather of descriptor slabs per step issued by liborbfe.so (orbfe_mc_extract_exchange_"
                              "submit / _wait) on its own stream, overlapped with the next step's extraction" if mc is not None
                              else "1 all-gather of descriptor slabs per step through torch.distributed, overlapped with the "
                              "next step's extraction" + (" (C-ABI handle unavailable: %s)" % mc_note if mc_note else ""))
                             if dist.is_initialized() else "none"),
            },
            "roofline": {
                # the contract's accounting: algorithmic bytes against the HBM roof (achieved / peak / frac).  What actually limits
                # the kernel is named in `limiter`: its vector ALUs are busier than its memory system by far (`issue_frac`)
                "bound": "hbm",
                "limiter": ("valu-issue" if (issue and issue.get("issue_frac", 0) > achieved / HBM_PEAK_GBPS) else "hbm"),
                "issue_frac": issue.get("issue_frac") if issue else None,
                "issue": issue,
                "kernel": kernel_of[dom],
                "stage": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_over_algorithmic": (traffic / launch_bytes) if (traffic and launch_bytes) else None,
                "algorithmic_bytes_per_launch": launch_bytes,
                "event_sampling": "stage hipEvents on every %d-th of the timed steps; the sampled steps carry six event "
                                  "records, so the stage times add up to a few per cent more than ms_per_step"
                                  % max(args.event_every, 1),
                "avg_launch_ms": stage_ms[dom],
                "measured_with": "one lane (orbfe_set_lanes(1)), %d steps right after the timed region of the same run; "
                                 "ms_per_step of that region: %.5f" % (roof_steps, one_lane_ms),
                "one_lane_ms_per_step": one_lane_ms,
                "stage_ms": stage_ms,
                # the same figures for every kernel of the step (the two largest are within a few per cent of
                # each other, so which one is "dominant" can change from run to run) and for the whole step
                "kernels": {st: {"kernel": kernel_of[st], "avg_launch_ms": ms,
                                 "algorithmic_bytes_per_launch": abytes[st] * B,
                                 "achieved": abytes[st] * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                 "frac": (abytes[st] * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ms > 0 else 0.0,
                                 "traffic": per_kernel.get(st)}
                            for st, ms in stage_ms.items() if st != "trigfix"},
                # the whole step against the same roof: SURVEY 8(d)'s bytes per step / ms_per_step (all lanes) / 8 TB/s, and the
                # measured traffic of its kernels the same way
                "step": {"algorithmic_bytes": step_bytes,
                         "achieved": step_bytes / dt * args.steps / 1e9,
                         "frac": step_bytes / dt * args.steps / 1e9 / HBM_PEAK_GBPS,
                         "traffic": sum(per_kernel.values()) if per_kernel else None,
                         "traffic_frac": (sum(per_kernel.values()) / dt * args.steps / 1e9 / HBM_PEAK_GBPS) if per_kernel else None,
                         # the step against the vector-issue roof: the issue time of ALL its kernels (same definition as
                         # issue_frac, counts of this run) over the wall time of a step with the lanes overlapping them --
                         # what is left of this is all that a better overlap could still buy
                         "issue_us": (sum(e["issue_us"] for e in issue["kernels"].values())
                                      if issue and issue.get("kernels") and world == 1 else None),
                         "issue_frac": (sum(e["issue_us"] for e in issue["kernels"].values()) / (1e6 * dt / args.steps)
                                        if issue and issue.get("kernels") and world == 1 else None)},
            },
        }
        if same_batch is not None:
            out["same_batch"] = same_batch
        if pipelined is not None:
            out["pipelined"] = pipelined
        if single is not None:
            out["single_frame"] = single
        if cross is not None:
            out["cross_camera"] = cross
        if step_dist is not None:
            out["step_ms_dist"] = step_dist
        out["first_call_ms"] = first_call_ms
        if world == 1 and not args.no_pcie:
            torch.cuda.synchronize()
            try:  # (a secondary leg measured by a child process: its failure must not take `value` down)
                out["pcie_inclusive"] = pcie_inclusive(H, W, B, args.nfeatures, local_rank)
            except (SystemExit, Exception) as e:  # noqa: BLE001
                out["pcie_inclusive"] = {"error": str(e)}
            try:
                out["startup"] = startup_cost(H, W, args.nfeatures, local_rank)
            except (SystemExit, Exception) as e:  # noqa: BLE001
                out["startup"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(H, W, args.nfeatures)
            except (SystemExit, Exception) as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(e)}
        # the SURVEY 8(d) metric (host pointers, H2D + D2H inside the clock) and both rates against the CPU path, as
        # top-level fields next to `value` (`vs_baseline` stays null: BASELINE.md holds no published number)
        pi = out.get("pcie_inclusive") or {}
        bp = (pi.get("batch_pipelined") or {}) if isinstance(pi, dict) else {}
        if "keypoints_per_s" in bp:
            out["boundary_value"] = bp["keypoints_per_s"]
            out["boundary_definition"] = ("orbfe_extract_batch_submit/_wait with host pointers, two pinned batches in flight, "
                                          "H2D of the images and D2H of keypoints + descriptors inside the clock (SURVEY.md 8d)")
        cb = out.get("cpu_baseline") or {}
        if isinstance(cb, dict) and cb.get("value"):
            out["vs_cpu"] = {"value_over_cpu_all_cores": out["value"] / cb["value"],
                             "value_over_cpu_one_thread": out["value"] / cb["one_thread"]["value"],
                             "cpu_cores": cb["cores"], "cpu": "vectorised timing build (cpu_baseline.variant)"}
            if isinstance(cb.get("scalar_port"), dict):  # (the ratio rounds 1-5 quoted: against the scalar parity oracle)
                out["vs_cpu"]["value_over_scalar_port_all_cores"] = out["value"] / cb["scalar_port"]["value"]
                out["vs_cpu"]["value_over_scalar_port_one_thread"] = out["value"] / cb["scalar_port"]["one_thread"]["value"]
            if "boundary_value" in out:
                out["vs_cpu"]["boundary_over_cpu_all_cores"] = out["boundary_value"] / cb["value"]
                out["vs_cpu"]["boundary_over_cpu_one_thread"] = out["boundary_value"] / cb["one_thread"]["value"]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)  # (whatever the teardown prints is not part of the line either)
    if mc is not None:
        barrier()
        mc.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
