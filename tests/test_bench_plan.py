"""bench.py's N-rank launch path rehearsed without a GPU: `bench.py --gpus 8 --config c4 --plan` under the
driver's launcher (torch.distributed.run, 8 processes, gloo) resolves LOCAL_RANK -> device and the shard of every rank with
the library's own arithmetic, agrees on the exchange transport with a MIN all-reduce, and prints ONE line from rank 0 whose
config says 64 frames in total, 8 per rank."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--plan"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE line, from rank 0
    return json.loads(lines[0])


def test_eight_ranks_c4_plan():
    j = _run(8, ["--config", "c4"])
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong"
    assert c["frames_per_step"] == 64 and c["frames_per_rank"] == 8 and "64 frames in total, 8 per rank" in c["workload"]
    ranks = sorted(c["ranks"], key=lambda p: p["rank"])
    assert [p["rank"] for p in ranks] == list(range(8))
    assert [p["device"] for p in ranks] == ["cuda:%d" % i for i in range(8)]      # LOCAL_RANK -> device
    assert [p["first_frame"] for p in ranks] == [8 * i for i in range(8)] and all(p["frames"] == 8 for p in ranks)
    assert len({p["cabi"] for p in ranks}) == 1                                    # every rank saw the same library
    assert ("ncclAllGather" in c["exchange"]) == bool(ranks[0]["cabi"])
    assert c["lanes"] == 3                                                          # < 16 frames per rank: three batch lanes


def test_two_ranks_default_config_plan_and_forced_torch_exchange():
    j = _run(2, ["--exchange", "torch"])
    c = j["config"]
    assert j["scaling"] == "weak" and c["frames_per_step"] == 128 and c["frames_per_rank"] == 64
    assert all(p["cabi"] == 0 for p in c["ranks"]) and "torch.distributed" in c["exchange"]


def test_counter_passes_are_skipped_under_an_outer_profiler(monkeypatch):
    # `rocprofv3 ... -- python3 bench.py`: the children of bench.py's own counter passes would load a second tool library into a
    # process that already has one.  bench.py then leaves traffic / issue_frac empty and says why (no GPU needed: it returns
    # before anything is started).
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    args = types.SimpleNamespace(rows=480, cols=752, nfeatures=1000, rotate=12)
    r = bench.pmc_measure(args, 64)
    if "rocprofv3 not found" in r.get("error", ""):
        return  # (an image without the profiler: the earlier exit)
    assert "error" in r and "under a profiler" in r["error"], r
