"""bench.py's harness on the GPU box: the N > 1 path (self-spawned ranks, collation, one JSON line) exercised with two
ranks sharing the box's single device over gloo, and the default line's contract keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,scaling", [("cfg2", "weak"), ("cfg4", "strong")])
def test_two_ranks_on_one_device(workload, scaling):
    """`bench.py --gpus 2` spawns its own two ranks (no launcher); both share device 0 and talk over gloo here, RCCL on a
    multi-GPU node.  cfg2: weak scaling, every rank its own complexes; cfg4: the split LPT-sharded over the ranks."""
    extra = ["--split-limit", "60", "--min-seconds", "0.3"] if workload == "cfg4" else []
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline", "--no-gather",
                "--no-secondary", "--no-pmc"] + extra, {"NAMP_BENCH_ONE_DEVICE": "1", "NAMP_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["warmup"] == 1
    assert out["scaling"] == scaling and out["value"] > 0 and out["higher_is_better"] is True
    col = out["collation"]
    assert col["ranks"] == 2 and col["backend"] == "gloo"
    if workload == "cfg2":
        assert out["steps"] == 3
        assert col["collated_residues"] == col["expected_residues"] == 2 * 1000
        assert out["config"]["global_batch"] == 2
        assert out["dtype"] == "f32" and out["x3"]["value"] > 0            # both evaluations are in the line on every N
    else:
        sh = out["shard"]
        # the pass count is scaled until the timed region lasts >= --min-seconds on the slowest rank
        assert out["steps"] >= 3 and sh["steps_requested"] == 3
        # (one calibration pass sets the count: two ranks sharing one device — and the first passes on a fresh box — jitter by tens of per cent)
        assert out["steps"] * out["ms_per_step"] >= 0.3e3 * 0.6, (out["steps"], out["ms_per_step"], sh)
        assert col["residues_collated"] > 0
        assert sh["rank0_residues"] < col["residues_collated"]   # the other rank really took a share
        pr, ps = sh["per_rank_residues"], sh["per_rank_seconds"]
        assert pr["min"] <= pr["mean"] <= pr["max"] and 0 < ps["min"] <= ps["mean"] <= ps["max"]
        assert abs(pr["mean"] * 2 - col["residues_collated"]) < 1.0
        assert 1.0 <= sh["lpt_imbalance"] < 1.1, sh


def test_two_ranks_emit_both_scaling_points():
    """`bench.py --gpus 2` (the driver's SCALE shape, no --no-secondary): ONE invocation yields the cfg2 weak-scaling value AND a
    `cfg4_strong` sub-object — a timed region of passes over the LPT-sharded split with per-rank min / mean / max."""
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-gather", "--no-pmc", "--split-limit", "60",
                "--min-seconds", "0.3"], {"NAMP_BENCH_ONE_DEVICE": "1", "NAMP_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    c4 = out["cfg4_strong"]
    assert "error" not in c4, c4
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["value"] > 0 and c4["backend"] == "gloo"
    assert c4["passes"] * c4["ms_per_pass"] >= 0.3e3 * 0.6, c4
    ps = c4["per_rank_seconds"]
    assert 0 < ps["min"] <= ps["mean"] <= ps["max"] and 1.0 <= c4["lpt_imbalance"] < 1.1


def test_eight_ranks_on_one_device():
    """The driver's N = 8 launch shape — `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` — with the eight ranks
    sharing the box's one device over gloo: weak scaling of cfg2, every rank's complex collated (8 x 1000 residues)."""
    out = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-gather", "--no-secondary", "--no-pmc"],
               {"NAMP_BENCH_ONE_DEVICE": "1", "NAMP_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "2"}, timeout=1500)
    assert out["n_gpus"] == 8 and out["steps"] == 3 and out["scaling"] == "weak"
    col = out["collation"]
    assert col["ranks"] == 8 and col["backend"] == "gloo"
    assert col["collated_residues"] == col["expected_residues"] == 8000
    assert out["config"]["global_batch"] == 8 and out["value"] > 0


def test_default_line_contract():
    """N = 1 default run (short): driver keys, exact-fp32 headline with its roofline on the EXECUTED flops vs the fp32 MFMA peak,
    live PMC traffic, the x3 object, parity with sequence recovery, the CPU baselines, the secondary workloads — in < 6 KB."""
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"], capture_output=True, text=True,
                       env=env, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) < 6144, len(lines[0])
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    keys = list(out)
    assert keys.index("gather") < keys.index("secondary") and keys.index("x3") < keys.index("secondary")
    rf = out["roofline"]
    assert out["dtype"] == "f32" and rf["peak"] == 157.3 and rf["bound"] == "mfma" and rf["basis"] == "executed FLOPs"
    # `frac` prices the FLOPs the launch EXECUTES; the dense formulation's rate sits beside it and may come close to the peak
    assert 0 < rf["frac"] <= 1.0 and rf["frac"] < rf["algorithmic_frac"] <= 1.05
    assert rf["traffic"] is not None, rf.get("traffic_source")
    assert 0.5 * rf["traffic_algorithmic"] < rf["traffic"] < 1.6 * rf["traffic_algorithmic"], rf
    # the clock the dominant launch ran at under the profiler sits beside frac (peaks are quoted at 2.4 GHz)
    if "clock_mhz" in rf:                        # (absent when the box exposes neither amdsmi nor pp_dpm_sclk)
        assert 100 < rf["clock_mhz"] <= 2500 and rf["frac_at_measured_clock"] >= rf["frac"] * 0.95
    ft = out["features"]
    assert "error" not in ft, ft
    assert ft["kernel"] == "edge_features_kernel" and 0 < ft["frac"] <= 1 and 0 < ft["algorithmic_frac"] <= 1.5
    assert ft["cfg2_from_X"]["avg_launch_ms"] > 0 and ft["cfg4_batch"]["avg_launch_ms"] > 0 and ft["cfg4_batch"]["knn_select_ms"] > 0
    assert ft["cfg4_batch"]["tokens"] <= 32000 and ft["traffic"] is not None
    assert keys.index("features") < keys.index("secondary")
    g = out["gather"]
    assert 0 < g["frac"] <= 1.0 and g["peak"] == 8000.0 and g["traffic"] is not None
    assert 0.8 * g["bytes_per_launch"] < g["traffic"] < 2 * g["bytes_per_launch"]
    x3 = out["x3"]
    assert x3["value"] > 0 and 0 < x3["executed_frac"] <= 1.0 and x3["argmax_equal"] is True
    par = out["parity"]
    assert par["argmax_equal"] is True and par["max_abs_dlogp_vs_cpu"] < 1e-3 and par["seq_recovery_gpu_vs_cpu_argmax"] == 1.0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["one_thread"] > 0 and cb["full_forward_from_X"] > 0
    sec = {s_["workload"]: s_ for s_ in out["secondary"]}
    assert set(sec) == {"cfg3", "cfg5", "cfg1", "cfg1s", "cfg4"}, list(sec)
    assert sec["cfg1s"]["cpu_baseline"] > 0 and sec["cfg1s"]["value"] > sec["cfg1"]["value"]        # 30 sequences per call
    for name, s_ in sec.items():
        assert "error" not in s_, (name, s_.get("error"))
        assert s_["value"] > 0
    assert sec["cfg4"]["residues"] > 1_000_000
    # the full detail is on disk when gpurun_out/ exists
    if "detail_file" in out:
        with open(os.path.join(ROOT, out["detail_file"])) as f:
            det = json.load(f)
        assert "per_kernel" in det and len(det["secondary"]) == 5
