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
    extra = ["--split-limit", "60"] if workload == "cfg4" else []
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline", "--no-gather",
                "--no-secondary"] + extra, {"NAMP_BENCH_ONE_DEVICE": "1", "NAMP_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == scaling and out["value"] > 0 and out["higher_is_better"] is True
    col = out["collation"]
    if workload == "cfg2":
        assert col["ranks"] == 2 and col["backend"] == "gloo"
        assert col["collated_residues"] == col["expected_residues"] == 2 * 1000
        assert out["config"]["global_batch"] == 2
        assert out["dtype"] == "f32" and out["x3"]["value"] > 0            # both evaluations are in the line on every N
    else:
        assert col["residues_collated"] > 0
        assert out["shard"]["rank0_residues"] < col["residues_collated"]   # the other rank really took a share


def test_default_line_contract():
    """N = 1 default run (short): driver keys, exact-fp32 headline with its roofline vs the fp32 MFMA peak, the x3 object,
    parity with sequence recovery, the CPU baselines, and the secondary workloads."""
    out = _run(["--steps", "5", "--warmup", "2"], timeout=1500)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["dtype"] == "f32" and out["roofline"]["peak"] == 157.3 and out["roofline"]["bound"] == "mfma"
    # `frac` prices the ALGORITHMIC flops (the reference's dense formulation) against the peak: the launch executes half of
    # them (hoisted first layer, layer 3 behind the K-sum), so it may come close to — on a fast box pass — 1; `executed_frac` may not
    assert 0 < out["roofline"]["frac"] <= 1.5 and 0 < out["roofline"]["executed_frac"] <= 1.0
    assert out["x3"]["roofline"]["peak"] == 2500.0 and out["x3"]["parity"]["argmax_equal"] is True
    par = out["parity"]
    assert par["argmax_equal"] is True and par["max_abs_dlogp_vs_cpu"] < 1e-3 and par["seq_recovery"]["gpu_vs_cpu_argmax"] == 1.0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["one_thread"]["cores"] == 1 and cb["full_forward_from_X"]["value"] > 0
    sec = {s.get("config", {}).get("workload", s.get("workload", ""))[:4]: s for s in out["secondary"]}
    assert set(sec) == {"cfg3", "cfg5", "cfg1"}, list(sec)
    for name, s in sec.items():
        assert "error" not in s, (name, s.get("error"))
        assert s["value"] > 0
