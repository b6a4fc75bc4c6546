"""The bench.py output contract, checked on the committed sample line (profiles/r01_bench_default_under_rocprof.json: the default
`python bench.py` run of make_profiles.sh on an MI355X) and on bench.py's own constants -- CPU tier, no GPU needed."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r01_bench_default_under_rocprof.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("fused poses/sec") and d["unit"] == "fused poses/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["alg_bytes_per_launch"]
    # value = poses per step / step time; the kernel's share of the step is what the roofline is computed from
    poses = d["config"]["trajectories_per_gpu"] * d["config"]["poses_per_trajectory"]
    assert abs(d["value"] - poses / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "fused poses/s" and c["sample"]
    assert d["max_abs_pos_err_m"] < 1e-6 and d["status_bits_equal"] is True


def test_bench_constants():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    consts = {n.targets[0].id: ast.literal_eval(n.value) for n in tree.body
              if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) and n.targets[0].id in ("ALG_BYTES_PER_POSE", "HBM_PEAK_GBS")}
    assert consts == {"ALG_BYTES_PER_POSE": 145, "HBM_PEAK_GBS": 8000.0}       # 89 B read + 56 B written; MI355X HBM3E spec
