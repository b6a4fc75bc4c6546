"""The bench.py output contract, checked on the committed sample line (profiles/rNN_bench_default_under_rocprof.json of the latest round: the default
`python bench.py` run of make_profiles.sh on an MI355X) and on bench.py's own constants and helpers -- CPU tier, no GPU needed."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_tag():
    """rNN of the newest committed profile set (profiles/rNN_bench_default_under_rocprof.json)"""
    import glob
    return os.path.basename(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_under_rocprof.json")))[-1]).split("_")[0]


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", f"{latest_tag()}_bench_default_under_rocprof.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("fused poses/sec") and d["unit"] == "fused poses/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if "valu" in r:                                    # the issue-side roofline from the committed PMC profile of the same kernel sources
        v = r["valu"]
        assert abs(v["floor_ms"] - v["valu_wave_instructions_per_launch"] * 4 / (1024 * 2.4e9) * 1e3) < 1e-12 and abs(v["frac"] - v["floor_ms"] / r["kernel_ms"]) < 1e-12
        assert (r["bound"] == "valu") == (v["frac"] > r["frac"])
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["alg_bytes_per_launch"]
    # value = poses per step / step time; the kernel's share of the step is what the roofline is computed from
    poses = d["config"]["trajectories_per_gpu"] * d["config"]["poses_per_trajectory"]
    assert abs(d["value"] - poses / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "fused poses/s" and c["sample"]
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] > 0 and c["cpu_model"]
    # the gate is on the TIMED outputs of the timed kernel, and the reference's own error metric agrees between GPU and CPU results
    assert d["max_abs_pos_err_m"] < 1e-6 and d["status_bits_equal"] is True and d["gated"].startswith("timed fused-pipeline outputs")
    q = d["ref_style_error_q15"]
    assert q["max_abs_diff_m"] < 1e-6 and q["trajectories"] > 0
    assert r["kernel_source_hash"] and (r["traffic"] is None) == (not str(r["traffic_source"]).endswith("_traffic.json"))
    assert abs(r["frac_of_peak_on_moved_bytes"] - r["moved_bytes_per_launch_incl_fit_pass"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-12
    e = d["extra"]
    assert e["pcie_inclusive"]["poses_per_s"] < d["value"] and e["c1_drop_in"]["end_to_end_ms"]["best"] > 0
    assert e["robust_chain_c2"]["ms"] > 0 and e["geodetic_chain_c2"]["ms"] > 0
    if "all_trials" in e["robust_chain_c2"]:             # round 5: the exact early exit next to the chain that draws every trial, and the whole-run chain
        rc = e["robust_chain_c2"]
        assert rc["outputs_identical_in_both_modes"] is True and rc["ms"] < rc["all_trials"]["ms"] and 0.0 <= rc["saturated_share"] <= 1.0
        assert e["full_chain_c2"]["ms"] > 0 and e["full_chain_c2"]["run_status_nonzero"] == 0


def test_committed_traffic_profile_matches_the_kernel_sources():
    """bench.py only quotes profiles/rNN_traffic.json while the kernels are the ones the profile measured."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    doc = json.load(open(os.path.join(ROOT, "profiles", f"{latest_tag()}_traffic.json")))
    assert doc["kernel_source_hash"] == bench.kernel_source_hash(), "re-run tools/make_profiles.sh + tools/collect_profiles.py after changing a kernel"
    got, src = bench.profiled_traffic("c2", "ekf_wave_kernel<true, true, 1>", 64000)
    assert src == f"{latest_tag()}_traffic.json" and 0.9 * 39295000 < got < 1.3 * 39295000      # C2: counters ~ algorithmic bytes (nothing re-read from HBM)
    v = bench.profiled_issue("c2", "ekf_wave_kernel<true, true, 1>", 64000, 0.018)
    assert v is not None and 4.0e6 < v["valu_wave_instructions_per_launch"] < 6.0e6 and 0.3 < v["frac"] < 0.6


def test_spawned_ranks_use_loopback_and_fresh_processes(monkeypatch):
    """`bench.py --gpus N` without a launcher: N child processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR=127.0.0.1, the parent
    itself imports neither torch nor the library (checked by running spawn_ranks with a stub interpreter)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    class P:
        def __init__(self, cmd, env=None, stdout=None):
            seen.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.returncode = 0
        def communicate(self, timeout=None):
            return (b'{"ok": 1}\n', None)
        def wait(self, timeout=None):
            return 0
        def kill(self):
            pass
    monkeypatch.setattr(bench.subprocess, "Popen", P)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    assert "torch" not in bench.__dict__
    assert bench.spawn_ranks(3) == 0
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2"] and all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in seen)
    assert len({e["MASTER_PORT"] for _, e in seen}) == 1 and all(c[1].endswith("bench.py") and c[2:] == ["--gpus", "3", "--steps", "2"] for c, _ in seen)


def test_bench_constants():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    consts = {n.targets[0].id: ast.literal_eval(n.value) for n in tree.body
              if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) and n.targets[0].id in ("ALG_BYTES_PER_POSE", "HBM_PEAK_GBS")}
    assert consts == {"ALG_BYTES_PER_POSE": 145, "HBM_PEAK_GBS": 8000.0}       # 89 B read + 56 B written; MI355X HBM3E spec


def test_spawned_ranks_report_a_stalled_or_failed_rank(monkeypatch):
    """A rank that stalls (its communicate() / wait() times out) or exits non-zero makes `bench.py --gpus N` exit non-zero; the parent
    never waits forever, and rank 0's partial line still reaches stdout."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench

    def popen_factory(hang_rank0, rc_other):
        class P:
            n = 0
            def __init__(self, cmd, env=None, stdout=None):
                self.rank = int(env["RANK"]); self.returncode = None; self.killed = False; self.calls = 0
            def communicate(self, timeout=None):
                self.calls += 1
                if hang_rank0 and self.calls == 1:
                    raise subprocess.TimeoutExpired("bench", timeout)
                self.returncode = -9 if self.killed else 0
                return (b'{"partial": 1}\n', None)
            def wait(self, timeout=None):
                self.returncode = rc_other
                return rc_other
            def kill(self):
                self.killed = True
        return P
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("GSF_BENCH_RANK_TIMEOUT_S", "0.01")
    monkeypatch.setattr(bench.subprocess, "Popen", popen_factory(True, 0))
    assert bench.spawn_ranks(2) == 124
    monkeypatch.setattr(bench.subprocess, "Popen", popen_factory(False, bench.STALL_EXIT_CODE))
    assert bench.spawn_ranks(2) == bench.STALL_EXIT_CODE
    monkeypatch.setattr(bench.subprocess, "Popen", popen_factory(False, 0))
    assert bench.spawn_ranks(2) == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os._exit(0)" not in src and "os._exit(STALL_EXIT_CODE)" in src


def test_stream_rate_of_the_committed_aux_profile():
    """roofline.stream of big batches divides the counter bytes by the rate K3 moved the same pose rows at in the committed profile of the
    auxiliary kernels: the figure must be there and be a plausible HBM rate (between half and all of the 8 TB/s peak)."""
    import bench
    got = bench.profiled_stream_rate()
    assert got is not None, "profiles/rNN_aux_by_kernel_and_grid.csv holds no apply_sim3 row at 1e8 poses"
    rate, src = got
    assert 4.0e12 < rate < 8.0e12, rate
    assert src.endswith("_aux_by_kernel_and_grid.csv")


def test_design_document_quotes_the_committed_profiles():
    """the generated table and the chains paragraph are what tools/design_table.py / tools/design_chains.py produce from the
    committed profiles (a profile refreshed without the document, or the other way round, fails here)."""
    import subprocess
    import sys
    for tool in ("design_table.py", "design_chains.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), latest_tag(), "--check"], capture_output=True, text=True)
        assert r.returncode == 0, (tool, r.stdout[-300:], r.stderr[-300:])
