#!/usr/bin/env python3
"""Regenerates the measurement table of DESIGN.md section 5 from the committed rocprofv3 summaries (profiles/rNN_*): kernel time =
rocprofv3's average duration of that kernel at that grid in the kernel trace of the default `python3 bench.py` run (and of
tools/bench_kernels.py for the auxiliary kernels), traffic = the PMC passes collected in rNN_traffic.json.  The table between the
markers in DESIGN.md is REPLACED, so the document cannot quote a number the profiles do not hold.

usage: python tools/design_table.py [r05] [--check]     (--check: exit 1 if DESIGN.md is not up to date)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN GENERATED TABLE (tools/design_table.py) -->", "<!-- END GENERATED TABLE -->"
HBM = 8000.0

# (file, kernel as rocprof names it, grid threads) -> (row label, units, algorithmic bytes per launch, unit name)
ROWS = [
    ("bench_default", "ekf_wave_kernel<true, true, 1>", 64000, "**C2 1 000 x 271 (bench default), fused pipeline, the reference's Sim3 rows**", 271_000, 271_000 * 145, "poses"),
    ("bench_default", "ekf_wave_big_kernel<true, 1>", 6400000, "C3 100 000 x 1 000, fused pipeline", 100_000_000, 100_000_000 * 145, "poses"),
    ("bench_default", "ekf_wave_big_kernel<false, 1>", 6400000, "C3, K4 only", 100_000_000, 100_000_000 * 145, "poses"),
    ("bench_default", "ekf_wave_big_kernel<true, 1>", 2097152, "C5 shard chunk 32 768 x 1 000 (38 per pass), fused pipeline", 32_768_000, 32_768_000 * 145, "poses"),
    ("bench_default", "fuse_pipeline_kernel<1, 2, 2>", 1245184, "C5 shard 1 245 184 x 1 000 time-major, lane per trajectory, fused pipeline", 1_245_184_000, 1_245_184_000 * 145, "poses"),
    ("bench_default", "fuse_pipeline_kernel<1, 2, 2>", 100032, "C3 time-major, lane per trajectory, fused pipeline", 100_000_000, 100_000_000 * 145, "poses"),
    ("bench_default", "ekf_fuse_kernel<1, 2, 2>", 100032, "C3 time-major, lane per trajectory, K4 only", 100_000_000, 100_000_000 * 145, "poses"),
    ("bench_default", "windows_fused_kernel", 1000064, "C4 1 M windows x 50 pairs (Umeyama)", 1_000_000, 1_000_000 * 2504, "windows"),
    ("bench_default", "sim3_rows_kernel<true>", 64000, "row choice (ref :973-998) + compaction of the chosen rows, from a register tile: first stage of the robust chain, 1 000 x 271", 271_000, None, "rows"),
    ("bench_default", "robust_probe_kernel", 64000, "early-exit probe of the robust chain, 1 000 x 271: draws + scores its own trials until one counts every row (ref :413), then the final fit", 1000, None, "trajectories"),
    ("bench_default", "gps_prefilter_chain_kernel", 64000, "GPS pre-filter of 1 000 logs (whole-run chain), windows walked on the device", 261_564, None, "fixes"),
    ("bench_default", "eval_errors_lds_kernel", 768000, "error metric of raw SLAM / Sim3 / EKF in one launch (whole-run chain), 1 000 x 271", 813_000, None, "poses"),
    ("aux", "apply_sim3_slab_kernel", 25600000, "K3 apply Sim3, 1e8 poses", 100_000_000, 100_000_000 * 112, "poses"),
    ("aux", "utm_kernel<false>", 25600000, "K1 UTM forward, 1e8 points", 100_000_000, 100_000_000 * 32, "points"),
    ("aux", "utm_kernel<true>", 25600000, "K1 UTM inverse, 1e8 points", 100_000_000, 100_000_000 * 32, "points"),
    ("aux", "gps_rows_to_utm_kernel", 25600000, "geodesy slice (mask + zone + forward + alt), 1e8 rows", 100_000_000, 100_000_000 * 48, "rows"),
    ("aux", "enu_kernel", 12800000, "WGS84 -> local ENU, 1e8 points", 100_000_000, 100_000_000 * 48, "points"),
    ("aux", "ransac_batch_kernel", 256000, "K2b 1 000 x (271 points, 1 000 fed trials)", 1000, None, "trajectories"),
    ("aux", "mt_choice_kernel", 64000, "device draws: 1 000 streams x 1 000 trials of permutation(271)[:4]", 1_000_000, None, "trials"),
    ("bench_default", "mt_tape_kernel", 512, "device draws, ONE stream x 1 000 trials (C1 drop-in), chip-wide route: the tape", 1000, None, "trials"),
    ("bench_default", "mt_transition_kernel", 47744, "... transition tables of 746 segments x 270 start states", 1000, None, "trials"),
    ("bench_default", "mt_compose_kernel", 6912, "... composed tables (27 groups of 28 segments)", 1000, None, "trials"),
    ("bench_default", "mt_expand_kernel", 256, "... walk of the 27 composed tables", 1000, None, "trials"),
    ("bench_default", "mt_expand_kernel", 6912, "... starts handed down to the segments", 1000, None, "trials"),
    ("bench_default", "mt_resolve_kernel", 47744, "... replay of the segments", 1000, None, "trials"),
    ("bench_default", "mt_tape_trace_kernel", 4032, "... trace of the four sample positions", 1000, None, "trials"),
    ("bench_default", "gps_prefilter_chain_kernel", 64, "GPS pre-filter chain of ONE log (279 fixes, 12 window-axis problems)", 279, None, "fixes"),
    ("bench_default", "ransac_scan_kernel", 1024, "K2b for ONE set: 1 000 hypotheses over 16 single-wave blocks", 1000, None, "hypotheses"),
    ("bench_default", "ransac_finish_kernel", 256, "... winner's mask and final fit", 1000, None, "hypotheses"),
    ("bench_default", "ransac_batch_kernel", 256000, "K2b inside the robust chain 1 000 x 271 (valid rows compacted, 1 000 drawn trials)", 1000, None, "trajectories"),
    ("bench_default", "eval_errors_lds_kernel", 256, "error metric of ONE 271-pose track", 271, None, "poses"),
    ("aux", "time_align_kernel", 64000, "time alignment 1 000 x (271 stamps, 279 fixes)", 271_000, None, "stamps"),
    ("aux", "ransac_poly_kernel<128>", 3840000, "GPS pre-filter problems: 30 000 x (150 rows, 50 trials of 6)", 30000, None, "problems"),
    ("aux", "eval_errors_lds_kernel", 256000, "error metric, 1 000 x 271", 271_000, None, "poses"),
    ("aux", "eval_errors_lds_kernel", 2560000, "error metric at the C3 track length, 10 000 x 1 000 (pruned nearest-fix search, sorted median)", 10_000_000, None, "poses"),
]


def load_trace(tag, name):
    path = os.path.join(ROOT, "profiles", f"{tag}_{name}_by_kernel_and_grid.csv")
    out = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            out[(r["kernel"], int(r["grid_threads"]))] = (float(r["avg_us"]), int(r["calls"]))
    return out


def table(tag):
    traces = {"bench_default": load_trace(tag, "bench_default"), "aux": load_trace(tag, "aux")}
    tpath = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    hbm = {}
    for sec, d in traffic.items():
        if isinstance(d, dict):
            for k, v in d.items():
                if isinstance(v, dict) and "hbm_bytes" in v:
                    hbm.setdefault(k, v["hbm_bytes"])
    lines = [f"| workload | kernel (grid threads) | rocprofv3 avg per launch ({tag}) | launches | units/s | algorithmic GB/s | of 8 TB/s | counter traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|"]
    for src, kern, grid, label, units, alg, uname in ROWS:
        hit = traces[src].get((kern, grid))
        if hit is None:                                   # template arguments are printed with or without spaces depending on the tool version
            hit = next((v for (k, g), v in traces[src].items() if g == grid and k.replace(" ", "") == kern.replace(" ", "")), None)
        if hit is None:
            continue
        us, calls = hit
        rate = units / (us * 1e-6)
        gbs = f"{alg / (us * 1e-6) / 1e9:,.0f}" if alg else "--"
        frac = f"{alg / (us * 1e-6) / 1e9 / HBM:.1%}" if alg else "--"
        key = f"{kern} grid={grid}"
        tr = f"{hbm[key] / alg:.2f}x ({hbm[key] / 1e9:.2f} GB)" if alg and key in hbm else "--"
        t = f"{us:,.2f} us" if us < 1000 else f"{us / 1e3:,.3f} ms"
        lines.append(f"| {label} | `{kern}` ({grid:,}) | {t} | {calls:,} | {rate:,.3g} {uname}/s | {gbs} | {frac} | {tr} |")
    return "\n".join(lines)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r05"
    path = os.path.join(ROOT, "DESIGN.md")
    doc = open(path).read()
    if BEGIN not in doc or END not in doc:
        print("DESIGN.md holds no generated-table markers", file=sys.stderr)
        return 2
    new = doc[:doc.index(BEGIN) + len(BEGIN)] + "\n" + table(tag) + "\n" + doc[doc.index(END):]
    if "--check" in sys.argv:
        if new != doc:
            print("DESIGN.md section 5 is not what the profiles say: run python tools/design_table.py", file=sys.stderr)
            return 1
        return 0
    open(path, "w").write(new)
    print(table(tag))
    return 0


if __name__ == "__main__":
    sys.exit(main())
