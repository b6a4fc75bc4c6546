"""Turn gpurun_out/profiles_raw/ (tools/make_profiles.sh) into the committed, judged artefacts under profiles/:
   rNN_bench_default_kernel_stats.csv, rNN_bench_default_by_kernel_and_grid.csv, rNN_pmc_<workload>.txt, rNN_traffic.json
usage: python tools/collect_profiles.py r01"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw, out = os.path.join(root, "gpurun_out", "profiles_raw"), os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
ks = sorted(glob.glob(raw + "/trace_bench_default/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)   # newest first: gpurun MERGES runs
kt = sorted(glob.glob(raw + "/trace_bench_default/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime, reverse=True)
if ks:
    shutil.copy(ks[0], f"{out}/{tag}_bench_default_kernel_stats.csv")
if kt:
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "summarize_trace.py"), kt[0], f"{out}/{tag}_bench_default_by_kernel_and_grid.csv"])
if os.path.exists(raw + "/bench_default.json"):
    shutil.copy(raw + "/bench_default.json", f"{out}/{tag}_bench_default_under_rocprof.json")
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_hash: the profile is only quoted by bench.py while the kernels are the ones it measured)
traffic = {"kernel_source_hash": bench.kernel_source_hash()}
pat = re.compile(r"(ekf_wave_kernel<[\w, ]+>|ekf_wave_big_kernel<[\w, ]+>|ekf_wave_duo_kernel<[\w, ]+>|ekf_block_kernel<[\w, ]+>|windows_fused_kernel|windows_moments_kernel|windows_finalize_kernel|sim3_rows_kernel<\w+>|sim3_rows_kernel|fuse_pipeline_kernel<[\w, ]+>|ekf_fuse_kernel<[\w, ]+>|fuse_pipeline_kernel|ekf_fuse_kernel|umeyama_batch_kernel|ransac_batch_kernel|"
                 r"apply_sim3_kernel|apply_sim3_slab_kernel|utm_kernel<\w+>|gps_rows_to_utm_kernel|enu_kernel|time_align_kernel|eval_errors_kernel|eval_errors_lds_kernel|ransac_poly_kernel|mt_choice_kernel|"
                 r"ransac_rows_kernel|ransac_scan_kernel|ransac_finish_kernel|mt_tape_kernel|mt_transition_kernel|mt_compose_kernel|mt_expand_kernel|mt_resolve_kernel|mt_tape_trace_kernel|"
                 r"compact_valid_kernel|transpose_kernel<[\w, ]+>|robust_probe_kernel|gps_prefilter_chain_kernel|run_compact_rows_kernel|run_filtered_rows_kernel|run_outcome_kernel|"
                 r"robust_init_pose_kernel|robust_finish_kernel)")
kt_aux = sorted(glob.glob(raw + "/trace_aux/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime, reverse=True)
if kt_aux:
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "summarize_trace.py"), kt_aux[0], f"{out}/{tag}_aux_by_kernel_and_grid.csv"])
ks_aux = sorted(glob.glob(raw + "/trace_aux/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)
if ks_aux:
    shutil.copy(ks_aux[0], f"{out}/{tag}_aux_kernel_stats.csv")
if os.path.exists(raw + "/aux_kernels.json"):
    txt = open(raw + "/aux_kernels.json").read()
    open(f"{out}/{tag}_aux_kernels.json", "w").write(txt[txt.index("{"):])
for wl in ("c2", "c3", "c3ekf", "c2ekf", "c2block", "c2blockekf", "c5chunk", "c3lane", "c3laneekf", "c2all", "c3all", "aux"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    # gpurun MERGES a call's outputs into gpurun_out/: a pass that was run again leaves the earlier run's files (another pid) beside the
    # new ones -- only the newest file of each pass directory is read
    newest = {}
    for f in glob.glob(f"{raw}/pmc_{wl}_*/**/*_counter_collection.csv", recursive=True):
        d = f[len(raw) + 1:].split(os.sep)[0]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in sorted(newest.values()):
        for row in csv.DictReader(open(f)):
            m = pat.search(row["Kernel_Name"])
            if m:
                per[m.group(1) + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = []
    for k in sorted(per):
        lines.append(f"== {k}")
        for c in sorted(per[k]):
            v = per[k][c]
            lines.append(f"   {c:24s} mean/dispatch {sum(v) / len(v):14.6g}   ({len(v)} dispatches)")
        fs, ws = per[k].get("FETCH_SIZE"), per[k].get("WRITE_SIZE")
        if fs and ws:
            # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
            # coalesced streaming read -> doubled before it is compared with a byte count; WRITE_SIZE is exact
            fetch = 2.0 * 1024.0 * sum(fs) / len(fs)
            write = 1024.0 * sum(ws) / len(ws)
            traffic.setdefault(wl, {})[k] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
                                            "FETCH_SIZE_raw_KiB": sum(fs) / len(fs), "WRITE_SIZE_raw_KiB": sum(ws) / len(ws)}
            lines.append(f"   -> HBM traffic per launch: fetch {fetch / 1e9:.3f} GB (2 x FETCH_SIZE, gfx950 correction) + write {write / 1e9:.3f} GB")
        # issue-side counters per launch (bench.py's roofline.valu: SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x clock))
        iss = {c: sum(per[k][c]) / len(per[k][c]) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES",
                                                           "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if per[k].get(c)}
        if iss.get("SQ_INSTS_VALU"):
            traffic.setdefault(wl, {}).setdefault(k, {})["issue"] = iss
            fl = iss["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9)
            lines.append(f"   -> VALU issue floor: {iss['SQ_INSTS_VALU']:.4g} wave-instructions x 4 cycles / (1 024 SIMDs x 2.4 GHz) = {fl * 1e6:.2f} us per launch"
                         + (f"; VALU active {iss['SQ_ACTIVE_INST_VALU'] / iss['SQ_WAVE_CYCLES']:.0%} / waiting {iss['SQ_WAIT_ANY'] / iss['SQ_WAVE_CYCLES']:.0%} of the wave cycles"
                            if iss.get("SQ_WAVE_CYCLES") and iss.get("SQ_ACTIVE_INST_VALU") and iss.get("SQ_WAIT_ANY") else ""))
    if lines:
        open(f"{out}/{tag}_pmc_{wl}.txt", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print(open(f"{out}/{tag}_traffic.json").read())
