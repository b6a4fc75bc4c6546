"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches / n dispatches).
usage: python tools/pmc_summary.py <dir-with-pmc_*-subdirs> [kernel-regex]"""
import collections, csv, glob, re, sys
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"(ekf_fuse_kernel|fuse_pipeline_kernel|ekf_wave_kernel<[\w, ]+>|ekf_wave_big_kernel<[\w, ]+>|umeyama_batch_kernel|ransac_batch_kernel|utm_kernel<\w+>)")
res, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = pat.search(row["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + " grid=" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        res[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(res):
    print("==", k)
    for c in sorted(res[k]):
        n = max(1, len(cnt[k][c]))
        print(f"   {c:28s} {res[k][c] / n:14.5g}  (per dispatch, {n} dispatches)")
