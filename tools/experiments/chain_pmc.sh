#!/bin/bash
# PMC passes over the kernels of the whole-run chain (tools/experiments/robust_chain_trace.py: 40 robust chains + 20 whole-run chains at 1 000 x 271):
# instruction mix, wait shares, instruction-cache counters, per kernel.  Separate runs per counter group (no trace domains alongside --pmc).
# usage (GPU box): bash tools/experiments/chain_pmc.sh  -> gpurun_out/chain_pmc/summary.txt  (copy to profiles/rNN_pmc_chain_kernels.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/chain_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $OUT/p$i -- python3 $R/tools/experiments/robust_chain_trace.py > $OUT/p$i.log 2>&1 || echo "pass $i ($p) failed" >> $OUT/errors.txt
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", "")).split("(")[0].replace("void ", "")
        grid = r.get("Grid_Size", r.get("Grid_Size_X", "0"))
        if any(k in name for k in ("gps_prefilter_chain", "eval_errors_lds", "robust_probe", "time_align", "sim3_rows", "ransac_batch", "gps_rows_to_utm", "ekf_wave_kernel")):
            acc[f"{name} grid={grid}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per-launch averages over the launches of tools/experiments/robust_chain_trace.py (1 000 x 271); SQ_*_CYCLES in units of 4 clocks")
for kern in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[kern].items()}
    n = len(next(iter(acc[kern].values())))
    print(f"\n{kern}  ({n} launches)")
    for k in sorted(c): print(f"  {k:26s} {c[k]:16.1f}")
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w:
        tot = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
        print(f"  -> {tot / max(c.get('SQ_WAVES', 1.0), 1.0):.0f} instructions per wave, {4.0 * w / max(tot, 1.0):.1f} clocks per instruction, "
              f"{c.get('SQ_WAIT_ANY', 0.0) / w:.0%} of the wave cycles in s_waitcnt, {c.get('SQ_WAIT_INST_ANY', 0.0) / w:.1%} waiting for instructions, "
              f"VALU busy {c.get('SQ_ACTIVE_INST_VALU', 0.0) / w:.0%}; instruction-cache misses {c.get('SQC_ICACHE_MISSES', 0.0) / max(c.get('SQC_ICACHE_REQ', 1.0), 1.0):.2%}")
PY
cat $OUT/summary.txt | head -120
