#!/usr/bin/env python3
"""Price ONE 64-pose chunk of the wave-per-trajectory kernel by instruction class.

The 4-cycles-per-instruction issue floor of bench.py's roofline.valu prices a row_bcast DPP move like an FMA.  Here every class gets
the cost tools/ubench/ilp.hip MEASURED for it on this chip (a lone wave on its SIMD, four independent chains -- the best a wave can do
by itself; gpurun_out/r4a/ilp_w1.log), and the dynamic counts are the hardware's own:

  * counts: rocprofv3 --pmc passes of `bench.py --workload c2 --poses 256` and `--poses 320` (tools/pmc_classes.sh): the two launches
    differ by exactly one full chunk per trajectory, so (counters at 320 - counters at 256) / waves = one chunk's dynamic instructions:
    SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_INSTS_VMEM, and the FP64 arithmetic split SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64;
  * the remaining VALU instructions (cross-lane moves, v_readlane, compares, selects, integer / mask work) are split by the STATIC census of
    the chunk loop's scan stages in the shipped object: DPP moves by control (row_shr / quad_perm inside a 16-lane row vs row_bcast /
    wave_shr across rows) are counted from the disassembly of the loop blocks that hold the scans (they execute once per chunk);
  * time per chunk: (i) the in-kernel clock64 stamps of a clean track (tools/chunk_timing.py on the `make timing` build; pass the cycles with
    --stamp-cycles=N), the like-for-like figure for the per-class costs, which are in clock64 cycles too; (ii) the launch-level increment:
    kernel time at 320 poses minus kernel time at 256 poses from CLEAN (unprofiled) runs of the same two commands (--clean=DIR with
    chunk_time_{ekf,pipeline}.jsonl) -- that one belongs to the slowest wave of the launch.

usage: python tools/price_mix.py [gpurun_out/pmc_classes] [--stamp-cycles=3724] [--clean=gpurun_out/r4t] [--json=profiles/r04_chunk_price.json]
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cycles per instruction, one wave per SIMD, four independent chains (tools/ubench/ilp.hip on MI355X, gpurun_out/r4a/ilp_w1.log);
# dependent-chain figures in the comments
COST = {
    "fp64_fma_mul_add": 4.51,      # v_fma_f64: 6.28 dependent, 5.26 two chains, 4.51 four chains
    "fp64_trans": 10.44,           # v_rcp_f64 / v_rsq_f64 (+ add): 13.5 dependent, 10.4 four chains
    "dpp_in_row": 4.46,            # row_shr / quad_perm move (+ fma): 7.35 dependent, 4.46 four chains
    "dpp_cross_row": 7.75,         # row_bcast:15 / row_bcast:31 / wave_shr:1 -- 7.75 even with four chains
    "readlane": 9.35 / 3 * 2 + 0,  # v_readlane x2 + fma with the scalar: 9.35 per instruction of a dependent step; two of three are readlanes
    "valu_other": 4.5,             # compares, selects, moves, integer / mask work at the plain VALU rate (cmp -> mask -> select chains: 30-45 per step when dependent)
    "salu": 4.0,                   # scalar instructions do not co-issue with the lone wave's vector instructions (7.6 per op in a dependent chain)
    "lds": 16.0,                   # ds_bpermute / LDS access: 16 per instruction with four chains
    "vmem": 4.0,                   # issue slot only (latency hidden by the prefetch of the next chunk)
}
CLOCK_GHZ = 2.4


def read_counters(d):
    """{(kernel, n): {counter: mean per dispatch}} from the csv files of tools/pmc_classes.sh"""
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*_n*_p*", "**", "*_counter_collection.csv"), recursive=True):
        tag = re.search(r"/(ekf|pipeline)_n(\d+)_p\d+/", f)
        for row in csv.DictReader(open(f)):
            if "ekf_wave_kernel" in row["Kernel_Name"]:
                res[(tag.group(1), int(tag.group(2)))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}


def kernel_ms(d, k, n):
    """best kernel time of the CLEAN runs (bench.py --poses n, no profiler) listed in d/chunk_time_<k>.jsonl"""
    ms = []
    path = os.path.join(d, f"chunk_time_{k}.jsonl") if d else None
    if path and os.path.exists(path):
        for line in open(path):
            if line.startswith("{"):
                try:
                    r = json.loads(line)
                    if r["config"]["poses_per_trajectory"] == n:
                        ms.append(r["roofline"]["kernel_ms"])
                except Exception:
                    pass
    return min(ms) if ms else None


def static_dpp_split():
    """DPP moves of the chunk loop by control, from the disassembly of the shipped small-batch object (hipcc -S with the Makefile's flags):
    the blocks that hold the scan stages are the ones with >= 40 DPP instructions (the scans of one chunk; cold quaternion-product scans sit
    in their own block and are listed separately)."""
    src = os.path.join(ROOT, "gps_optimize_slam_amd", "csrc", "gsf_ekf_wave.hip")
    out = "/tmp/price_mix_wave.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-ffp-contract=on",
                           "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    res = {}
    for key, name in (("ekf", "ekf_wave_kernelILb0ELb1ELi1E"), ("pipeline", "ekf_wave_kernelILb1ELb1ELi1E")):
        start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and name in l][0]
        end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
        blocks, cur = [], []
        for l in lines[start + 1:end]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur); cur = []
            else:
                t = l.strip()
                if t.startswith("v_") or t.startswith("s_") or t.startswith("ds_") or t.startswith("global_"):
                    cur.append(t)
        blocks.append(cur)
        inrow = cross = rl = 0
        for b in blocks:
            dpp = [t for t in b if any(c in t for c in ("row_shr", "row_shl", "quad_perm", "row_bcast", "wave_shr", "row_newbcast"))]
            if len(dpp) < 40 or any("v_mul_f64" in t for t in b) is False:
                continue
            quat_scan = sum("row_bcast" in t for t in dpp) >= 16 and len([t for t in b if "f64" in t.split()[0]]) > 3 * len(dpp)   # the generic quaternion prefix product (cold)
            if quat_scan:
                continue
            inrow += sum(any(c in t for c in ("row_shr", "row_shl", "quad_perm")) for t in dpp)
            cross += sum(any(c in t for c in ("row_bcast", "wave_shr")) for t in dpp)
            rl += sum(t.startswith("v_readlane") for t in b)
        res[key] = {"dpp_in_row": inrow, "dpp_cross_row": cross, "readlane_in_scan_blocks": rl}
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    d = args[0] if args else os.path.join(ROOT, "gpurun_out", "pmc_classes")
    opt = {a.split("=", 1)[0]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    clean = opt.get("--clean")
    stamp_cycles = float(opt["--stamp-cycles"]) if "--stamp-cycles" in opt else None
    cnt = read_counters(d)
    stat = static_dpp_split()
    out = {"costs_cycles_per_instruction": COST, "clock_GHz": CLOCK_GHZ, "source": "tools/pmc_classes.sh (rocprofv3 --pmc) + tools/ubench/ilp.hip costs", "kernels": {}}
    for k in ("ekf", "pipeline"):
        a, b = cnt.get((k, 256)), cnt.get((k, 320))
        if not a or not b:
            continue
        waves = b.get("SQ_WAVES", 1000.0)
        per = {c: (b[c] - a[c]) / waves for c in b if c in a and c.startswith("SQ_INSTS")}
        fp = sum(per.get(c, 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
        trans = per.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        valu = per.get("SQ_INSTS_VALU", 0.0)
        s = stat["ekf"]                                                    # the chunk loop's scans are the same in both kernels (the pipeline kernel's prelude has butterflies of its own)
        # v_readlane count of a chunk: carries + lane broadcasts; static count of the hot loop is not separable from cold blocks, so the
        # readlanes inside the scan blocks are taken as a lower bound and the rest stays in "valu_other"
        classes = {"fp64_fma_mul_add": fp, "fp64_trans": trans, "dpp_in_row": s["dpp_in_row"], "dpp_cross_row": s["dpp_cross_row"],
                   "readlane": s["readlane_in_scan_blocks"]}
        classes["valu_other"] = max(0.0, valu - sum(classes.values()))
        classes["salu"] = per.get("SQ_INSTS_SALU", 0.0)
        classes["lds"] = per.get("SQ_INSTS_LDS", 0.0)
        classes["vmem"] = per.get("SQ_INSTS_VMEM", 0.0)
        cyc = {c: n * COST[c] for c, n in classes.items()}
        t256, t320 = kernel_ms(clean, k, 256), kernel_ms(clean, k, 320)
        chunk_us = (t320 - t256) * 1e3 if t256 and t320 else None
        flat4 = (valu) * 4.0
        row = {"dynamic_instructions_per_chunk": {c: round(n, 1) for c, n in classes.items()}, "valu_total": round(valu, 1),
               "raw_counter_differences_per_wave": {c: round(v, 1) for c, v in per.items()},
               "class_weighted_cycles_per_chunk": round(sum(cyc.values()), 0), "cycles_by_class": {c: round(v, 0) for c, v in cyc.items()},
               "flat_4_cycle_valu_floor_cycles": round(flat4, 0),
               "launch_level_us_per_added_chunk": chunk_us}
        if stamp_cycles and k == "ekf":
            row["clean_track_cycles_per_chunk_by_in_kernel_stamps"] = stamp_cycles
            row["class_weighted_floor_over_measured"] = round(sum(cyc.values()) / stamp_cycles, 3)
            row["flat_floor_over_measured"] = round(flat4 / stamp_cycles, 3)
        out["kernels"][k] = row
    txt = json.dumps(out, indent=1)
    print(txt)
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            open(a.split("=", 1)[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
