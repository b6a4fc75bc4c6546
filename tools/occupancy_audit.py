#!/usr/bin/env python3
"""Occupancy audit of every kernel of libgsf.so from the code-object metadata (hipcc -S of each translation unit with the Makefile's plain flags):
registers (allocated in blocks of 8; 512 per SIMD lane), LDS per workgroup (160 KB per CU) -> waves per SIMD each of them admits.  A kernel that
sits just past a step (170 registers = 176 allocated = 2 waves; 14.5 KB of LDS per one-wave block = 11 blocks per CU) shows up here.
usage: python tools/occupancy_audit.py [unit ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gps_optimize_slam_amd", "csrc")
FLAGS = {"gsf_ekf_wave": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-ffp-contract=on"], "gsf_ekf_wave_big": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-ffp-contract=on"],
         "gsf_ekf_block": ["-ffp-contract=on"]}
units = sys.argv[1:] or sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip"))
filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "c++filt"
for u in units:
    out = f"/tmp/occ_{u}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S"] + FLAGS.get(u, []) + [os.path.join(CSRC, u + ".hip"), "-o", out],
                   stderr=subprocess.DEVNULL, check=True)
    txt = open(out).read()
    for b in re.findall(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))
        v, a, l, mx, sp = g("vgpr_count"), g("agpr_count"), g("group_segment_fixed_size"), g("max_flat_workgroup_size"), g("vgpr_spill_count")
        alloc = (v + 7) // 8 * 8
        wv = min(8, 512 // alloc) if alloc else 8
        wpb = max(1, mx // 64)
        wl = min(8, (163840 // l) * wpb // 4) if l else 8
        try:
            dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dn = name
        dn = re.sub(r"\(anonymous namespace\)::", "", dn).split("(")[0][:58]
        near = ""
        if alloc and 512 // alloc < 8 and 512 // max(8, alloc - 8) > 512 // alloc: near = "  <- 8 registers from one more wave"
        if l and (163840 // l) < 32 and 163840 // max(1, l - 1024) > 163840 // l: near += "  <- 1 KB of LDS from one more block"
        print(f"{u:16s} {dn:58s} vgpr {v:3d} ({alloc:3d}) -> {wv} | lds {l:6d} B, {wpb:2d} waves/block -> {wl} | spills {sp}{near}")
