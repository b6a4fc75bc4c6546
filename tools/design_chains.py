#!/usr/bin/env python3
"""Rewrites the "Chains and host-level figures" paragraph of DESIGN.md section 5 from the committed bench line (profiles/rNN_bench_default_under_rocprof.json),
so that the prose quotes what the committed line holds.  usage: python tools/design_chains.py [r04] [--check]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r04")
d = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench_default_under_rocprof.json")).read().strip().splitlines()[-1]); e = d["extra"]
r, r8, c, c5 = e["robust_chain_c2"], e["robust_chain_8192"], e["c1_drop_in"], e["c5_shard_1gpu"]
hp = e.get("host_pointer_entry", {})
new = f"""Chains and host-level figures of the same bench line (`profiles/{tag}_bench_default_under_rocprof.json`, `extra` block; wall-clock of the
whole chain, not one kernel; all with the reference's Sim3 rows): robust chain 1 000 × 271 (row choice → compaction → draw → K2b → Sim3 → K4) {r['ms']:.2f} ms (draws {r['draws_ms']:.2f} + K2b 0.29)
and {r8['ms']:.1f} ms for 8 192 streams ({r8['ms_per_1000_streams']:.2f} ms per 1 000 streams: {r8['per_stream_cost_vs_1000_streams']:.3f} of the 1 000-stream cost per stream, the draws at {r8['draws_ns_per_stream_trial']:.2f} ns per stream-trial either way);
chain from the geodetic log (K1 → align → pipeline) {e['geodetic_chain_c2']['ms']:.3f} ms; C2 in the time-major layout (two fused transposes + pipeline) {e['c2_time_major_pipeline_ms']:.3f} ms; C1 drop-in (271 poses, warm, under the
profiler) GPS leg {c['gps_projection_and_prefilter_ms']['best']:.2f} ms + steps 2–5 {c['steps_2_to_5_ms']['best']:.2f} ms + metric {c['step_6_error_metric_ms']['best']:.2f} ms = **{c['end_to_end_ms']['best']:.2f} ms** end to end (reference ≈ 130–150 ms); C5 shard on one GPU
{c5['traj_major_wave_per_traj']['pass_ms']:.1f} ms per pass trajectory-major ({c5['traj_major_wave_per_traj']['hbm_frac']:.3f} of peak) vs {c5['time_major_lane_per_traj']['pass_ms']:.1f} ms time-major ({c5['time_major_lane_per_traj']['hbm_frac']:.3f}; 54.6 ms in round 3, fitting every valid row);
the host-pointer entry of the boundary (host arrays in and out, PCIe included) {hp.get('ms_per_call', float('nan')):.2f} ms per C2 call = {hp.get('poses_per_s', float('nan')) / 1e9:.2f} G poses/s (the torch-side
`pcie_inclusive` loop of the same line reads {e['pcie_inclusive']['ms_per_step']:.1f} ms under rocprofv3, which slows the copies; 0.8–1.1 ms without it).
"""
p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
i = s.index("Chains and host-level figures of the same bench line"); j = s.index("Box to box these figures move")
if "--check" in sys.argv:
    sys.exit(0 if s[i:j] == new else 1)
open(p, "w").write(s[:i] + new + s[j:])
