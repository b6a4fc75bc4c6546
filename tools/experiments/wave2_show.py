"""prints what tools/experiments/wave2_run.sh left in gpurun_out/TAG"""
import ast, glob, json, os, sys
T = os.path.join("gpurun_out", sys.argv[1])
if os.path.exists(T + "/check.log"):
    print(open(T + "/check.log").read().strip().splitlines()[-2:])
for l in open(T + "/t.log"):
    if l.startswith("raw status"):
        raw = ast.literal_eval(l.split("=", 1)[1].strip())
        names = {0: "entry", 6: "prelude done", 8: "chunk1 top", 9: "masks+decisions", 10: "variance scans", 11: "orientation/generic", 12: "position scans", 13: "rts", 14: "carry+arrived", 15: "stores"}
        prev = None
        for k in sorted(names):
            c = raw[2 * k]
            print(f"{names[k]:22s} {c - raw[0]:8d} (+{(c - prev) if prev is not None else 0:6d})  wall {(raw[2 * k + 1] - raw[1]) / 100:.2f} us"); prev = c
res = {}
for f in sorted(glob.glob(T + "/b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res.setdefault(os.path.basename(f)[2:-7], []).append(round(d["roofline"]["kernel_ms"] * 1e3, 2))
    except Exception as e:
        print(f, "unreadable", e)
for k, v in res.items():
    print(k, v)
