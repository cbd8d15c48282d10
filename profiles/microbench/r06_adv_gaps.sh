#!/bin/bash
# r06: where an adversarial pass spends its time: kernel trace (start / end of every kernel) of bench.py --config adversarial
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/advgaps
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/advgaps/prof -o adv -- python /root/repo/bench.py --config adversarial --steps 8 > /root/repo/gpurun_out/advgaps/line.json 2>/dev/null
cd /root/repo
f=$(ls gpurun_out/advgaps/prof/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: find the last k_lazy_flush/k_digest (verification) and walk back over 64 vote kernels
names = [r["Kernel_Name"] for r in rows]
votes = [i for i, n in enumerate(names) if "k_phase2" in n]
last = votes[-1]; first = votes[-64]
t0 = int(rows[first]["Start_Timestamp"])
out = []
prev_end = None
for i in range(first - 4, last + 1):
    r = rows[i]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    out.append("%9.2f %7.2f gap %6.2f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    prev_end = e
open("gpurun_out/advgaps/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
tot = (int(rows[last]["End_Timestamp"]) - t0) / 1e3
ksum = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(first, last + 1)) / 1e3
print("pass: %.1f us wall, %.1f us in kernels" % (tot, ksum))
PY
