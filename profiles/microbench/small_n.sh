# kernel trace of small_n_scan.py: per-call durations of k_phase2 / k_finalize by batch size
mkdir -p gpurun_out/smalln; R=$PWD; cd /tmp && export TMPDIR=/tmp
python $R/profiles/microbench/small_n_scan.py > $R/gpurun_out/smalln/events.txt 2>&1
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/smalln -o t -- python $R/profiles/microbench/small_n_scan.py > /dev/null 2> $R/gpurun_out/smalln/err.txt
find /tmp/smalln -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/smalln/trace.csv \;
cd $R; python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/smalln/trace.csv")) if "k_phase2" in r["Kernel_Name"] or "k_finalize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("gpurun_out/smalln/kernels.txt", "w")
for r in rows:
    out.write("%-24s grid %8s  %8.2f us\n" % (r["Kernel_Name"][:24], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
