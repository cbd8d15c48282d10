# what bounds k_phase2<1,0,0,true> on config 5: instruction mix and wait cycles (SQ counters, separate passes, kernel trace only)
mkdir -p gpurun_out/cfg5sq; R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_$i -o k -- python $R/bench.py --config 5 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $R/gpurun_out/cfg5sq/err_$i.txt
  find /tmp/sq_$i -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/cfg5sq/set_$i.csv \;
done
cd $R; python - <<'PY'
import csv, collections, glob
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("gpurun_out/cfg5sq/set_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_phase2" not in k and "k_ranges_fill_lg" not in k: continue
        k = "k_phase2" if "k_phase2" in k else "k_ranges_fill_lg"
        e = tab[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
for k, d in tab.items():
    print(k)
    for c, (v, n) in d.items(): print("   %-24s %14.0f per launch" % (c, v / n))
PY
