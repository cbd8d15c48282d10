# SQ counters of k_epx_key2 (bench.py --config 4): instruction mix, LDS bank conflicts, busy / wait cycles; one pass per group
mkdir -p gpurun_out/k5sq; R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/k5sq_$i -o k5 -- python $R/bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/k5sq/err_$i.txt
  find /tmp/k5sq_$i -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/k5sq/g$i.csv \;
done
cd $R; python - <<PY
import csv, collections, glob
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/k5sq/g*.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for kn in ("k_epx_key2", "k_kp_scatter", "k_kp_hist"):
            if kn in n: tab[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in tab.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %14.0f" % (c, sum(v) / len(v)))
PY
