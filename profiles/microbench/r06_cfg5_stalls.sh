# r06: config 5's vote kernel against the headline's -- which queue is full?  TCC / TCP stall and level counters, separate passes,
# kernel trace only ("... or a counter table that names the bound")
R=$PWD; O=$R/gpurun_out/r06c5s; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum" \
           "TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/s5_$i -o k -- python $R/bench.py --config 5 --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $O/cfg5_err_$i.txt
  find /tmp/s5_$i -name "*counter_collection.csv" -exec cp {} $O/cfg5_set_$i.csv \;
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sh_$i -o k -- python $R/bench.py --no-cpu-baseline --configs-block-steps 0 --steps 4 --warmup 1 > /dev/null 2> $O/head_err_$i.txt
  find /tmp/sh_$i -name "*counter_collection.csv" -exec cp {} $O/head_set_$i.csv \;
done
cd $R; python - <<'PY'
import csv, collections, glob
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("gpurun_out/r06c5s/*_set_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_phase2_band" in k: k = "k_phase2_band<1,0,0,true> (config 5)"
        elif "k_ranges_fill_lg_fin" in k: k = "k_ranges_fill_lg_fin (config 5)"
        elif "k_phase2<64, 0, 1, true>" in k: k = "k_phase2<64,0,1,true> (headline)"
        else: continue
        e = tab[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
names = sorted({c for d in tab.values() for c in d})
ks = sorted(tab)
with open("gpurun_out/r06c5s/table.md", "w") as out:
    print("| counter (per launch) | " + " | ".join(ks) + " |", file=out)
    print("|---|" + "---:|" * len(ks), file=out)
    for c in names:
        print("| `%s` | " % c + " | ".join(("%.4g" % (tab[k][c][0] / tab[k][c][1])) if c in tab[k] else "" for k in ks) + " |", file=out)
print(open("gpurun_out/r06c5s/table.md").read())
PY
