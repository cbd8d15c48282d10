cd /tmp; export TMPDIR=/tmp
for b in acceptor per_slot; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/thp_$b -o t -- python $GRAFT_REPO_ROOT/bench.py --config thrifty --ballot $b --no-cpu-baseline --steps 10 > /dev/null 2>&1
echo "== $b"; find /tmp/thp_$b -name "*kernel_stats.csv" -exec cat {} \; | grep "fpx::" | sed 's/fpx::Geom, fpx::State, fpx::Batch//' | cut -c1-110 | head -6
done
