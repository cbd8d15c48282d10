# r05: the round's rocprofv3 evidence in one go (run from the repo root on the GPU box)
set -u; R=$PWD
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
for c in 4 5 thrifty_random adversarial; do
  bash profiles/microbench/cfg_pmc.sh $c > gpurun_out/r05_pmc_cfg$c.md 2>&1
done
for c in 4 5; do bash profiles/microbench/cfg_prof.sh $c; done
ls gpurun_out/prof_r05 gpurun_out/cfg4 gpurun_out/cfg5 | head -30
tail -3 gpurun_out/r05_pmc_cfg4.md
