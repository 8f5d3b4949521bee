#!/bin/bash
# On the MI355X box: kernel trace + the three PMC passes (HBM fetch, HBM write, wave issue) of the BAL-scale profiling target, summaries under gpurun_out/$1
# (copy what is to be kept into profiles/).  Usage: scripts/profile_all.sh r02 [n_points]
set -e
tag=${1:-r02}; npts=${2:-1000000}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/profile_target.py $npts > $out/run.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 scripts/profile_target.py $npts > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 scripts/profile_target.py $npts > $out/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python3 scripts/profile_target.py $npts > $out/pmc_sq.log 2>&1
wl=$(grep '^workload: ' $out/run.log | sed 's/^workload: //')
python3 scripts/summarize_sq.py $out/pmc_sq $out/issue.json "$wl" > $out/issue.txt 2>&1
python3 scripts/summarize_pmc.py $out/pmc_fetch $out/pmc_write $out/traffic.json "$wl" > $out/traffic_summary.log 2>&1
cp $(ls -t $(find $out/trace -name '*kernel_stats.csv') | head -1) $out/kernel_stats.csv   # (the newest, should the tag be reused)
# the raw per-dispatch files are large: keep the summaries only
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
find $out/trace -name '*kernel_trace.csv' -size +20M -delete
tail -3 $out/run.log; cat $out/traffic_summary.log | tail -12
