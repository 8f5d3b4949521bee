#!/bin/bash
# After scripts/round_profiles.sh $1 on the box: copy the summaries into profiles/ under the round's names.
set -e
tag=${1:-r03}
src=gpurun_out/$tag
cp $src/kernel_stats.csv profiles/${tag}_bal_scale_kernel_stats.csv
grep -v "^[WEI]2026" $src/run.log | sed '/^$/d' > profiles/${tag}_bal_scale_run.log
cp $src/issue.txt profiles/${tag}_bal_scale_issue.txt
cp $src/issue.json $src/traffic.json profiles/
cp $(ls -t $src/kitti/vm/*kernel_stats.csv | head -1) profiles/${tag}_kitti_like_kernel_stats.csv
grep -v "^[WEI]2026" $src/kitti_run.log > profiles/${tag}_kitti_like_run.log
cp $(ls -t $src/batch/vm/*kernel_stats.csv | head -1) profiles/${tag}_batched_windows_kernel_stats.csv
grep -v "^[WEI]2026" $src/batch_run.log > profiles/${tag}_batched_windows_run.log
tail -1 gpurun_out/${tag}_bench.json > profiles/${tag}_bench_line.json
ls -la profiles | grep $tag
