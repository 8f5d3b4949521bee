#!/bin/bash
# On the MI355X box: everything the per-round files under profiles/ come from -- the GPU tests, the three kernel traces,
# the PMC passes and the bench line -- under gpurun_out/$1.  Then, back in the repository, scripts/collect_profiles.sh $1.
# Usage: scripts/round_profiles.sh r03
set -e
tag=${1:-r03}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rm -rf gpurun_out/$tag
bash scripts/profile_all.sh $tag > gpurun_out/${tag}_profile_all.log 2>&1
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kitti -- python3 scripts/profile_small.py kitti > gpurun_out/$tag/kitti_run.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/batch -- python3 scripts/profile_small.py batch > gpurun_out/$tag/batch_run.log 2>&1
# the bench reads the stamped PMC summaries from profiles/
cp gpurun_out/$tag/issue.json gpurun_out/$tag/traffic.json profiles/
timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_tests.log
