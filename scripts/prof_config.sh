# kernel trace of one bench configuration, single stream (clean per-kernel times): bash scripts/prof_config.sh <tag> <bench args...>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
O=gpurun_out/prof_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-timing --single-stream "$@" > $O.log 2>&1 &&
python scripts/kernel_breakdown.py $(find $O -name "*_results.db" | head -1) 3 > gpurun_out/breakdown_$tag.txt && head -5 gpurun_out/breakdown_$tag.txt
