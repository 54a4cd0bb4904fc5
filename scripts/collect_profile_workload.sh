#!/bin/bash
# kernel trace of one bench workload (c2 / giant: the giants' mode with the tile-parallel build); output under gpurun_out/$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-c2}
OUT=$R/gpurun_out/${2:-prof_$W}
mkdir -p $OUT
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --workload $W --steps 1 --warmup 0 --passes-per-step 4 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
