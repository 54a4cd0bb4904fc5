#!/bin/bash
# rocprofv3 passes over the C3 bench (run on the GPU box through gpurun; outputs under gpurun_out/prof_final).
# Pass 1: kernel trace + stats.  Passes 2/3: HBM-side PMC counters, each in its own run (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd $R
CMD="python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- $CMD > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > $OUT/write.log 2>&1 || exit 1
python3 bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
tail -1 $OUT/bench_c3.json | cut -c1-400
