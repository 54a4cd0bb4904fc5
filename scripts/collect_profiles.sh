#!/bin/bash
# rocprofv3 passes over the C3 bench (run on the GPU box through gpurun; outputs under gpurun_out/$1, default prof_final).
# Pass 1: kernel trace + stats.  Passes 2/3: HBM-side PMC counters, each in its own run (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Passes 4..: SQ / L2 counters of
# the halo kernels, a few per run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_final}
mkdir -p $OUT
cd $R
CMD="python3 bench.py --workload c3 --steps 1 --warmup 0 --passes-per-step 4 --no-cpu-baseline --no-full-set"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- $CMD > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > $OUT/write.log 2>&1 || exit 1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp -d $OUT/sq$i --output-format csv -- $CMD > $OUT/sq$i.log 2>&1 || echo "counter group $i failed: $grp" >> $OUT/sq_failed.log
done
python3 bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
tail -1 $OUT/bench_c3.json | cut -c1-400
