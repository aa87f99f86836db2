#!/bin/bash
# tools/profile.sh <tag> <bench args...> -- rocprofv3 kernel-trace stats + PMC passes of one bench.py command.
# Run ON THE GPU BOX (via gpurun) from the repo root. Outputs under gpurun_out/prof_<tag>/; summarise with
# tools/summarize_prof.py and copy the summary into profiles/.
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-steady $*"
echo "== kernel trace + stats: $CMD"
timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD --manifest-out $OUT/manifest.json > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '+' | cut -c1-40)
  echo "== pmc pass: $pass"
  timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD --steps 2 --warmup 1 > $OUT/pmc_$name.log 2>&1
  tail -1 $OUT/pmc_$name.log
done
cd $REPO
find $OUT -name '*.csv' | head -40
python tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
head -60 $OUT/summary.md
python tools/trace_by_grid.py $OUT/trace depthwise wino_gemm gemm_mfma stream_gemm wino_chain > $OUT/by_grid.txt 2>/dev/null
# the raw traces are large: keep only the stats + counter csvs
find $OUT -name '*kernel_trace.csv' -size +8M -delete
