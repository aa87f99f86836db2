#!/bin/bash
# tools/pmc_gemm.sh -- issue / stall counters of the tile GEMM (separate --pmc passes of tools/_build/gemm_bench) -> gpurun_out/pmc_gemm/
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_gemm
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -o pmc -- $REPO/tools/_build/gemm_bench 3 > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
cd $REPO
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_gemm/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))
        a = agg[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(agg):
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print('    %-32s %.4g' % (c, v / n))
PY
