#!/bin/bash
# Runs ON the GPU box: PMC passes (one counter group per run, --kernel-trace only: MI355X_MICROARCH.md HBM section) of ONE
# kernel of ONE layer of tools/layer_bench.py.
# Usage: bash tools/pmc_layer.sh <tag> <layer> <kernel substring> <grid_x workgroups (0 = any)> <out name>
set -u
TAG=$1; L=$2; KERN=$3; GX=$4; NAME=$5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DIRS=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$OUT/pmcl_${NAME}_$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -f csv -d $d -- python $ROOT/tools/layer_bench.py $L 3 > $d.log 2>&1
  DIRS="$DIRS $d"
done
python $ROOT/tools/pmc_collect.py "$KERN" $GX $OUT/${TAG}_pmc_${NAME}.json $DIRS > $OUT/pmcl_${NAME}_collect.log 2>&1
rm -rf $DIRS
cd $ROOT
