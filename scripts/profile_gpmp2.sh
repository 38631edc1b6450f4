#!/bin/bash
# rocprofv3 passes of the GPMP2 iteration at C4 (run on the GPU box from the repo root):  bash scripts/profile_gpmp2.sh r02
# kernel trace + stats, then PMC passes (each its own run) of scripts/prof_gpmp2.py; summaries: pmc_summary.py <dir> <tag>_gpmp2
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_${TAG}_gpmp2
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o g -- python3 scripts/prof_gpmp2.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/pmc5 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc5.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/pmc6 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc6.log 2>&1
echo "gpmp2 passes done"
