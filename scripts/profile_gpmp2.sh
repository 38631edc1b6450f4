#!/bin/bash
# rocprofv3 passes of the GPMP2 iteration at C4 (run on the GPU box from the repo root):  bash scripts/profile_gpmp2.sh r02
# kernel trace + stats, then PMC passes (each its own run) of scripts/prof_gpmp2.py; summaries: pmc_summary.py <dir> <tag>_gpmp2
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_${TAG}_gpmp2
mkdir -p $OUT
export TMPDIR=/tmp
# the launcher's form (round 6: the low-rank form, csrc/mpb_gpmp2_lr.hip): kernel trace + stats, two counter passes
rocprofv3 --kernel-trace --stats -d $OUT/stats -o g -- python3 scripts/prof_gpmp2.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python3 scripts/prof_gpmp2.py > $OUT/pmc4.log 2>&1
echo "gpmp2 (low-rank form) passes done"
# the block elimination of rounds 1-5 (still the path of several chained fields at H > 64 and of H > 128): its own counter passes
export MPB_GPMP2_FORM=block
OUTB=${OUT}_block
mkdir -p $OUTB
rocprofv3 --kernel-trace --stats -d $OUTB/stats -o g -- python3 scripts/prof_gpmp2.py > $OUTB/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUTB/pmc1 -o p -- python3 scripts/prof_gpmp2.py > $OUTB/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUTB/pmc2 -o p -- python3 scripts/prof_gpmp2.py > $OUTB/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUTB/pmc3 -o p -- python3 scripts/prof_gpmp2.py > $OUTB/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUTB/pmc4 -o p -- python3 scripts/prof_gpmp2.py > $OUTB/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES -d $OUTB/pmc6 -o p -- python3 scripts/prof_gpmp2.py > $OUTB/pmc6.log 2>&1
unset MPB_GPMP2_FORM
echo "gpmp2 (block form) passes done"
