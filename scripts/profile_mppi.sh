set -e
export TMPDIR=/tmp
TAG=r04a
OUT=gpurun_out/prof_${TAG}_mppi
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 scripts/prof_mppi.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_mppi.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_mppi.py > $OUT/pmc2.log 2>&1
python3 scripts/pmc_summary.py $OUT ${TAG}_mppi > /dev/null
cp profiles/${TAG}_mppi_* gpurun_out/
cat $OUT/stats.log | tail -2
