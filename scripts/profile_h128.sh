set -e
export TMPDIR=/tmp MPB_H=128 MPB_ITERS=20 MPB_FUSED=1 MPB_LAUNCHES=12
OUT=gpurun_out/prof_hx
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc2.log 2>&1
python3 scripts/pmc_summary.py $OUT ${1:-rXX}_h128 > /dev/null || true
ls profiles | grep h128
mkdir -p gpurun_out/profiles_h128; cp profiles/${1:-rXX}_h128* gpurun_out/profiles_h128/ 2>/dev/null || true
rm -rf $OUT/pmc1 $OUT/pmc2
