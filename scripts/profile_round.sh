#!/bin/bash
# rocprofv3 passes of one round (run on the GPU box from the repo root):  bash scripts/profile_round.sh r02
#   1. kernel trace + stats of the bench command (main line only, the driver's --steps 20 --warmup 5);
#   2.-6. PMC passes (each its own run, --kernel-trace only next to --pmc) of scripts/prof_stomp.py: the same C3 loop
#         (one launch of the persistent kernel = MPB_ITERS iterations).
# Raw output under gpurun_out/prof_<tag>/ (scratch); scripts/pmc_summary.py writes the summaries kept under profiles/.
set -e
TAG=${1:-rXX}
ITERS=${2:-20}        # iterations per profiled launch: the driver times --steps 20
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python3 bench.py --steps $ITERS --warmup 5 --no-cpu-baseline --main-only > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats pass done"
export MPB_ITERS=$ITERS MPB_FUSED=1 MPB_LAUNCHES=24     # scripts/prof_stomp.py: the persistent kernel, 6 launches of 200 iterations from the initial means = bench.py's timed launch
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc3.log 2>&1
echo "pmc3 done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc4.log 2>&1
echo "pmc4 done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/pmc5 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc5.log 2>&1
echo "pmc5 done"
# instruction classes of the VALU work (full-rate add / mul, fma, transcendental, integer, convert) and the cycles in which a
# matrix instruction and a vector instruction of a SIMD were executing together
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/pmc6 -o p -- python3 scripts/prof_stomp.py > $OUT/pmc6.log 2>&1
echo "pmc6 done"
python3 scripts/pmc_summary.py $OUT $TAG
