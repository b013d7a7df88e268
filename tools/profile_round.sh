#!/bin/bash
# rocprofv3 collection for one round (run on the GPU box via gpurun):
#   tools/profile_round.sh r01
# kernel trace + stats, two SQ counter passes, and FETCH_SIZE / WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: TCC slots cannot hold both).  Summaries land in gpurun_out/prof_<tag>/.
set -x
TAG=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
BENCH="python3 $R/bench.py --steps 30 --warmup 5 --cpu-rows 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace_stdout.log 2>&1
echo trace rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o bench -- $BENCH --no-kernel-timing > $OUT/pmc_sq1_stdout.log 2>&1
echo pmc1 rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o bench -- $BENCH --no-kernel-timing > $OUT/pmc_sq2_stdout.log 2>&1
echo pmc2 rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH --no-kernel-timing > $OUT/pmc_fetch_stdout.log 2>&1
echo fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH --no-kernel-timing > $OUT/pmc_write_stdout.log 2>&1
echo write rc=$?
cd $R
python3 tools/pmc_summary.py $OUT/trace $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write > $OUT/summary.txt 2>&1
tail -5 $OUT/trace_stdout.log | cut -c1-600 >> $OUT/summary.txt
