set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_trace $R/gpurun_out/prof_pmc1 $R/gpurun_out/prof_pmc2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -o bench -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-rows 0 > $R/gpurun_out/prof_trace/stdout.log 2>&1
echo trace rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_pmc1 -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-kernel-timing > $R/gpurun_out/prof_pmc1/stdout.log 2>&1
echo pmc1 rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof_pmc2 -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-kernel-timing > $R/gpurun_out/prof_pmc2/stdout.log 2>&1
echo pmc2 rc=$?
ls -R $R/gpurun_out/prof_trace | head -30
