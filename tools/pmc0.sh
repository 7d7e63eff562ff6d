#!/bin/bash
# single SQ pass: tools/pmc0.sh TAG [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/p0_$tag/pass0 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/p0_$tag.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/p0_$tag | grep -A12 path_kernel
