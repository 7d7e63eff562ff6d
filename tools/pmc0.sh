#!/bin/bash
# one PMC pass (wave / VALU counters) of the bench workload: tools/pmc0.sh OUTDIR [bench args...]
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$out/pass0 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pass0.log 2>&1 || { echo "pass failed"; tail -3 $R/$out/pass0.log; exit 1; }
