#!/bin/bash
# PMC counter passes for the bench workload (one rocprofv3 run per pass; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Usage: tools/pmc.sh OUTDIR [bench args...]
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_IFETCH"
)
# PMC_PASSES="2 3": only those passes (FETCH_SIZE and WRITE_SIZE: the HBM traffic of tools/traffic.py)
i=0
for p in "${passes[@]}"; do
  if [ -n "$PMC_PASSES" ] && ! [[ " $PMC_PASSES " == *" $i "* ]]; then i=$((i+1)); continue; fi
  timeout -s KILL 180 rocprofv3 --pmc $p --output-format csv -d $R/$out/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also '' "$@" > $R/$out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/$out/pass$i.log; exit 1; }
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $R/$out
