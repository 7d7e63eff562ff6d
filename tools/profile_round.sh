#!/bin/bash
# The round's profile set (run on the GPU box through gpurun): tools/profile_round.sh TAG
#   gpurun_out/TAG_bench.json                     python bench.py (C2 headline + C4 under "also")
#   gpurun_out/TAG_{c2,c4,c5}_kstats.txt          rocprofv3 --kernel-trace --stats: per-kernel totals + last frame's launches
#   gpurun_out/TAG_{c2,c4}_pmc.json/_levels.txt   rocprofv3 --pmc passes (tools/pmc.sh), per kernel and per launch
#   gpurun_out/TAG_traffic.json                   HBM bytes per rt_render from the FETCH_SIZE / WRITE_SIZE passes
tag=$1
set -o pipefail
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
for w in c3 c5; do
  timeout -k 10 300 python bench.py --workload $w --also "" --steps 5 > gpurun_out/${tag}_bench_$w.json 2>> gpurun_out/${tag}_bench.err || exit 1
done
# the multi-GPU workload of BASELINE configs[4] on one rank (frames mode: the accumulate step is inside the timed region)
timeout -k 10 300 python bench.py --workload c5 --mode frames --also "" > gpurun_out/${tag}_bench_c5_frames.json 2>> gpurun_out/${tag}_bench.err || exit 1
for w in c2 c4 c5; do
  extra="--workload $w"; [ $w = c5 ] && extra="$extra --steps 2"
  rm -rf gpurun_out/ks_${tag}_$w
  # the bench's own command (two frames in flight: the last "frame" listed interleaves two) ...
  bash tools/kstats.sh ${tag}_$w $extra > gpurun_out/${tag}_${w}_kstats.txt 2>&1 || exit 1
  cp gpurun_out/ks_${tag}_$w/*/*_kernel_stats.csv gpurun_out/${tag}_${w}_kernel_stats.csv
  # ... and one frame after the other, for the launch-by-launch table
  rm -rf gpurun_out/ks_${tag}_${w}_serial
  bash tools/kstats.sh ${tag}_${w}_serial $extra --inflight 1 > gpurun_out/${tag}_${w}_kstats_serial.txt 2>&1 || exit 1
done
timeout -k 10 300 python tools/band_share.py all > gpurun_out/${tag}_band_share.txt 2>&1 || exit 1
timeout -k 10 300 python tools/recommit.py > gpurun_out/${tag}_recommit.txt 2>&1 || exit 1
for w in c2 c4; do
  rm -rf gpurun_out/pmc_${tag}_$w
  bash tools/pmc.sh gpurun_out/pmc_${tag}_$w --workload $w --inflight 1 > gpurun_out/${tag}_${w}_pmc.txt 2>&1 || exit 1
  cp gpurun_out/pmc_${tag}_$w/summary.json gpurun_out/${tag}_${w}_pmc.json
  python3 tools/pmc_levels.py gpurun_out/pmc_${tag}_$w > gpurun_out/${tag}_${w}_pmc_levels.txt
  python3 tools/traffic.py gpurun_out/pmc_${tag}_$w $w 3 > gpurun_out/${tag}_${w}_traffic.json
done
for w in c3 c5; do   # HBM traffic only
  rm -rf gpurun_out/pmc_${tag}_$w
  PMC_PASSES="2 3" bash tools/pmc.sh gpurun_out/pmc_${tag}_$w --workload $w --inflight 1 > gpurun_out/${tag}_${w}_pmc.txt 2>&1 || exit 1
  python3 tools/traffic.py gpurun_out/pmc_${tag}_$w $w 3 > gpurun_out/${tag}_${w}_traffic.json
done
python3 tools/traffic.py merge gpurun_out/${tag}_c2_traffic.json gpurun_out/${tag}_c4_traffic.json gpurun_out/${tag}_c3_traffic.json gpurun_out/${tag}_c5_traffic.json > gpurun_out/${tag}_traffic.json
# the bench line again, now that profiles/traffic.json can be of this build: copy gpurun_out/${tag}_traffic.json there first
echo done
