#!/bin/bash
# The round's profile set on the GPU box: kernel trace of the default bench command, then — each in runs of their own —
# the traffic passes for the headline workload and for C1, C3, C4, and the SQ/TCC/TCP passes of the three BVH-bound
# configs.  usage: tools/profile_round.sh <tag>
set -o pipefail
tag=${1:-r02}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $PWD/bench.py --no-cpu-baseline > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
tail -c 600 $out/trace.log
bash tools/pmc_traffic.sh ${tag}_cornell "cornell_box 1024x1024, 4096 spp, max depth 50" &&
bash tools/pmc_traffic.sh ${tag}_c1 "random_scene 400x225, 100 spp, max depth 50" --scene random_scene --width 400 --aspect 1.7777777777777777 --spp 100 &&
bash tools/pmc_traffic.sh ${tag}_c3 "final_scene 1024x1024, 512 spp, max depth 50" --scene final_scene --width 1024 --aspect 1 --spp 512 &&
bash tools/pmc_traffic.sh ${tag}_c4 "bvh_stress 3840x2160, 4 spp, max depth 50, n=1000000" --scene bvh_stress --width 3840 --aspect 1.7777777777777777 --spp 4 --n 1000000 || exit 1
[ -n "$SKIP_WALK" ] && exit 0
bash tools/pmc_walk.sh ${tag}_c1 0 0 400 100 > $out/walk_c1.txt 2>&1 &&
bash tools/pmc_walk.sh ${tag}_c3 8 0 1024 512 1.0 > $out/walk_c3.txt 2>&1 &&
bash tools/pmc_walk.sh ${tag}_c4 7 1000000 3840 4 > $out/walk_c4.txt 2>&1
rc=$?
for f in $out/walk_c*.txt; do tail -n 8 $f; done
exit $rc
