#!/bin/bash
# The round's profile set on the GPU box: kernel trace of the default bench command, then — each in runs of their own —
# the traffic passes for the headline workload (f64), its f32 peer, C1, C3, C4 and the 8 M-sphere scene, and the SQ/TCC/TCP
# passes of the three BVH-bound configs and of the f64 kernel.  usage: tools/profile_round.sh <tag>
set -o pipefail
tag=${1:-r03}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $PWD/bench.py --no-cpu-baseline > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
tail -c 600 $out/trace.log
A=1.7777777777777777
bash tools/pmc_traffic.sh ${tag}_cornell_f64 "cornell_box 1024x1024, 4096 spp, max depth 50, f64" --precision f64 &&
bash tools/pmc_traffic.sh ${tag}_cornell "cornell_box 1024x1024, 4096 spp, max depth 50" --precision f32 &&
bash tools/pmc_traffic.sh ${tag}_c1 "random_scene 400x225, 100 spp, max depth 50" --precision f32 --scene random_scene --width 400 --aspect $A --spp 100 &&
bash tools/pmc_traffic.sh ${tag}_c3 "final_scene 1024x1024, 512 spp, max depth 50" --precision f32 --scene final_scene --width 1024 --aspect 1 --spp 512 &&
bash tools/pmc_traffic.sh ${tag}_c4 "bvh_stress 3840x2160, 4 spp, max depth 50, n=1000000" --precision f32 --scene bvh_stress --width 3840 --aspect $A --spp 4 --n 1000000 &&
bash tools/pmc_traffic.sh ${tag}_c4hbm "bvh_stress 1920x1080, 8 spp, max depth 50, n=8000000" --precision f32 --scene bvh_stress --width 1920 --aspect $A --spp 8 --n 8000000 &&
bash tools/pmc_traffic.sh ${tag}_c4sparse "bvh_stress_sparse 1920x1080, 8 spp, max depth 50, n=8000000" --precision f32 --scene bvh_stress_sparse --width 1920 --aspect $A --spp 8 --n 8000000 || exit 1
[ -n "$SKIP_WALK" ] && exit 0
bash tools/pmc_walk.sh ${tag}_c1 0 0 400 100 > $out/walk_c1.txt 2>&1 &&
bash tools/pmc_walk.sh ${tag}_c3 8 0 1024 512 1.0 > $out/walk_c3.txt 2>&1 &&
bash tools/pmc_walk.sh ${tag}_c4 7 1000000 3840 4 > $out/walk_c4.txt 2>&1 &&
bash tools/pmc_walk.sh ${tag}_f64 6 0 1024 1024 1.0 f64 > $out/walk_f64.txt 2>&1
rc=$?
for f in $out/walk_c*.txt $out/walk_f64.txt; do tail -n 8 $f; done
exit $rc
