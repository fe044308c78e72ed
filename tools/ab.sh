#!/bin/bash
# A/B on the GPU box: tools/ab.sh "<bench args>" lib1 lib2 ...   (interleaved, 2 rounds)
args=$1; shift
for round in 1 2; do for lib in "$@"; do
  echo -n "$lib: "; RT_HIP_LIBNAME=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1),'Mrays/s', round(d['ms_per_step'],1),'ms/step', 'rays/sample',round(d['rays_per_sample'],3))" || exit 1
done; done
