#!/bin/bash
# GPU box: frame times of all five configurations (lights at rest and moving), without the CPU leg and the
# per-kernel figures.  usage: tools/gpu_quick5.sh TAG [bench args...]
tag=${1:-q5}; shift
out=gpurun_out/$tag
mkdir -p $out
for w in soup64k_4k_p6 swordbunny_4k_p4 soup256k_64views_1080p sword_800_p2 bunny_1080p_p1; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-rooflines "$@" > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
    echo "== $w"; python tools/show_bench.py $out/bench_$w.json | grep -E "^value|moving|one_frame"
done
