#!/bin/bash
# GPU box: the bench lines of all five configurations.  usage: tools/gpu_bench5.sh TAG   (gpurun_out/TAG/)
tag=${1:-b5}
out=gpurun_out/$tag
mkdir -p $out
for w in soup64k_4k_p6 swordbunny_4k_p4 soup256k_64views_1080p sword_800_p2 bunny_1080p_p1; do
    timeout -k 10 300 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
    echo "== $w"; python tools/show_bench.py $out/bench_$w.json | grep -E "^value|moving|^roofline "
done
