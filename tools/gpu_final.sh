#!/bin/bash
# GPU box: the whole GPU suite, the bench lines of all five configurations, a rank's share of a tile split.
# usage: tools/gpu_final.sh TAG    (outputs under gpurun_out/TAG/)
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/gputest.log 2>&1
rc=$?
tail -3 $out/gputest.log
[ $rc -ne 0 ] && exit $rc
for w in soup64k_4k_p6 swordbunny_4k_p4 soup256k_64views_1080p sword_800_p2 bunny_1080p_p1; do
    timeout -k 10 300 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
    echo "== $w"; python tools/show_bench.py $out/bench_$w.json | grep -E "^value|moving"
done
for n in 2 4 8; do
    echo "== WORLD=$n, contexts bound to torch streams as bench.py binds them" >> $out/tile_shares.txt
    WORLD=$n INFLIGHT=1,4 timeout -k 10 200 python tools/diag_share_streams.py 2>/dev/null >> $out/tile_shares.txt || exit 1
done
cat $out/tile_shares.txt
