#!/bin/bash
# GPU box: the whole GPU suite, then a bench line; outputs under gpurun_out/<tag>/.
# usage: tools/gpu_check.sh TAG [bench args...]
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/gputest.log 2>&1
rc=$?
tail -4 $out/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python tools/show_bench.py $out/bench.json
