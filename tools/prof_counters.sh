#!/bin/bash
# Collects rocprofv3 evidence for one frame of the default bench workload:
#   pass 0: kernel trace + stats;  passes 1..4: hardware counters, one --pmc set per run
# (counter runs carry no trace options).  Usage: bash tools/prof_counters.sh <tag> [workload]
tag=${1:-r02}
workload=${2:-soup64k_4k_p6}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# the kernels of a bounce pass one after the other, so that every launch has a duration of its own
cmd="python3 bench.py --workload $workload --steps 3 --warmup 1 --no-cpu-baseline --sweeps-in-turn --frames-in-flight 1"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $cmd > $out/bench_trace.json 2> $out/trace.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -o sq -- $cmd > $out/bench_sq.json 2> $out/sq.err || exit 2
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- $cmd > $out/bench_f.json 2> $out/f.err || exit 3
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- $cmd > $out/bench_w.json 2> $out/w.err || exit 4
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_misc -o m -- $cmd > $out/bench_m.json 2> $out/m.err || exit 5
find $out -name "*.csv" | head -30
