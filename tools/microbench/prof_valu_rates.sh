#!/bin/bash
# GPU box: what SQ_ACTIVE_INST_VALU counts per instruction kind (is a plain v_fma_f32 half as long as a
# v_pk_fma_f32 to the counter as well as to the clock?).  One --pmc run of the microbenchmark.
# usage: tools/microbench/prof_valu_rates.sh OUTDIR
out=${1:-gpurun_out/valu_pmc}
mkdir -p $out
export TMPDIR=/tmp
cd "$(dirname "$0")"
root=$(cd ../.. && pwd)
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $root/$out -o v -- ./valu_rates > $root/$out/run.txt 2> $root/$out/run.err || exit 1
cd $root
python3 tools/microbench/pmc_summary.py $(find $out -name "v_counter_collection.csv" | head -1)
