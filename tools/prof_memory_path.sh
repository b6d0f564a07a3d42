#!/bin/bash
# Counters of the vector memory path (address units, L1) per kernel of one frame: bash tools/prof_memory_path.sh <tag> [workload]
tag=${1:-mem}
workload=${2:-soup64k_4k_p6}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="python3 bench.py --workload $workload --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines --sweeps-in-turn --frames-in-flight 1"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/a -o a -- $cmd > $out/a.json 2> $out/a.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum --output-format csv -d $out/b -o b -- $cmd > $out/b.json 2> $out/b.err || exit 2
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $out/c -o c -- $cmd > $out/c.json 2> $out/c.err || exit 3
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $out/d -o d -- $cmd > $out/d.json 2> $out/d.err || exit 4
python3 tools/memory_path_summary.py $out
