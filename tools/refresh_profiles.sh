#!/bin/bash
# After tools/prof_counters.sh <tag> soup64k_4k_p6 and <tag>sb swordbunny_4k_p4 have run on the GPU box:
# writes the summaries, stats tables and traced bench lines into profiles/ under the round's name.
# usage: tools/refresh_profiles.sh r03
set -e
tag=${1:-r03}
python tools/summarize_counters.py gpurun_out/prof_$tag profiles/${tag}_rocprof_summary_soup64k_4k_p6.txt soup64k_4k_p6
python tools/summarize_counters.py gpurun_out/prof_${tag}sb profiles/${tag}_rocprof_summary_swordbunny_4k_p4.txt swordbunny_4k_p4
cp gpurun_out/prof_$tag/trace/t_kernel_stats.csv profiles/${tag}_kernel_stats_soup64k_4k_p6_in_turn.csv
cp gpurun_out/prof_${tag}sb/trace/t_kernel_stats.csv profiles/${tag}_kernel_stats_swordbunny_4k_p4_in_turn.csv
tail -1 gpurun_out/prof_$tag/bench_trace.json > profiles/${tag}_bench_soup64k_4k_p6_in_turn_under_rocprof.json
