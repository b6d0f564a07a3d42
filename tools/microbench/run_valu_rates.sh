#!/bin/bash
# Builds valu_rates.hip, lists what the compiler emitted per row, and (on a GPU box) runs it.
# usage: tools/microbench/run_valu_rates.sh [build|run]   (run expects the binary built beforehand)
set -e
cd "$(dirname "$0")"
if [ "${1:-build}" = build ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o valu_rates valu_rates.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -S --cuda-device-only -o valu_rates.s valu_rates.hip
    python3 asm_counts.py valu_rates.s > valu_rates_asm_counts.txt
    cat valu_rates_asm_counts.txt
else
    cat valu_rates_asm_counts.txt
    timeout -k 10 120 ./valu_rates
fi
