#!/bin/bash
# here (no GPU): rebuilds what travels to the GPU box, then runs the command there through gpurun
# usage: tools/gpu.sh TIMEOUT 'command'
set -e
cd "$(dirname "$0")/.."
make -s -C opengl_raytracer_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
make -s -C opengl_raytracer_amd/csrc/host >/dev/null
make -s -C oracle >/dev/null
make -s -C opengl_raytracer_amd/csrc/demo >/dev/null
make -s -C tests/native >/dev/null
gpurun --timeout "$1" -- "$2"
