# A/B timing of builds of the HIP library: bash tools/ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  RT_HIP_LIB=$PWD/opengl_raytracer_amd/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$lib.json 2> gpurun_out/ab_$lib.err || echo FAILED
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$lib.json"))
print(round(d["value"],2), "Mrays/s", round(d["ms_per_step"],1), "ms", d["stage_ms_per_step_rank0"])
for k in d["kernels_rank0"]: print("  ", k["kernel"], round(k["avg_ms"],3), "ms x", k["launches"], " Tpairs/s", round(k["pairs_per_launch"]/k["avg_ms"]/1e9,2))
print("  ", d["config"]["slow_steps_per_step"], "pass rate", d["config"]["filter_pass_rate"])
PY
done
