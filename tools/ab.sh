# A/B timing of builds of the HIP library: bash tools/ab.sh lib1.so lib2.so ...   (GPU box)
mkdir -p gpurun_out/ab
for lib in "$@"; do
  echo "== $lib"
  RT_HIP_LIB=$PWD/opengl_raytracer_amd/$lib timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/$lib.json 2> gpurun_out/ab/$lib.err || { echo FAILED; tail -3 gpurun_out/ab/$lib.err; continue; }
  python tools/show_bench.py gpurun_out/ab/$lib.json | grep -E "^value|one_frame|  kernel"
done
