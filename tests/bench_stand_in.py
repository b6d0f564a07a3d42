"""bench.py with the CPU oracle standing in for the GPU renderer: lets the tests of the benchmark's launcher
(--gpus N starting its own ranks, the default sharding, the gather) run on a machine without a GPU.
Test infrastructure only -- bench.py itself cannot render on anything but the HIP library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle.rt_oracle import Oracle  # noqa: E402

if __name__ == "__main__":
    bench.main(stand_in=lambda width, height, threads: Oracle(width, height, threads=threads))
