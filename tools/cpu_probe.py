"""Times the CPU oracle step with a given thread count (used once to size bench.py's cpu_baseline)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

a = argparse.Namespace(n_class=101, ch=32, frames=int(sys.argv[2]) if len(sys.argv) > 2 else 48, k_sample=8)
n = int(sys.argv[1])
os.cpu_count = lambda: n
t0 = time.time()
print(n, bench.cpu_baseline(a), "total", round(time.time() - t0, 1), flush=True)
