"""How many host CPUs does a job on the GPU box really get?  (dev probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/self/cgroup"):
    try:
        print(f, "->", open(f).read().strip().replace("\n", " | ")[:300])
    except OSError as e:
        print(f, "->", type(e).__name__)
import oracle
from libmultirobotplanning_amd import hl
ia = hl.generate_instances(10000, 2048, 32, 32, 204, 10)
for nt in (1, 8, 16, 32, 64, 128, 256):
    n = min(2048, 128 * nt)
    t0 = time.perf_counter()
    per, wall = oracle.mapf_solve_batch(oracle.ECBS, ia.dimx, ia.dimy, ia.obstacles[:n], ia.starts[:n], ia.goals[:n], w=1.3, cap_total=50000, n_threads=nt)
    print("threads %3d: %d instances, pool wall %.3f s -> %.3e exp/s wall-clock (sum of per-instance search time %.3f s)" % (
        nt, n, wall, per[:, 4].sum() / wall, per[:, 5].sum() / 1e9), flush=True)
