"""Quick throughput probe of the batched ECBS driver (not the contract bench; see bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import hl
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
agents = int(sys.argv[2]) if len(sys.argv) > 2 else 10
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 512
t0 = time.time()
insts = hl.generate_instances(1000 * agents, n_inst, 32, 32, 204, agents)
print("generated %d instances in %.2fs" % (n_inst, time.time() - t0), flush=True)
s = hl.BatchSolver(device=0, n_threads=threads, slots=slots, lds_nodes=int(os.environ.get('MRP_LDS_NODES', '0')),
                   arena_nodes=int(os.environ.get('MRP_ARENA_NODES', '0')),
                   _lib_path=os.environ.get('MRP_HL_LIB'))  # MRP_HL_LIB: A/B against another build of libmrp_hl.so
print("solver created %.2fs" % (time.time() - t0), flush=True)
cpu_n = int(sys.argv[5]) if len(sys.argv) > 5 else 0
mode = int(os.environ.get('MRP_HL_MODE', '0'))
for rep in range(int(os.environ.get('MRP_REPS', '3'))):
    s.ll_stats(reset=True)
    def _thr():
        try:
            return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat") if "throttled" in l}
        except OSError:
            return {}
    th0 = _thr()
    res, st = s.solve(insts, algo=hl.ECBS, w=1.3, want_paths=False, max_ll_expansions=int(os.environ.get("MRP_CAP", "50000")), mode=mode)
    ls = s.ll_stats()
    th1 = _thr()
    print("   cgroup throttling during the call: " + str({k: th1[k] - th0.get(k, 0) for k in th1}), flush=True)
    print("rep %d: wall %.3fs  solved %d/%d  LL exp %d  => %.3e exp/s, %.1f inst/s ; rounds %d searches %d ; kernel_ms(sum) %.1f launches %d migrated %d" % (
        rep, st["wall_seconds"], st["solved"], n_inst, st["ll_expansions"], st["ll_expansions"] / st["wall_seconds"],
        n_inst / st["wall_seconds"], st["rounds"], st["ll_searches"], ls["kernel_ms"], ls["launches"], ls["migrated"]), flush=True)
    print("   host thread-seconds: build %.3f  ll_call %.3f (pack %.3f unpack %.3f kernel %.3f h2d %.3f d2h %.3f)  consume %.3f" % (
        st["build_seconds"], st["ll_call_seconds"], ls["pack_ms"] / 1e3, ls["unpack_ms"] / 1e3, ls["kernel_ms"] / 1e3,
        ls["h2d_ms"] / 1e3, ls["d2h_ms"] / 1e3, st["consume_seconds"]), flush=True)
    print("   staged in pinned host memory: %.1f MB = %.0f bytes per search" % (ls["staged_bytes"] / 1e6, ls["staged_bytes"] / max(st["ll_searches"], 1)), flush=True)
    import collections
    print("   statuses: " + str(dict(collections.Counter(r["status"] for r in res))), flush=True)
    pf = ls["prof"]
    print("   kernel tiers: LDS %.2f us/expansion over %.4g expansions; beyond it (wide LDS tier / arena) %.2f us/expansion over %.4g expansions" % (
        pf[0] / 100.0 / max(pf[1], 1), pf[1], pf[2] / 100.0 / max(pf[3], 1), pf[3]), flush=True)
    print("   whole job on the device: %.1f us over %d jobs" % (pf[4] / 100.0 / max(pf[5], 1), pf[5]), flush=True)
    print("   resident workgroups: busy %.3f s, waiting %.3f s (sum over workgroups) -> busy fraction %.2f" % (
        ls["session_busy_ms"] / 1e3, ls["session_idle_ms"] / 1e3,
        ls["session_busy_ms"] / max(ls["session_busy_ms"] + ls["session_idle_ms"], 1e-9)), flush=True)
    print("   heavy workgroups: %d active, busy %.3f s, waiting %.3f s; fallbacks to one launch %d" % (
        ls["heavy_active_wgs"], ls["heavy_busy_ms"] / 1e3, ls["heavy_idle_ms"] / 1e3, ls["heavy_fallbacks"]), flush=True)

if cpu_n:
    import oracle
    t = 0.0; e = 0; mism = 0
    for inst, r in zip(insts[:cpu_n], res[:cpu_n]):
        o = oracle.mapf_solve(oracle.ECBS, inst, w=1.3, cap_total=50000, path_cap=2048)
        t += o["elapsed_ns"] / 1e9; e += o["ll_expanded"]
        if o["rc"] == 1 and (r["status"], r["cost"], r["hl_expanded"], r["ll_expanded"]) != (0, o["cost"], o["hl_expanded"], o["ll_expanded"]):
            mism += 1
    print("cpu oracle on first %d: %.3e exp/s, %.2f inst/s, mismatches %d" % (cpu_n, e / t, cpu_n / t, mism), flush=True)

try:
    print("peak host memory: " + [l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith("VmHWM")][0], flush=True)
except Exception:
    pass
