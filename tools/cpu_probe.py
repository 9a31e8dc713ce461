"""Host CPU capacity of the GPU box as the oracle sees it: cgroup quota, load, and the SCvx twin's throughput against the thread count
(oracle_scvx_run_batch).  usage: python tools/cpu_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import oracle_lib
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | grep -i 'model name\\|socket\\|core(s)\\|thread(s)\\|numa node(s)'")
lib = oracle_lib.lib(); cfg = oracle_lib.CONFIG_ROOT.encode()
def batch(n, th):
    c = (C.c_longlong * 4)(); t0 = time.time()
    lib.oracle_scvx_run_batch(cfg, 50, C.c_ulonglong(20260927), C.c_ulonglong(0), n, 1, th, c)
    return time.time() - t0, list(c)
print("1 thread:", batch(1, 1))
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    dt, c = batch(2 * th, th)
    print("threads %3d: %4d instances in %6.2f s = %6.1f /s (%.2f per thread-second), converged %d" % (th, 2 * th, dt, 2 * th / dt, 2 / dt, c[0]), flush=True)
