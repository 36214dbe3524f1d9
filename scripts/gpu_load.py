"""A second process that keeps the GPU busy (the engine's own streaming copy kernel, k_copy16, in a loop): stretches the timing
of whatever runs beside it -- hand-overs between persistent tiles, sub-batch streams -- for scripts/soak_r05.sh.
    python scripts/gpu_load.py SECONDS [MiB per copy]"""
import ctypes as C
import sys
import time

sys.path.insert(0, "tests")
import lqr_ctypes as L

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = L.engine_api().lib
lib.lqrhip_copy_bandwidth.argtypes = [C.c_ulonglong, C.c_int, C.POINTER(C.c_double)]
g = C.c_double(0)
t_end = time.time() + secs
n = 0
while time.time() < t_end:
    lib.lqrhip_copy_bandwidth(mib << 20, 50, C.byref(g))
    n += 1
    time.sleep(0.002 * (n % 7))          # uneven duty cycle
print("gpu_load: %d bursts, last %.0f GB/s" % (n, g.value), flush=True)
