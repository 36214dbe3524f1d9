import sys, os; sys.path.insert(0,"tests")
import lqr_ctypes as L
L.ENGINE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", "liblqr-hip-timing.so")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
e = L.engine_api()
img = bench.make_image(3840, 2160, 100)
c = L.Carver(e, img).configure()
assert c.resize(3840-6, 2160) == 1
