import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vil_sensor_fusion_amd import Engine, EngineOpts
import ctypes
e0 = Engine(EngineOpts(windows=1, capacity=1192)); 
for w in (1, 6, 12, 48):
    for rep in range(3):
        t0 = time.perf_counter(); e = Engine(EngineOpts(windows=w, capacity=1192)); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
        print(f"windows {w:3d}: create {1e3 * (t1 - t0):7.2f} ms, destroy {1e3 * (t2 - t1):7.2f} ms")
hip = ctypes.CDLL("libamdhip64.so")
for mb in (1, 16, 140, 1300):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mb << 20)); t1 = time.perf_counter(); hip.hipMemset(p, 0, ctypes.c_size_t(mb << 20)); hip.hipDeviceSynchronize(); t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
    print(f"hipMalloc {mb:5d} MB: {1e3 * (t1 - t0):7.3f} ms, memset+sync {1e3 * (t2 - t1):7.3f} ms, free {1e3 * (t3 - t2):7.3f} ms")
