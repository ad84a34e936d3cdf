"""How long do large device / pinned allocations take on this stack, alone and side by side?  (ctypes on libamdhip64)"""
import ctypes as C, threading, time
hip = C.CDLL("libamdhip64.so")
def dev(gb):
    p = C.c_void_p(); t = time.time(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(int(gb * 2**30))); dt = time.time() - t
    t2 = time.time(); hip.hipMemset(p, 0xFF, C.c_size_t(int(gb * 2**30))); hip.hipDeviceSynchronize(); dm = time.time() - t2
    return p, dt, dm, rc
def pin(gb):
    p = C.c_void_p(); t = time.time(); rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(int(gb * 2**30)), 0); return p, time.time() - t, rc
hip.hipSetDevice(0); hip.hipFree(None)
for rep in range(2):
    p, dt, dm, rc = dev(16); print(f"hipMalloc 16 GiB: {dt*1e3:.0f} ms (rc {rc}), memset {dm*1e3:.0f} ms"); hip.hipFree(p)
    q, dt, rc = pin(4); print(f"hipHostMalloc 4 GiB: {dt*1e3:.0f} ms (rc {rc})"); hip.hipHostFree(q)
res = {}
def a(): res["dev"] = dev(16)
def b(): res["dev2"] = dev(8)
def c(): res["pin"] = pin(4)
ths = [threading.Thread(target=f) for f in (a, b, c)]
t = time.time(); [x.start() for x in ths]; [x.join() for x in ths]
print(f"side by side: hipMalloc 16 GiB {res['dev'][1]*1e3:.0f} ms, hipMalloc 8 GiB {res['dev2'][1]*1e3:.0f} ms, hipHostMalloc 4 GiB {res['pin'][1]*1e3:.0f} ms; wall {(time.time()-t)*1e3:.0f} ms")
# allocation right after large frees (what a second command-line run in one process does: its tables go back to the driver
# when the idle-block cache is over its limit, the next run allocates them again)
ps = [dev(17)[0], dev(17)[0]]
t = time.time(); [hip.hipFree(p) for p in ps]; print(f"hipFree 2 x 17 GiB: {(time.time()-t)*1e3:.0f} ms")
for rep in range(2):
    p, dt, dm, rc = dev(17); print(f"hipMalloc 17 GiB after the frees: {dt*1e3:.0f} ms (rc {rc}), memset {dm*1e3:.0f} ms")
    t = time.time(); hip.hipFree(p); print(f"hipFree: {(time.time()-t)*1e3:.0f} ms")
