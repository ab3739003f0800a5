import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so.7")
class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                ("isManaged", C.c_int), ("allocationFlags", C.c_uint), ("pad", C.c_char * 64)]
hip.hipSetDevice(0)
x = np.zeros(256 << 20, np.uint8)
a = Attr()
rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(x.ctypes.data))
print("pageable numpy: rc", rc, "type", a.type)
t0 = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(x.ctypes.data), C.c_size_t(x.nbytes), 0); t1 = time.perf_counter()
print("hipHostRegister 256 MB: rc", rc, f"{(t1 - t0) * 1e3:.2f} ms")
rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(x.ctypes.data))
print("registered: rc", rc, "type", a.type)
t0 = time.perf_counter(); rc = hip.hipHostUnregister(C.c_void_p(x.ctypes.data)); t1 = time.perf_counter()
print("hipHostUnregister: rc", rc, f"{(t1 - t0) * 1e3:.2f} ms")
rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(x.ctypes.data))
print("after unregister: rc", rc, "type", a.type)
