import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from cafe_amd import _lib
H=_lib.load()
rs=np.random.RandomState(1)
for n in (160000, 1200000, 4000000):
    xs=np.ascontiguousarray(rs.poisson(8.0,n).astype(np.int32))
    for la in (0,1):
        lam, sc, it, ps = C.c_double(), C.c_double(), C.c_int(), C.c_long()
        t0=time.time()
        H.cafehost_poisson_fit_selftest(xs.ctypes.data_as(C.POINTER(C.c_int32)), n, 0.5557, la, C.byref(lam), C.byref(sc), C.byref(it), C.byref(ps))
        print("n=%d lookahead=%d: %.4f s, %d sweeps, %d iterations, lambda %.12g" % (n, la, time.time()-t0, ps.value, it.value, lam.value))
