"""Where a tool process's start-up goes: loading the library, creating the context, the first allocation, the first kernels."""
import time, sys, os
t0=time.perf_counter()
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np
t1=time.perf_counter()
lib=C.CDLL(os.path.join(os.getcwd(),'ecckd_amd','libecckd_hip.so'))
t2=time.perf_counter()
lib.ecckd_init.restype=C.c_int; lib.ecckd_init.argtypes=[C.c_int,C.POINTER(C.c_void_p)]
ctx=C.c_void_p(); assert lib.ecckd_init(0, C.byref(ctx))==0
t3=time.perf_counter()
lib.ecckd_dev_alloc.restype=C.c_int; lib.ecckd_dev_alloc.argtypes=[C.c_void_p,C.c_size_t,C.POINTER(C.c_void_p)]
def alloc(nbytes):
    q=C.c_void_p(); assert lib.ecckd_dev_alloc(ctx, nbytes, C.byref(q))==0; return q
p=alloc(1<<20)
t4=time.perf_counter()
# first kernel: stable argsort of 1000 keys
n=1000
dk=alloc(n*8); dr=alloc(n*4)
b0=(C.c_int64*1)(0); b1=(C.c_int64*1)(n-1)
lib.ecckd_stable_argsort_bands_dev.argtypes=[C.c_void_p,C.c_size_t,C.c_void_p,C.c_int,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p]
rc=lib.ecckd_stable_argsort_bands_dev(ctx,n,dk,1,b0,b1,dr,None)
lib.ecckd_synchronize.argtypes=[C.c_void_p]; lib.ecckd_synchronize(ctx)
t5=time.perf_counter()
rc=lib.ecckd_stable_argsort_bands_dev(ctx,n,dk,1,b0,b1,dr,None); lib.ecckd_synchronize(ctx)
t6=time.perf_counter()
print("import numpy %.3f  dlopen %.3f  ecckd_init %.3f  first alloc %.3f  first kernels %.3f  again %.4f" % (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5))
