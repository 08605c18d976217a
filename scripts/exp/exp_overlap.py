"""Experiment: does a multi-threaded preadv into one page-locked buffer overlap with a hipMemcpyAsync out of another?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from concurrent.futures import ThreadPoolExecutor
from bionumpy_amd.io.pinned import PinnedBuffer
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, ptr
dev = Device.get()
N = 1 << 30
path = "/tmp/ovl.bin"
np.random.default_rng(0).integers(0, 255, size=N, dtype=np.uint8).tofile(path)
a, b = PinnedBuffer(N), PinnedBuffer(N)
d = torch.empty(N, dtype=torch.uint8, device="cuda")
f = open(path, "rb"); fd = f.fileno()
def read_into(buf, threads=16):
    view = memoryview(buf.array); step = N // threads
    def one(i):
        lo, hi = i * step, (i + 1) * step
        while lo < hi:
            lo += os.preadv(fd, [view[lo:hi]], lo)
    with ThreadPoolExecutor(threads) as p: list(p.map(one, range(threads)))
def upload(buf):
    lib.bnpk_copy_h2d_async(ptr(d), C.c_void_p(buf.ptr.value), N, None); torch.cuda.synchronize()
read_into(a); read_into(b); upload(a)
t = time.perf_counter(); read_into(a); tr = time.perf_counter() - t
t = time.perf_counter(); upload(b); tu = time.perf_counter() - t
t = time.perf_counter(); th = threading.Thread(target=read_into, args=(a,)); th.start(); upload(b); th.join(); tb = time.perf_counter() - t
print("read 1 GiB into pinned: %.1f ms; upload 1 GiB: %.1f ms; both at once: %.1f ms" % (tr * 1e3, tu * 1e3, tb * 1e3))
for threads in (4, 8, 32):
    t = time.perf_counter(); read_into(a, threads); print("  read with %d threads: %.1f ms" % (threads, (time.perf_counter() - t) * 1e3))
os.remove(path)
