"""Experiment: how fast a plain file gets into HBM (bnpk_pread_parallel), by threads / piece size / where the file lies."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bionumpy_amd._native import lib, check
from bionumpy_amd.device import Device

dev = Device.get()
size = int(float(os.environ.get("GB", 4)) * (1 << 30))
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for where in ("/dev/shm", "/tmp"):
    path = os.path.join(where, "bnpk_feed.bin")
    block = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8).tobytes()
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        for _ in range(size // len(block)):
            f.write(block)
    print(where, "written in %.2f s" % (time.perf_counter() - t0))
    fd = os.open(path, os.O_RDONLY)
    host = torch.empty(size, dtype=torch.uint8).pin_memory()
    d = torch.empty(size, dtype=torch.uint8, device="cuda")
    hp, dp = host.data_ptr(), d.data_ptr()
    got = C.c_int64(0)
    stream = torch.cuda.current_stream().cuda_stream
    # the link alone
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d.copy_(host, non_blocking=True); torch.cuda.synchronize()
        link = size / (time.perf_counter() - t0) / 1e9
    print("  pinned -> HBM alone: %.1f GB/s" % link)
    for threads in (8, 16, 32, 64, 96):
        for piece_mb in (4, 16):
            for up in (False, True):
                best = 0
                for _ in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    check(lib.bnpk_pread_parallel(dev.ctx, fd, 0, C.c_void_p(hp), size, threads, piece_mb << 20,
                                                  C.c_void_p(dp) if up else None, C.c_void_p(stream), C.byref(got)))
                    torch.cuda.synchronize()
                    best = max(best, size / (time.perf_counter() - t0) / 1e9)
                print("  threads %3d piece %2d MB upload %-5s %.1f GB/s" % (threads, piece_mb, up, best))
    # mmap + memcpy by threads (numpy releases the GIL for big copies)
    import mmap
    m = mmap.mmap(fd, size, prot=mmap.PROT_READ)
    src = np.frombuffer(m, dtype=np.uint8)
    dst = host.numpy()
    from concurrent.futures import ThreadPoolExecutor
    for threads in (16, 32, 64):
        step = size // threads
        with ThreadPoolExecutor(threads) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda i: np.copyto(dst[i * step:(i + 1) * step], src[i * step:(i + 1) * step]), range(threads)))
            print("  mmap copy threads %d: %.1f GB/s" % (threads, size / (time.perf_counter() - t0) / 1e9))
    del src; m.close()
    os.close(fd); os.unlink(path)
    del host, d
