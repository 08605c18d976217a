import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libvgpr24.so"))
for which in (24, 32, 24, 32):
    for n_waves, spins in ((1024, 2000), (16384, 200), (65536, 50), (262144, 10)):
        out = torch.full((n_waves * 64,), 77, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        rc = lib.run(which, C.c_void_p(out.data_ptr()), n_waves, spins, None)
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(n_waves, 64)
        bad_waves = np.flatnonzero(o.any(axis=1))
        print("alloc", which, "waves", n_waves, "spins", spins, "rc", rc, "waves with a changed register:", bad_waves.size, "first", bad_waves[:8].tolist(),
              "max regs changed per lane", int(o.max()))
