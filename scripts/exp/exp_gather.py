"""gather_rows under different row layouts (what makes the entry gather of the filter config slower per byte?)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
n = 50_000_000
buf = HArray(dev=torch.randint(0, 255, (n * 320,), dtype=torch.uint8, device="cuda"))
def case(name, starts, lens):
    st, ln = HArray(host=starts.astype(np.int64)), HArray(host=lens.astype(np.int64))
    off, total = ops.row_offsets(ln, 1)
    o = ops.gather_rows(buf, st, off, len(lens), total, 0); del o
    torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
    for _ in range(3):
        o = ops.gather_rows(buf, st, off, len(lens), total, 0); del o
    torch.cuda.synchronize(); rep = dev.prof_report(); dev.prof_enable(False)
    ms = rep["gather_rows"]["total_ms"] / 3
    print(json.dumps({"case": name, "rows": len(lens), "out_gb": round(total / 1e9, 2), "ms": round(ms, 2), "out_gb_per_s": round(total / ms / 1e6, 1)}))
r = np.arange(n, dtype=np.int64)
case("316 B rows, all, stride 316", r * 316, np.full(n, 316))
keep = r[r % 3 != 0]
case("316 B rows, 2 of 3, stride 316", keep * 316, np.full(keep.size, 316))
case("316 B rows, 2 of 3, stride 320", keep * 320, np.full(keep.size, 316))
case("320 B rows, 2 of 3, stride 320", keep * 320, np.full(keep.size, 320))
case("150 B rows, all, stride 316", r * 316 + 10, np.full(n, 150))
case("150 B rows, 2 of 3, stride 316", keep * 316 + 10, np.full(keep.size, 150))
