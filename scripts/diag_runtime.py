import sys, subprocess
code1 = """
import torch
print('torch first: avail', torch.cuda.is_available(), torch.cuda.device_count())
import ctypes
lib = ctypes.CDLL('bionumpy_amd/csrc/libbnpk.so')
print('bnpk count after torch', lib.bnpk_device_count())
"""
code2 = """
import ctypes
lib = ctypes.CDLL('bionumpy_amd/csrc/libbnpk.so')
print('bnpk count first', lib.bnpk_device_count())
import torch
print('torch after: avail', torch.cuda.is_available(), torch.cuda.device_count())
"""
for c in (code1, code2):
    r = subprocess.run([sys.executable, '-c', c], capture_output=True, text=True)
    print(r.stdout, r.stderr[-1500:])
