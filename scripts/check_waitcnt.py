#!/usr/bin/env python
"""CLI of bionumpy_amd/csrc/isa_lint.py (the ISA lint the build runs): object files, .s listings or nothing (= every object of
the library);  --kernel SUBSTR restricts the kernels."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "bionumpy_amd", "csrc", "isa_lint.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

if __name__ == "__main__":
    sys.exit(lint.main(sys.argv[1:]))
