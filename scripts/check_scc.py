"""Lint for a hipcc 7.2 miscompile met in round 3 (csrc/encode.hip, gather_rows_kernel): a wave-uniform select on a 64-bit
compare is emitted as V_CMP_*_{I,U}64 + S_CSELECT, and when the same compare also feeds a branch the copy of VCC into SCC
(S_AND_B64 vcc, exec, vcc) is dropped — the S_CSELECT then reads the carry / overflow bit of whatever scalar arithmetic
came last.  This script disassembles every kernel source and reports each S_CSELECT / S_CBRANCH_SCC whose SCC comes from
scalar ARITHMETIC (add, sub, shifts ...) while a 64-bit V_CMP sits between the two: the shape of the bug.  (A carry that
is really meant — 64-bit adds — is consumed by S_ADDC / S_SUBB, never by a select.)

    python scripts/check_scc.py          # exit status 1 if anything is reported
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bionumpy_amd", "csrc")
ARITH = re.compile(r"^\s*s_(add|sub|addc|subb|lshl|lshr|ashr|bfe|bcnt|min|max|abs|not|absdiff)\w*\s")
COMPARE = re.compile(r"^\s*s_(cmp|cmpk|bitcmp|and|or|xor|andn2|orn2|nand|nor|xnor)\w*\s")
USER = re.compile(r"^\s*s_(cselect|cbranch_scc)")
VCMP64 = re.compile(r"^\s*v_cmp_\w+_[iu]64")


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    found = 0
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        with tempfile.NamedTemporaryFile(suffix=".s") as out:
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out.name, src],
                           check=True, capture_output=True)
            lines = open(out.name).read().splitlines()
        last, last_i, func = None, -1, "?"
        for i, line in enumerate(lines):
            text = line.strip()
            if text.endswith(":"):
                if not text.startswith("."):
                    func = text[:-1]
                last = None                                  # (a label: SCC may come from another path)
                continue
            if USER.match(line):
                if last is not None and ARITH.match(last) and any(VCMP64.match(x) for x in lines[last_i + 1:i]):
                    found += 1
                    print("%s: %s line %d: `%s` reads the SCC of `%s` with a 64-bit V_CMP in between" %
                          (os.path.basename(src), func, i + 1, text, last.strip()))
                continue
            if ARITH.match(line) or COMPARE.match(line):
                last, last_i = line, i
    print("check_scc: %d suspicious select(s)" % found)
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
