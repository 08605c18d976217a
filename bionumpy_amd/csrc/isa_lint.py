"""ISA lint of the built kernels (run by build.py after every build; ``python bionumpy_amd/csrc/isa_lint.py`` by hand).

Two checks over the disassembly (llvm-objdump) of every object file of the library, both born from wrong results that no
source-level review could have found:

1. **waits** — is every register that a memory load writes waited for before it is touched again?  gfx9 memory
   instructions return asynchronously; the compiler (SIInsertWaitcnts) has to put an ``s_waitcnt`` between a load and the
   first instruction that reads or overwrites its destination.  Re-derived here over the control-flow graph of every kernel
   with the counters' rules as the hardware documents them:
     vmcnt    global / buffer / flat / scratch loads AND stores, returned in issue order: ``vmcnt(N)`` = all but the N most
              recent have landed;
     lgkmcnt  LDS (in order among themselves) and scalar memory loads (out of order): with a scalar load pending only
              ``lgkmcnt(0)`` says anything about it; flat instructions count on both.
   State per program point = the list of pending operations (most recent last) with the registers each one writes; at a
   join the lists are merged position by position from the most recent end.  A finding = an instruction that names a
   register of a pending load.  (Mutation-tested: of the 30 ``s_waitcnt`` of one kernel, removing any of the 29 that guard
   a register is reported; the 30th orders LDS writes before a barrier.)
2. **SCC** — hipcc 7.2 was seen (round 3, gather_rows_kernel; round 4, finish_multi) to emit a wave-uniform select on a
   64-bit compare as V_CMP_*_{I,U}64 + S_CSELECT and to DROP the copy of VCC into SCC when the compare also feeds a branch:
   the select then reads the carry of whatever scalar arithmetic came last.  Reported: every S_CSELECT / S_CBRANCH_SCC whose
   SCC comes from scalar ARITHMETIC while a 64-bit V_CMP sits between the two.

It is a conservative model of two compiler passes, not a simulator: a hit is a place to read.  0 hits is what the build asserts.
"""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0|scc)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
        else:
            out.add((m.group(6), 0))
    return out


def first_operand(ops):
    return ops.split(",")[0] if ops else ""


class Ins:
    __slots__ = ("addr", "op", "ops", "line")

    def __init__(self, addr, op, ops, line):
        self.addr, self.op, self.ops, self.line = addr, op, ops, line


def parse_listing(text):
    """{kernel: [Ins or ('label', name)]} from llvm-objdump -d output or a .s file"""
    kernels, cur, name = {}, None, None
    for raw in text.splitlines():
        line = raw.split("//")[0].split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line.strip())
        if m:                                                # objdump: function or label
            lab = m.group(1)
            if not lab.startswith("L") and not lab.startswith(".L"):
                name, cur = lab, []
                kernels[name] = cur
            elif cur is not None:
                cur.append(("label", lab))
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:                                                # .s: label
            lab = m.group(1)
            if lab.startswith(".L") or lab.startswith("L"):
                if cur is not None:
                    cur.append(("label", lab))
            else:
                name, cur = lab, []
                kernels[name] = cur
            continue
        s = line.strip()
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        if not re.match(r"^(s_|v_|ds_|global_|buffer_|flat_|scratch_|tbuffer_|image_)", op):
            continue
        if cur is not None:
            cur.append(Ins(None, op, parts[1] if len(parts) > 1 else "", s))
    return kernels


def classify(ins):
    """(counter kinds this op increments, registers it writes asynchronously)"""
    op = ins.op
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_scratch_load"):
        return ("smem",), regs_of(first_operand(ins.ops))
    if op.startswith("s_store") or op.startswith("s_dcache") or op.startswith("s_atomic"):
        return ("smem",), set()
    if op.startswith("ds_"):
        writes = set()
        if "read" in op or "_rtn" in op or "bpermute" in op or "permute" in op or "swizzle" in op or "consume" in op or "append" in op or "ordered" in op or "load" in op:
            writes = regs_of(first_operand(ins.ops))
        return ("lds",), writes
    for pre in ("global_", "buffer_", "scratch_", "tbuffer_", "image_"):
        if op.startswith(pre):
            is_load = "load" in op or ("atomic" in op and ("sc0" in ins.ops or "glc" in ins.ops))
            lds_dst = " lds" in ins.ops                        # buffer_load ... lds: writes LDS, not a register
            return ("vm",), (regs_of(first_operand(ins.ops)) if is_load and not lds_dst else set())
    if op.startswith("flat_"):
        is_load = "load" in op or ("atomic" in op and ("sc0" in ins.ops or "glc" in ins.ops))
        return ("vm", "lds"), (regs_of(first_operand(ins.ops)) if is_load else set())
    return (), set()


def parse_wait(ins):
    """{'vm': n, 'lgkm': n} of an s_waitcnt (missing = no constraint)"""
    out = {}
    if ins.op == "s_waitcnt":
        for name, key in (("vmcnt", "vm"), ("lgkmcnt", "lgkm")):
            m = re.search(name + r"\((\d+)\)", ins.ops)
            if m:
                out[key] = int(m.group(1))
        if not out and re.match(r"^\s*(0x)?[0-9a-f]+\s*$", ins.ops):      # raw immediate
            imm = int(ins.ops.strip(), 0)
            out = {"vm": (imm & 0xf) | ((imm >> 14) & 3) << 4, "lgkm": (imm >> 8) & 0xf}
    elif ins.op == "s_waitcnt_vscnt":
        pass
    return out


def apply(state, ins, findings, kernel):
    """state = {'vm': [set, ...], 'lgkm': [(kind, set), ...]}  (most recent last)"""
    vm, lgkm = state
    w = parse_wait(ins)
    if w:
        if "vm" in w:
            vm = vm[len(vm) - w["vm"]:] if w["vm"] < len(vm) else vm
            if w["vm"] == 0:
                vm = []
        if "lgkm" in w:
            n = w["lgkm"]
            if n == 0:
                lgkm = []
            elif not any(k == "smem" for k, _ in lgkm):
                lgkm = lgkm[len(lgkm) - n:] if n < len(lgkm) else lgkm
        return (vm, lgkm)
    kinds, writes = classify(ins)
    named = regs_of(ins.ops)
    # implicit operands
    if ins.op.startswith("v_") and ("cndmask" in ins.op or "addc" in ins.op or "subb" in ins.op or ins.op.endswith("_e32") and ins.op.startswith("v_cmp")):
        named.add(("vcc", 0))
    pending = set()
    for regs in vm:
        pending |= regs
    for _, regs in lgkm:
        pending |= regs
    hit = named & pending
    if hit and writes:
        # a load whose DESTINATION is that of an older pending load of the same kind is no hazard: a counter's operations
        # return in issue order, the younger value lands last (software-pipelined loops re-use their registers that way)
        address = regs_of(",".join(ins.ops.split(",")[1:]))
        same_kind = set()
        if "vm" in kinds and "lds" not in kinds:
            for regs in vm:
                same_kind |= regs
        elif kinds == ("lds",):
            for k, regs in lgkm:
                if k == "lds":
                    same_kind |= regs
        other_kind = pending - same_kind
        hit = (hit & address) | (hit & other_kind)
    if hit:
        findings.append((kernel, ins.line, sorted(hit)))
        # (report once: treat them as landed from here on)
        vm = [r - hit for r in vm]
        lgkm = [(k, r - hit) for k, r in lgkm]
    if "vm" in kinds:
        vm = vm + [set(writes)]
    if "smem" in kinds:
        lgkm = lgkm + [("smem", set(writes))]
    elif "lds" in kinds:
        lgkm = lgkm + [("lds", set(writes))]
    return (vm[-64:], lgkm[-16:])


def merge(a, b):
    def m_vm(x, y):
        n = max(len(x), len(y))
        x = [set()] * (n - len(x)) + x
        y = [set()] * (n - len(y)) + y
        return [p | q for p, q in zip(x, y)]

    def m_lg(x, y):
        n = max(len(x), len(y))
        x = [("lds", set())] * (n - len(x)) + x
        y = [("lds", set())] * (n - len(y)) + y
        return [("smem" if "smem" in (p[0], q[0]) else "lds", p[1] | q[1]) for p, q in zip(x, y)]
    return (m_vm(a[0], b[0]), m_lg(a[1], b[1]))


def same(a, b):
    return a[0] == b[0] and a[1] == b[1]


def check_kernel(name, items):
    # basic blocks
    blocks, cur, labels = [], [], {}
    for it in items:
        if isinstance(it, tuple):
            if cur:
                blocks.append(cur)
                cur = []
            labels[it[1]] = len(blocks)
            continue
        cur.append(it)
        if it.op.startswith("s_cbranch") or it.op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    if not blocks:
        return []
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1]
        out = []
        target = None
        m = re.search(r"<([^>+]+)(\+0x[0-9a-f]+)?>|(\.?L[\w.$]+)", last.ops) if last.op.startswith("s_cbranch") or last.op == "s_branch" else None
        if m:
            target = labels.get(m.group(1) or m.group(3))
        if last.op == "s_branch":
            out = [target] if target is not None else []
        elif last.op.startswith("s_cbranch"):
            out = ([target] if target is not None else []) + ([i + 1] if i + 1 < len(blocks) else [])
        elif last.op in ("s_endpgm", "s_setpc_b64"):
            out = []
        else:
            out = [i + 1] if i + 1 < len(blocks) else []
        succ.append(out)
    entry = {0: ([], [])}
    work = [0]
    seen_findings = {}
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        i = work.pop()
        state = entry[i]
        findings = []
        for ins in blocks[i]:
            state = apply(state, ins, findings, name)
        for f in findings:
            seen_findings[(f[0], f[1])] = f
        for j in succ[i]:
            if j is None:
                continue
            if j not in entry:
                entry[j] = state
                work.append(j)
            else:
                m = merge(entry[j], state)
                if not same(m, entry[j]):
                    entry[j] = m
                    work.append(j)
    return list(seen_findings.values())



def device_listing(obj_path):
    """disassembly of the gfx950 code object inside a host object file built by hipcc (or of a .s listing / bare code object)"""
    if obj_path.endswith(".s"):
        return open(obj_path).read()
    llvm = os.path.dirname(OBJDUMP)
    tmp = "/tmp/bnpk_lint_%d" % os.getpid()
    try:
        co = obj_path
        r = subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + tmp + ".fb", obj_path],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(tmp + ".fb"):
            subprocess.run([BUNDLER, "--unbundle", "--type=o", "--input=" + tmp + ".fb",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + ".co"], check=True, capture_output=True)
            co = tmp + ".co"
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", co], capture_output=True, text=True,
                              check=True).stdout
    finally:
        for ext in (".fb", ".co"):
            if os.path.exists(tmp + ext):
                os.remove(tmp + ext)


# (S_MUL_I32 / S_MUL_HI_* do not write SCC: a select behind them still reads the compare in front of them)
ARITH = re.compile(r"^s_(add|sub|addc|subb|lshl|lshr|ashr|bfe|bcnt|min|max|abs|not|absdiff)\w*$")
COMPARE = re.compile(r"^s_(cmp|cmpk|bitcmp|and|or|xor|andn2|orn2|nand|nor|xnor)\w*$")
USER = re.compile(r"^s_(cselect|cbranch_scc)")
VCMP64 = re.compile(r"^v_cmp_\w+_[iu]64")


def check_scc(name, items):
    """S_CSELECT / S_CBRANCH_SCC that read the SCC of scalar arithmetic across a 64-bit vector compare"""
    found, last, since = [], None, []
    for it in items:
        if isinstance(it, tuple):                            # a label: SCC may come from another path
            last, since = None, []
            continue
        if USER.match(it.op):
            if last is not None and ARITH.match(last.op) and any(VCMP64.match(x.op) for x in since):
                found.append((name, it.line, "reads the SCC of `%s` with a 64-bit V_CMP in between" % last.line))
            continue
        if ARITH.match(it.op) or COMPARE.match(it.op):
            last, since = it, []
        else:
            since.append(it)
    return found


def kernels_with_24_vgprs(obj_path):
    """names of the kernels of an object whose code uses exactly 24 VGPRs (``.vgpr_count`` of the code object's metadata):
    the allocation with which rc_packed_kernel's unrolled form went wrong on gfx950 (common.h: BNPK_VGPR_FLOOR_32)"""
    llvm = os.path.dirname(OBJDUMP)
    tmp = "/tmp/bnpk_lint24_%d" % os.getpid()
    try:
        r = subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + tmp + ".fb", obj_path], capture_output=True)
        if r.returncode != 0:
            return []
        subprocess.run([BUNDLER, "--unbundle", "--type=o", "--input=" + tmp + ".fb", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--output=" + tmp + ".co"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", tmp + ".co"], capture_output=True, text=True).stdout
    finally:
        for ext in (".fb", ".co"):
            if os.path.exists(tmp + ext):
                os.remove(tmp + ext)
    out = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        a = int(blk.split()[0])
        name = re.search(r"\.symbol:\s+(\S+)\.kd", blk)
        if v and name and int(v.group(1)) == 24 and a == 0:
            out.append(name.group(1))
    return out


def lint_objects(paths, only=None, verbose=True):
    """(findings, kernels, instructions) over the device code of the given object files"""
    findings, n_kernels, n_ins = [], 0, 0
    for path in paths:
        kernels = parse_listing(device_listing(path))
        for name, items in kernels.items():
            if only and only not in name:
                continue
            if not any(not isinstance(it, tuple) for it in items):
                continue
            n_kernels += 1
            n_ins += sum(1 for it in items if not isinstance(it, tuple))
            for f in check_kernel(name, items):
                findings.append("%s: %s: `%s` touches %s while a load into it may be in flight"
                                % (os.path.basename(path), f[0], f[1], ", ".join("%s%d" % r for r in f[2])))
            for f in check_scc(name, items):
                findings.append("%s: %s: `%s` %s" % (os.path.basename(path), f[0], f[1], f[2]))
        if not path.endswith(".s"):
            for name in kernels_with_24_vgprs(path):
                if only and only not in name:
                    continue
                if "rocprim" in name:                        # (the library sort of the heavy-hitter fall-back: not ours to pad; said once)
                    if verbose:
                        print("isa_lint: note: %s uses exactly 24 VGPRs (third-party kernel)" % name[:90])
                    continue
                findings.append("%s: %s uses exactly 24 VGPRs: put BNPK_VGPR_FLOOR_32() first in its body (common.h)"
                                % (os.path.basename(path), name))
    if verbose:
        for f in findings:
            print(f)
        print("isa_lint: %d kernels, %d instructions, %d findings" % (n_kernels, n_ins, len(findings)))
    return findings, n_kernels, n_ins


def main(argv):
    import glob
    paths = [a for a in argv if not a.startswith("--")]
    only = None
    if "--kernel" in argv:
        only = argv[argv.index("--kernel") + 1]
        paths = [p for p in paths if p != only]
    if not paths:
        paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "*.o")))
    findings, _, _ = lint_objects(paths, only)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
