"""Builds libbnpk.so (the C-ABI of include/bnpk.h) for gfx950 with hipcc, in-tree.

    python bionumpy_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU; one object per .hip file so that an edit rebuilds one file.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libbnpk.so")
OBJ_DIR = os.path.join(HERE, "build")
SOURCES = ["api.hip", "scan.hip", "decode.hip", "multiline.hip", "fastq.hip", "encode.hip", "kmers.hip", "revcomp.hip", "rowops.hip", "radix.hip", "finish.hip", "finish_dup.hip", "finish_wave.hip", "finish_multi.hip", "finish_small.hip", "merge.hip", "count.hip", "sparse.hip", "synth.hip", "collectives.hip"]
LINT = os.path.join(HERE, "isa_lint.py")
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "scan.h"), os.path.join(HERE, "rows.h"),
           os.path.join(HERE, "kmer_gen.h"), os.path.join(HERE, "finish.h"),
           os.path.join(ROOT, "include", "bnpk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"] + os.environ.get("BNPK_HIPCC_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def _unit_hash(src):
    """hash of what one object file is built from: its source, every header, the flags"""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(HERE, src)] + HEADERS:
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(hipcc, src):
    """(object path, whether it was rebuilt); an object is current iff the stamp next to it carries the hash of its inputs
    (content, not mtimes: a snapshot copied without mtimes must never leave a stale object in the library)"""
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(HERE, src)
    want = _unit_hash(src)
    stamp = obj + ".sha256"
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj, False
    cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return obj, True


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(HERE, s) for s in SOURCES] + HEADERS:
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    # content hash, not mtimes: the snapshot shipped to the GPU box does not keep mtimes
    stamp = LIB + ".sha256"
    digest = _source_hash()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        if verbose:
            print("up to date", LIB)
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        built = list(ex.map(lambda s: _compile(hipcc, s), SOURCES))
    objs = [o for o, _ in built]
    if True:                                              # the digest differs from the library's stamp: always relink
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if os.environ.get("BNPK_SKIP_ISA_LINT", "0") == "0":
        _lint(objs, verbose)                             # (raises: a library with a finding gets no stamp and is rebuilt next time)
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    if verbose:
        print("built", LIB)
    return LIB


def _lint(objs, verbose):
    """the ISA lint over the device code of every object (isa_lint.py: a load's destination touched before its s_waitcnt;
    an S_CSELECT on the SCC of scalar arithmetic across a 64-bit V_CMP) — a finding fails the build (BNPK_LINT=warn: it is
    printed and the library is kept; BNPK_LINT=off, or the LLVM tools of the ROCm image absent: not run, said so)"""
    import importlib.util
    mode = os.environ.get("BNPK_LINT", "strict").lower()
    spec = importlib.util.spec_from_file_location("bnpk_isa_lint", os.path.join(HERE, "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    if mode == "off" or not (os.path.exists(lint.OBJDUMP) and os.path.exists(lint.BUNDLER)):
        print("isa_lint: not run (%s)" % ("BNPK_LINT=off" if mode == "off" else "llvm-objdump / clang-offload-bundler not found"))
        return
    findings, n_kernels, n_ins = lint.lint_objects(objs, verbose=False)
    if findings and mode == "warn":
        print("isa_lint: WARNING, %d finding(s) in the built kernels:\n%s" % (len(findings), "\n".join(findings)))
        return
    if findings:
        raise RuntimeError("isa_lint: %d finding(s) in the built kernels:\n%s" % (len(findings), "\n".join(findings)))
    if verbose:
        print("isa_lint: %d kernels, %d instructions, 0 findings" % (n_kernels, n_ins))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
