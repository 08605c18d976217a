"""Experiment helper: writes a copy of radix.hip with cycle-counter marks after every barrier of the finishing
kernel's fast path (compiled only with -DFN_EXP_PHASES=<wave>).  usage: add_phase_marks.py <in> <out>"""
import sys, re
s = open(sys.argv[1]).read()
a = s.index("while (cur.b < n_buckets) {")
head, body = s[:a], s[a:]
body = body.replace("while (cur.b < n_buckets) {", "while (cur.b < n_buckets) {\n    FN_MARK(0);", 1)
parts = body.split("__syncthreads();")
# barriers in order inside the loop: [0] nb==0 path, [1] B1, [2] B2, [3] B3, [4] B4, then dup-path ones ..., B5 is the one followed by 'p_b = ' or 'nn_b = fn_uniform(sh[0])'
out = parts[0]
names = {1: 1, 2: 2, 3: 3, 4: 4}
for i, part in enumerate(parts[1:], start=0):
    mark = ""
    if i in names:
        mark = "\n      FN_MARK(%d);" % names[i]
    if part.lstrip().startswith("p_b = fn_uniform(sh[0]);") or part.lstrip().startswith("nn_b = fn_uniform(sh[0]);"):
        mark = "\n    FN_MARK(5);"
        out = out.rstrip()
        # mark 6 before the ticket hand-over line
        k = out.rfind("if (tid == 0) sh[0]")
        out = out[:k] + "FN_MARK(6);\n    " + out[k:] + "\n    "
    out += "__syncthreads();" + mark + part
body = out
body = body.replace("  if (have_prev) emit_prev();                          // the last bucket", "  if (lane == 0 && wave == FN_EXP_PHASES) { for (int i = 0; i < 7; ++i) atomicAdd(&state[FS_BUCKETS + n_buckets + 8 + i], ph_t[i]); }\n  if (have_prev) emit_prev();                          // the last bucket", 1)
body = body.replace("  if (have_prev) {                                     // the last bucket of this workgroup", "  if (lane == 0 && wave == FN_EXP_PHASES) { for (int i = 0; i < 7; ++i) atomicAdd(&state[FS_BUCKETS + n_buckets + 8 + i], ph_t[i]); }\n  if (have_prev) {                                     // the last bucket of this workgroup", 1)
head = head.replace("  unsigned* fmask32 = reinterpret_cast<unsigned*>(fmask);", """  unsigned* fmask32 = reinterpret_cast<unsigned*>(fmask);
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#define FN_MARK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_t[i] += now_ - ph_last; ph_last = now_; }""", 1)
s = head + body
s = s.replace("return FS_BUCKETS + std::max<int64_t>(n_buckets, 0) + 1; }", "return FS_BUCKETS + std::max<int64_t>(n_buckets, 0) + 1 + 16; }")
open(sys.argv[2], "w").write(s)
