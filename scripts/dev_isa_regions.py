"""Dev (build container): instruction counts of one kernel by SOURCE REGION, from hipcc's own assembly with line tables.

    python scripts/dev_isa_regions.py bvh_trace.hip '<mangled-name substring>' [--loop N] [--flags "..."] name=lo-hi[,lo-hi] ...

The file is compiled to assembly with the library's flags + -gline-tables-only (the instruction stream is the shipped one: line tables do not change code
generation; the script checks the instruction count against a build without them). Every instruction is attributed to the last `.loc` of the kernel's own
source file seen before it (instructions inlined from headers go to the region of the call site that precedes them), then to the named line ranges.
--loop N restricts the count to the N-th largest loop body (0 = whole kernel; loops are found from backward branches to labels).
Classes: S = s_* (scalar, branch, waitcnt, nop), V = VALU split into arith (mul/add/sub/fma/min/max/cvt/rcp...) and select (v_cmp, v_cndmask, v_mov, integer / logic),
VMEM, LDS."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc", "-fno-slp-vectorize", "-fno-vectorize",
         "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def asm_of(src, extra, lines):
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + (["-gline-tables-only"] if lines else []) + [os.path.join(CSRC, src), "-o", "-"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return r.stdout.split("\n")


def kernel_body(asm, sub):
    st = [i for i, l in enumerate(asm) if re.match(r"^_Z\S*:", l) and sub in l]
    if not st:
        sys.exit("kernel not found: " + sub)
    st = st[0]
    en = next(i for i in range(st, len(asm)) if asm[i].startswith("\ts_endpgm"))
    return asm[st:en + 1]


def klass(op):
    if op.startswith("s_"): return "S"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("ds_"): return "LDS"
    if re.match(r"v_(fma|fmac|mul_f|add_f|sub_f|subrev_f|min|max|med|cvt|rcp|rsq|sqrt|div_|ldexp|frexp|mad_f|mac_f|trunc|floor|fract|rndne|exp|log)", op): return "Varith"
    if op.startswith("v_"): return "Vsel"
    return "other"


def main():
    args = sys.argv[1:]
    src, sub = args[0], args[1]
    loop_n, extra, regions = 0, [], []
    i = 2
    while i < len(args):
        if args[i] == "--loop": loop_n = int(args[i + 1]); i += 2
        elif args[i] == "--flags": extra = args[i + 1].split(); i += 2
        else:
            name, rs = args[i].split("=")
            regions.append((name, [tuple(int(x) for x in r.split("-")) for r in rs.split(",")])); i += 1
    body = kernel_body(asm_of(src, extra, True), sub)
    plain = kernel_body(asm_of(src, extra, False), sub)
    is_ins = lambda l: l.startswith("\t") and not l.startswith(("\t.", "\t;")) and l.strip()
    n_plain = sum(1 for l in plain if is_ins(l))
    # instruction list with attribution
    ins, cur_line, labels = [], 0, {}
    for l in body:
        if l.startswith("\t.loc"):
            # the OUTERMOST frame of the inline stack in hipcc's comment = the line of the kernel's own body (`; hdr.hpp:12:3 @[ file.hip:547:28 ]`)
            m = re.findall(re.escape(src) + r":(\d+):", l)
            if m and int(m[-1]) > 0: cur_line = int(m[-1])
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m: labels[m.group(1)] = len(ins); continue
        if is_ins(l):
            parts = l.strip().split(None, 1)
            ins.append((parts[0], parts[1] if len(parts) > 1 else "", cur_line))
    print("%s: %d instructions (%d in the build without line tables)" % (sub, len(ins), n_plain))
    lo, hi = 0, len(ins) - 1
    if loop_n:
        loops = []
        for k, (op, a, _) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                t = labels.get(a.strip().split()[-1])
                if t is not None and t <= k: loops.append((t, k))
        # outermost-distinct loops by header, largest first
        by_head = {}
        for t, k in loops: by_head[t] = max(by_head.get(t, k), k)
        loops = sorted(by_head.items(), key=lambda x: x[0] - x[1])
        lo, hi = loops[loop_n - 1]
        print("loop %d of %d: instructions %d..%d (%d)" % (loop_n, len(loops), lo, hi, hi - lo + 1))
    tot = collections.Counter(); per = collections.OrderedDict((n, collections.Counter()) for n, _ in regions); per["(other)"] = collections.Counter()
    ops = collections.OrderedDict((n, collections.Counter()) for n in per)
    for op, a, ln in ins[lo:hi + 1]:
        c = klass(op); tot[c] += 1
        r = next((n for n, rs in regions if any(a_ <= ln <= b_ for a_, b_ in rs)), "(other)")
        per[r][c] += 1; ops[r][re.sub(r"_e32$|_e64$", "", op)] += 1
    cols = ["S", "Varith", "Vsel", "VMEM", "LDS", "other"]
    print("%-22s" % "region" + "".join("%8s" % c for c in cols) + "%8s" % "all")
    for n, c in per.items():
        print("%-22s" % n + "".join("%8d" % c[k] for k in cols) + "%8d" % sum(c.values()))
    print("%-22s" % "total" + "".join("%8d" % tot[k] for k in cols) + "%8d" % sum(tot.values()))
    print("VALU = Varith + Vsel = %d" % (tot["Varith"] + tot["Vsel"]))
    if os.environ.get("ISA_OPS"):
        for n, c in ops.items():
            print("  [%s] " % n + "  ".join("%s %d" % kv for kv in c.most_common(14)))


if __name__ == "__main__":
    main()
