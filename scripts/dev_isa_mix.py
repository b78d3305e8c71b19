"""Dev (build container): instruction mix of the loops of one kernel, from the built object.

    python scripts/dev_isa_mix.py passes.o k_initial_gen [number of loops, innermost-sized first]"""
import collections, glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
obj, kern = sys.argv[1], sys.argv[2]
nloops = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tmp = tempfile.mkdtemp()
src = shutil.copy(os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "csrc", "obj", obj), tmp)
subprocess.run([OBJDUMP, "--offloading", src], capture_output=True, cwd=tmp)
asm = "".join(subprocess.run([OBJDUMP, "-d", f], capture_output=True, text=True).stdout for f in glob.glob(src + ".*gfx950"))
shutil.rmtree(tmp)
lines = asm.split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*%s" % re.escape(kern), l)]
if not starts:
    sys.exit("kernel not found")
start = starts[0]
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
print(lines[start][:150])
ins = []
for l in lines[start:end]:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
print(len(ins), "instructions in the kernel")


def klass(op):
    if op.startswith(("v_fma", "v_fmac")): return "v_fma/fmac"
    if re.match(r"v_(mul|add|sub|subrev)_f32", op): return "v_mul/add/sub_f32"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): return "v_transcendental"
    if op.startswith("v_div_"): return op
    if re.match(r"v_(min|max|med)", op): return "v_min/max"
    if op.startswith("v_cvt") or re.match(r"v_(floor|fract|trunc|rndne|ceil|ldexp|frexp)", op): return "v_cvt/round"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirst", "v_writelane", "v_swap")): return "v_mov"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_"): return "salu/branch/wait"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith("v_"): return "v_int/logic"
    return op


loops = []
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"(\d+)\s*$", args)
        if m:
            off = int(m.group(1))
            if off > 32767: off -= 65536
            tgt = a + 4 + off * 4
            if tgt < a: loops.append((tgt, a))
loops.sort(key=lambda x: x[1] - x[0])
for tgt, a in loops[:nloops]:
    n = [i for i in ins if tgt <= i[0] <= a]
    c = collections.Counter(klass(op) for _, op, _ in n)
    print("loop %x..%x: %d instructions" % (tgt, a, len(n)))
    print("    " + "  ".join("%s %d" % kv for kv in c.most_common()))
