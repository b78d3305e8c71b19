"""Dev: disassembly of one kernel's address range.  python scripts/dev_isa_dump.py bvh_trace.o <kernel substring> <lo hex> <hi hex>"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = "/opt/rocm/lib/llvm/bin/llvm-objdump"
tmp = tempfile.mkdtemp(); src = shutil.copy(os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "csrc", "obj", sys.argv[1]), tmp)
subprocess.run([OBJ, "--offloading", src], capture_output=True, cwd=tmp)
asm = "".join(subprocess.run([OBJ, "-d", f], capture_output=True, text=True).stdout for f in glob.glob(src + ".*gfx950"))
shutil.rmtree(tmp)
lines = asm.split("\n")
st = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*" + re.escape(sys.argv[2]), l)][0]
en = next((i for i in range(st + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
lo, hi = int(sys.argv[3], 16), int(sys.argv[4], 16)
for l in lines[st:en]:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", l)
    if m and lo <= int(m.group(3), 16) <= hi:
        print("%05x  %-28s %s" % (int(m.group(3), 16), m.group(1), m.group(2)))
