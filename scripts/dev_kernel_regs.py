"""Dev (build container): VGPR / SGPR / LDS / scratch of every kernel in the built objects (from the code-object metadata)."""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
pat = sys.argv[1] if len(sys.argv) > 1 else "."
for obj in sorted(glob.glob(os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "csrc", os.environ.get("MIRRES_OBJ_DIR", "obj"), "*.o"))):
    tmp = tempfile.mkdtemp(); src = shutil.copy(obj, tmp)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", src], capture_output=True, cwd=tmp)
    for f in glob.glob(src + ".*gfx950"):
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            if re.search(pat, name):
                print("%-28s %-44s vgpr %4s agpr %3s sgpr %4s lds %6s scratch %5s wg %5s" % (os.path.basename(obj), re.sub(r"^_ZN2mr\d+", "", name)[:44], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
    shutil.rmtree(tmp)
