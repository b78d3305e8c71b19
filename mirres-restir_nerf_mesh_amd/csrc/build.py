"""Builds libmirres.so (gfx950) in-tree with hipcc. Incremental: one object per .hip, relinked when any object changes.

    python mirres-restir_nerf_mesh_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libmirres.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the traversal / reservoir decisions must not depend on the compiler's FMA choices (DESIGN.md §FP policy)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
         "-I", os.path.join(HERE, "..", "..", "include")]

# No packed-fp32 VALU code (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, formed by the SLP and the loop vectoriser) anywhere in the library.
# History, as far as it could be established (profiles/r03_pk_mfma_hazard.txt): in round 1, frames were corrupted when packed-fp32 kernels
# shared SIMDs with the material-net kernel of that time (f16-split MFMA) on mirres_render's second stream; removing the packed instructions from the neighbouring
# kernels made the frames clean. Round 3 could reproduce the corruption ONLY with that (since deleted) kernel: a stand-alone reproducer of packed fp32 beside MFMA
# is clean (0 wrong results in 1e11), and a packed build beside today's fp32-MFMA kernel renders 32 of 32 frames identically. The root cause was never found.
# The flags stay for a measured reason, not a hardware claim: a build WITH packed fp32 is 5-6 % slower on the frame (1014 vs 1070 Msamples/s, round 3), and the
# hash-grid encoder's two-step fp16 rounding is pinned against contraction choices either way. tests/test_abi.py checks that no object contains v_pk_*_f32.
# MIRRES_ALLOW_PK=1 (experiments only, with MIRRES_BUILD_TAG): leaves both vectorisers on, i.e. lets packed-fp32 instructions into the kernels again
PER_FILE = {f: ([] if os.environ.get("MIRRES_ALLOW_PK") == "1" else ["-fno-slp-vectorize", "-fno-vectorize"]) for f in ("passes.hip", "shading.hip", "bvh_trace.hip", "bvh_build.hip", "eaw.hip", "render.hip", "backward.hip", "normal.hip", "matnet.hip", "dump.hip", "raster.hip", "antialias.hip", "selfcheck.hip", "comm.hip")}
# matnet.hip: MFMA accumulators in VGPRs (no v_accvgpr_read between the layers of the register-chained MLP: -15 % VALU in k_mlp_mfma)
PER_FILE["matnet.hip"] += ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def _newer(a, deps):
    return (not os.path.exists(a)) or any(os.path.getmtime(d) > os.path.getmtime(a) for d in deps)


def build(force=False, verbose=True):
    # MIRRES_BUILD_TAG=<tag>: an experiment variant (with MIRRES_*_FLAGS) into ab/libmirres_<tag>.so with its own objects; load it with MIRRES_LIB
    tag = os.environ.get("MIRRES_BUILD_TAG", "")
    global OUT
    if tag:
        os.makedirs(os.path.join(HERE, "..", "..", "ab"), exist_ok=True)
        OUT = os.path.abspath(os.path.join(HERE, "..", "..", "ab", "libmirres_%s.so" % tag))
    srcs = sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "..", "include")) if f.endswith(".h")]
    objdir = os.path.join(HERE, "obj" + ("_" + tag if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        if force or _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + os.environ.get("MIRRES_EXTRA_FLAGS", "").split() + PER_FILE.get(os.path.basename(src), []) + (os.environ.get("MIRRES_MATNET_FLAGS", "").split() if os.path.basename(src) == "matnet.hip" else []) + (os.environ.get("MIRRES_TRACE_FLAGS", "").split() if os.path.basename(src) == "bvh_trace.hip" else []) + (os.environ.get("MIRRES_PASSES_FLAGS", "").split() if os.path.basename(src) == "passes.hip" else []) + ["-c", src, "-o", obj]
        if verbose:
            print("[mirres build]", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[mirres build] linked", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
