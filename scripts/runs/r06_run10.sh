#!/bin/bash
# round 6, run 10: the occluder hint of the pixel-pair shadow rays (MIRRES_OCCL_CACHE=0/1): frame hashes, frames, how many rays the hint answers
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_occl_cache.txt
one() { python3 bench.py --mesh $1 --no-extras --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
{ echo "# occluder hint (engine.hpp RaySrc::occl), csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  echo "== frame hashes (8 spp; second frame of a process too: the hint then carries over)"
  for mesh in icosphere clustered; do for c in 0 1; do echo "$mesh cache=$c  $(MIRRES_MESH=$mesh MIRRES_OCCL_CACHE=$c timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; done; done
  echo "== frames, 128 spp, interleaved"
  for mesh in icosphere clustered; do for i in 1 2; do for c in 0 1; do echo "$mesh cache=$c  $(MIRRES_OCCL_CACHE=$c one $mesh)"; done; done; done
  echo "== rays answered by the hint (build with -DMR_OCCL_COUNT=1), one 32-spp frame"
  for mesh in icosphere clustered; do MIRRES_LIB=$PWD/ab/libmirres_occlcount.so MIRRES_MESH=$mesh timeout 300 python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench as B
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
S = M.scene; mesh = os.environ["MIRRES_MESH"]
v, t = S.mesh_by_name(mesh)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mlp = B.make_field(S, torch, torch.device("cuda", 0))
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
ctx = get_ctx(g["fx"], g["fy"])
ctx.stats(reset=True)
RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], 32, 2, 2, 2.0, 0.1, 0.001, 12345)
torch.cuda.synchronize()
st = ctx.stats(reset=True)
print(mesh, "shadow rays of the frame", st["rays_any"], "answered by the hint", st["any_dead"], "= %.3f of all shadow rays" % (st["any_dead"] / max(1, st["rays_any"])))
PY
  done
} 2>&1 | tee $O
