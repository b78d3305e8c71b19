"""BASELINE config 3: one stage-1 inverse-rendering step (forward + backward through FinalShading / EvaluateFinalSamples_di / EAW / material field)
at 800x800, spp 32 (reference training default, main.py:108), synthetic scene. Reports ms/step and peak memory.
    python scripts/train_step_bench.py [--res 800 --ssaa 1 --spp 32 --steps 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_step_bench.py ...
Under torch.distributed.run the step is data parallel (BASELINE configs[4]: "grads all-reduced over xGMI"): every rank renders its own view of the
same scene, the parameter gradients (hash grid 50 MB, MLP, environment map) are summed as ONE flat RCCL all-reduce (dist.allreduce_gradients) and
every rank takes the same optimiser step; the parameters are checked to be identical on all ranks at the end."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D

p = argparse.ArgumentParser(); p.add_argument("--res", type=int, default=800); p.add_argument("--ssaa", type=int, default=1)
p.add_argument("--spp", type=int, default=32); p.add_argument("--steps", type=int, default=3); p.add_argument("--fixed-seed", action="store_true"); a = p.parse_args()
import numpy as _np; _np.random.seed(0)
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
if world > 1:
    import torch.distributed as dist
    from mirres_restir_nerf_mesh_amd import dist as MD
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("MIRRES_DIST_BACKEND", "nccl")
    dist.init_process_group(backend, **({"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}))
torch.manual_seed(0)
S = M.scene
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
params, w0, w1, w2 = S.make_matnet_params(seed=0); mn, mx = S.material_min_max()
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()))
with torch.no_grad():
    mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
    for i, w in zip((0, 2, 4), (w0, w1, w2)): mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
g = harness.build_gbuffer(W, a.res, a.res, a.ssaa, azimuth_deg=30.0 + 360.0 * rank / world)       # one view per rank
fx, fy = g["fx"], g["fy"]; N = fx * fy
mods = RR.load_m_for_restir(fx, fy)
env = torch.full((256, 512, 3), 0.5, device="cuda", requires_grad=True)          # create_trainable_env_rnd(scale=0, bias=0.5), network.py:126
opt = torch.optim.Adam([{"params": mlp.parameters(), "lr": 1e-3}, {"params": [env], "lr": 1e-2}])
target = torch.rand((N, 3), device="cuda") * 0.5 + 0.25
fg = g["occ"][:, 0] > 0.5
z = lambda *s: torch.zeros(s, device="cuda")
def step():
    if a.fixed_seed: RR.set_random_offset(4242)
    opt.zero_grad(set_to_none=True)
    W.update_mesh(W.vrt, W.v_ind)
    kdks = mlp.sample(g["pos"])
    kd = kdks[:, 0:3].contiguous(); rm = torch.cat((kdks[:, 4:5], kdks[:, 5:6]), -1).contiguous()
    out = RR.run_restir_di_with_pt(False, 1.0, 1.0, 1.0, mlp, None, W, *mods[:8], *mods[8:17], env, g["occ"].clone(), g["normal"], g["depth"], kd, rm, g["ray_dir"], g["pos"],
                                   z(N, 1), z(N, 4), z(N, 3), z(N, 3), fx, fy, a.spp, 2, 2, 2.0, 0.1, 0.001)
    loss = (torch.clamp(out[0][fg], 0, 1) - target[fg]).abs().mean()
    loss.backward()
    if world > 1:
        MD.allreduce_gradients(list(mlp.parameters()) + [env])
    opt.step()
    with torch.no_grad(): env.clamp_(min=0.01)
    return float(loss.detach())
step(); torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter(); ls = [step() for _ in range(a.steps)]; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
if world > 1:
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    chk = torch.stack([p.detach().double().sum() for p in list(mlp.parameters()) + [env]])
    lo, hi = chk.clone(), chk.clone(); dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "parameters diverged across ranks"
if rank == 0 and world > 1:
    print(f"data parallel over {world} ranks (one view each, one flat all-reduce of {sum(p.numel() for p in list(mlp.parameters()) + [env]) * 4 / 2**20:.1f} MB of gradients per step); parameters identical on all ranks")
if rank == 0:
  print(f"stage-1 step {fx}x{fy} spp {a.spp}: {dt*1e3:.1f} ms/step  ({N*a.spp/dt/1e6:.1f} Msamples/s fwd+bwd)  peak torch memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB  losses {ls}")
if world > 1:
    dist.destroy_process_group()
