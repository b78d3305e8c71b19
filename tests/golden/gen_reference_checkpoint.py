"""Fixture generator (runs in the build container only): a stage-1 checkpoint written by the reference's OWN `Trainer.save_checkpoint`
(nerf/utils.py:1838-1883, taken from the file's AST) over a model whose material field is the reference's own `MLPTexture3D` / `_MLP` classes
(nerf/render_helper.py:28-124) — tests/golden/ref_checkpoint_stage1.pth — plus the tensors that went in (ref_checkpoint_stage1.npz).
mirres-restir_nerf_mesh_amd/checkpoint.py must read that file; conversely a file written by checkpoint.save_checkpoint is loaded here by the
reference's own `Trainer.load_checkpoint` (:1926-1990) and the values that arrive in the model are checked (the result is recorded in the npz).

Stand-ins (what the image lacks): tiny-cuda-nn — `tcnn.Encoding` is replaced by an nn.Module that, like tcnn's torch binding, keeps its table in one
flat parameter named `params` (64 entries here: the fixture pins names and nesting, not sizes); the Trainer and the NeRF network are plain namespaces
carrying only the attributes the two methods touch.  Nothing of the reference's text is stored: only the file its code wrote."""
import ast
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def ref_classes(path, names, ns):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert len(cls) == len(names)
    exec(compile(ast.Module(body=cls, type_ignores=[]), os.path.join(REF, path), "exec"), ns)
    return [ns[n] for n in names]


def ref_methods(path, cls, names, ns):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    fns = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    exec(compile(ast.Module(body=fns, type_ignores=[]), os.path.join(REF, path), "exec"), ns)
    return [ns[n] for n in names]


class Encoding(torch.nn.Module):          # tcnn.Encoding(n_input_dims, config): one flat parameter `params`
    n_output_dims = 32

    def __init__(self, n_in, cfg):
        super().__init__()
        self.params = torch.nn.Parameter(torch.zeros(64))


def build_model(seed):
    g = torch.Generator().manual_seed(seed)
    ns = {"torch": torch, "np": np, "tcnn": types.SimpleNamespace(Encoding=Encoding, free_temporary_memory=lambda: None)}
    _, MLPTexture3D = ref_classes("nerf/render_helper.py", ["_MLP", "MLPTexture3D"], ns)
    tc, mc = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self; torch.nn.Module.cuda = lambda self, *a, **k: self      # the reference hard-codes .cuda(); no GPU here
    try:
        mat = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6,
                           min_max=[torch.tensor([0, 0, 0, 0, 0.08, 0.0]), torch.tensor([1, 1, 1, 0, 1, 0.0])])
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = tc, mc
    model = torch.nn.Module()
    model.cuda_ray = False
    model.vertices_offsets = torch.nn.Parameter(torch.rand(12, 3, generator=g) * 0.01)
    model.mlp_mat_opt = mat
    model.register_buffer("aabb_train", torch.tensor([-1., -1, -1, 1, 1, 1]))                               # a stage-0 entry this path ignores
    model.lgt = types.SimpleNamespace(base=torch.nn.Parameter(torch.rand(8, 16, 3, generator=g) + 0.01))     # EnvironmentLight.base (not an nn.Module member)
    with torch.no_grad():
        mat.encoder.params.copy_(torch.rand(64, generator=g) * 2e-4 - 1e-4)
    return model


def trainer_for(model, ckpt_path):
    return types.SimpleNamespace(name="ngp_stage1", epoch=7, global_step=1234, stats={"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},
                                 opt=types.SimpleNamespace(stage=1, use_brdf=True), model=model, ckpt_path=ckpt_path, max_keep_ckpt=2, ema=None, device="cpu",
                                 log=lambda *a, **k: None, optimizer=None, lr_scheduler=None, scaler=None)


def main():
    ns = {"torch": torch, "np": np, "os": os, "glob": __import__("glob")}
    save_checkpoint, load_checkpoint = ref_methods("nerf/utils.py", "Trainer", ["save_checkpoint", "load_checkpoint"], ns)
    tmp = tempfile.mkdtemp()
    try:
        model = build_model(3)
        tr = trainer_for(model, tmp)
        save_checkpoint(tr)                                                           # the reference's code writes <name>_ep0007.pth
        src = os.path.join(tmp, "ngp_stage1_ep0007.pth")
        assert os.path.exists(src) and tr.stats["checkpoints"] == ["ngp_stage1_ep0007.pth"]
        shutil.copy(src, os.path.join(HERE, "ref_checkpoint_stage1.pth"))
        sd = model.state_dict()
        out = {"keys": np.array(sorted(sd.keys())), "voff": sd["vertices_offsets"].numpy(), "grid": sd["mlp_mat_opt.encoder.params"].numpy(),
               "w0": sd["mlp_mat_opt.net.net.0.weight"].numpy(), "w1": sd["mlp_mat_opt.net.net.2.weight"].numpy(), "w2": sd["mlp_mat_opt.net.net.4.weight"].numpy(),
               "light": model.lgt.base.detach().numpy(), "epoch": np.int64(7), "global_step": np.int64(1234)}
        # the other direction: a file written by this package's save_checkpoint, loaded by the reference's load_checkpoint into a fresh model
        from mirres_restir_nerf_mesh_amd import checkpoint as CK
        mine = build_model(11)
        holder = types.SimpleNamespace(encoder=mine.mlp_mat_opt.encoder, net=mine.mlp_mat_opt.net, AABB=mine.mlp_mat_opt.AABB, min_max=mine.mlp_mat_opt.min_max)
        p2 = os.path.join(tmp, "from_package.pth")
        CK.save_checkpoint(p2, holder, mine.vertices_offsets, mine.lgt.base, epoch=3, global_step=77)
        fresh = build_model(5); tr2 = trainer_for(fresh, tmp)
        tr2.optimizer = None
        load_checkpoint(tr2, p2, model_only=True)
        same = all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), mine.state_dict().values())) and torch.equal(fresh.lgt.base, mine.lgt.base.detach())
        assert same, "a file written by checkpoint.save_checkpoint does not load into the reference's model"
        out["package_file_loads_in_reference"] = np.bool_(same)
        # ---- a resumable file: the reference's save_checkpoint(full=True) (:1856-1867) over real optimisers / LambdaLR schedules that took 9 steps
        def train_objs(m, steps):
            sched = lambda it: max(0.0, 10 ** (-it * 0.0002))
            o_g = torch.optim.Adam([{"params": [m.vertices_offsets], "lr": 1e-3}]); s_g = torch.optim.lr_scheduler.LambdaLR(o_g, lambda it: 0.5 ** (it / 10))
            o_m = torch.optim.Adam([{"params": list(m.mlp_mat_opt.parameters()), "lr": 1e-2}]); s_m = torch.optim.lr_scheduler.LambdaLR(o_m, sched)
            o_l = torch.optim.Adam([{"params": [m.lgt.base], "lr": 3e-2}]); s_l = torch.optim.lr_scheduler.LambdaLR(o_l, sched)
            for _ in range(steps):
                for o_ in (o_g, o_m, o_l):
                    o_.zero_grad()
                (m.vertices_offsets.square().sum() + sum(p_.square().sum() for p_ in m.mlp_mat_opt.parameters()) + m.lgt.base.square().sum()).backward()
                for o_, s_ in ((o_g, s_g), (o_m, s_m), (o_l, s_l)):
                    o_.step(); s_.step()
            return o_g, s_g, o_m, s_m, o_l, s_l
        full_model = build_model(21)
        trf = trainer_for(full_model, tmp)
        trf.optimizer, trf.lr_scheduler, trf.optimizer_mat, trf.scheduler_mat, trf.optimizer_light, trf.scheduler_light = train_objs(full_model, 9)
        trf.scaler = torch.amp.GradScaler("cpu", enabled=False)
        trf.epoch = 9; trf.global_step = 9
        save_checkpoint(trf, full=True)
        srcf = os.path.join(tmp, "ngp_stage1_ep0009.pth")
        shutil.copy(srcf, os.path.join(HERE, "ref_checkpoint_stage1_full.pth"))
        top = torch.load(srcf, map_location="cpu", weights_only=False)
        out["full_top_keys"] = np.array(sorted(top.keys()))
        out["full_sched_mat_last_epoch"] = np.int64(top["scheduler_mat"]["last_epoch"]); out["full_sched_light_last_epoch"] = np.int64(top["scheduler_light"]["last_epoch"])
        out["full_lr_mat"] = np.float64(trf.optimizer_mat.param_groups[0]["lr"])
        # and the other direction with train state: this package writes, the reference's load_checkpoint (model_only=False) restores optimisers AND schedules
        ck_f = CK.read_checkpoint(srcf)
        p3 = os.path.join(tmp, "from_package_full.pth")
        holder_f = types.SimpleNamespace(encoder=full_model.mlp_mat_opt.encoder, net=full_model.mlp_mat_opt.net, AABB=full_model.mlp_mat_opt.AABB, min_max=full_model.mlp_mat_opt.min_max)
        CK.save_checkpoint(p3, holder_f, full_model.vertices_offsets, full_model.lgt.base, epoch=9, global_step=9, train_state=ck_f["train_state"])
        fresh_f = build_model(6); tr3 = trainer_for(fresh_f, tmp)
        tr3.optimizer, tr3.lr_scheduler, tr3.optimizer_mat, tr3.scheduler_mat, tr3.optimizer_light, tr3.scheduler_light = train_objs(fresh_f, 0)
        load_checkpoint(tr3, p3)
        ok_full = (tr3.scheduler_mat.last_epoch == 9 and tr3.scheduler_light.last_epoch == 9 and tr3.lr_scheduler.last_epoch == 9
                   and tr3.optimizer_mat.state_dict()["state"][0]["step"] == trf.optimizer_mat.state_dict()["state"][0]["step"] and tr3.global_step == 9)
        assert ok_full, "the reference's load_checkpoint did not restore the schedules / optimisers from a package-written full checkpoint"
        out["package_full_file_resumes_in_reference"] = np.bool_(ok_full)
        np.savez_compressed(os.path.join(HERE, "ref_checkpoint_stage1.npz"), **out)
        print("wrote ref_checkpoint_stage1.pth (%d bytes); keys: %s; package-written file loads in the reference: %s" % (os.path.getsize(src), list(sd.keys()), same))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
