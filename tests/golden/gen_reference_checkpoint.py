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
        np.savez_compressed(os.path.join(HERE, "ref_checkpoint_stage1.npz"), **out)
        print("wrote ref_checkpoint_stage1.pth (%d bytes); keys: %s; package-written file loads in the reference: %s" % (os.path.getsize(src), list(sd.keys()), same))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
