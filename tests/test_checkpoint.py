"""Reading the files a reference run leaves behind: stage-0 PLY meshes and stage-1 `.pth` checkpoints in the layout of
Trainer.save_checkpoint (nerf/utils.py:1840-1920).  The checkpoint files here are written by hand in that layout (keys as the reference model's
state dict names them); no reference run is available in this container to produce a real one."""
import os
import struct

import numpy as np
import pytest
import torch

from mirres_restir_nerf_mesh_amd import checkpoint as CK
from mirres_restir_nerf_mesh_amd import scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ply_round_trip_and_variants(tmp_path):
    v, t = scene.make_mesh(2, 4)
    for binary in (True, False):
        p = str(tmp_path / ("m%d.ply" % binary))
        CK.write_ply(p, v, t, binary=binary)
        v2, t2 = CK.read_ply(p)
        assert v2.dtype == np.float32 and t2.dtype == np.int32
        assert np.array_equal(v2, v.astype(np.float32)) and np.array_equal(t2, t)
    # big-endian file with extra vertex properties (normals, colour bytes), double coordinates, a quad and an unrelated element in between
    vv = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 0.5, 1]], np.float64)
    head = ("ply\nformat binary_big_endian 1.0\ncomment made by hand\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\n"
            "property float nx\nproperty uchar red\nelement misc 2\nproperty short a\nelement face 2\nproperty list uchar uint vertex_indices\nend_header\n")
    body = b"".join(struct.pack(">dddfB", *r, 0.25, 7) for r in vv) + struct.pack(">hh", 1, 2) + struct.pack(">B4I", 4, 0, 1, 2, 3) + struct.pack(">B3I", 3, 0, 1, 4)
    p = str(tmp_path / "be.ply")
    open(p, "wb").write(head.encode() + body)
    v3, t3 = CK.read_ply(p)
    assert np.array_equal(v3, vv.astype(np.float32)) and t3.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]     # the quad is fan-split
    open(str(tmp_path / "bad.ply"), "wb").write(b"solid not a ply\n")
    with pytest.raises(ValueError):
        CK.read_ply(str(tmp_path / "bad.ply"))
    bad = head.replace("element face 2", "element face 1").encode() + b"".join(struct.pack(">dddfB", *r, 0.25, 7) for r in vv) + struct.pack(">hh", 1, 2) + struct.pack(">B3I", 3, 0, 1, 9)
    open(str(tmp_path / "oob.ply"), "wb").write(bad)
    with pytest.raises(ValueError, match="out of range"):
        CK.read_ply(str(tmp_path / "oob.ply"))


def test_stage0_mesh_cascades(tmp_path):
    """nerf/renderer.py:146-171: the updated mesh wins when present, cascades are concatenated with shifted indices, --ckpt scratch ignores updates."""
    d = tmp_path / "mesh_stage0"; d.mkdir()
    v0, t0 = scene.make_mesh(1, 2); v1, t1 = scene.make_mesh(2, 2)
    CK.write_ply(str(d / "mesh_0.ply"), v0, t0)
    CK.write_ply(str(d / "mesh_0_updated.ply"), v0 * 1.5, t0)
    CK.write_ply(str(d / "mesh_1.ply"), v1, t1)
    v, t, vc, fc = CK.load_stage0_mesh(str(tmp_path), cascade=2)
    assert vc.tolist() == [0, v0.shape[0], v0.shape[0] + v1.shape[0]] and fc.tolist() == [0, t0.shape[0], t0.shape[0] + t1.shape[0]]
    assert np.array_equal(v[:vc[1]], (v0 * 1.5).astype(np.float32)) and np.array_equal(v[vc[1]:], v1.astype(np.float32))
    assert np.array_equal(t[:fc[1]], t0) and np.array_equal(t[fc[1]:], t1 + v0.shape[0])
    v_s, _, _, _ = CK.load_stage0_mesh(str(tmp_path), cascade=1, from_scratch=True)
    assert np.array_equal(v_s, v0.astype(np.float32))


def _reference_layout(n_vert=12, n_grid=64, half_light=False):
    g = torch.Generator().manual_seed(3)
    model = {"vertices_offsets": torch.rand(n_vert, 3, generator=g) * 0.01, "mlp_mat_opt.encoder.params": torch.rand(n_grid, generator=g),
             "mlp_mat_opt.net.net.0.weight": torch.rand(32, 32, generator=g), "mlp_mat_opt.net.net.2.weight": torch.rand(32, 32, generator=g),
             "mlp_mat_opt.net.net.4.weight": torch.rand(6, 32, generator=g),
             # stage-0 entries of the same state dict that this path ignores
             "aabb_train": torch.tensor([-1., -1, -1, 1, 1, 1]), "density_bitfield": torch.zeros(16, dtype=torch.uint8), "color_net.0.weight": torch.rand(8, 8, generator=g)}
    light = torch.rand(8, 16, 3, generator=g) + 0.01
    return {"epoch": 7, "global_step": 1234, "stats": {"results": []}, "stage": 1, "light_base": light.half() if half_light else light, "model": model}


def test_read_checkpoint_in_the_reference_layout(tmp_path):
    ck = _reference_layout(half_light=True)
    p = str(tmp_path / "ngp_stage1_ep0007.pth")
    torch.save(ck, p)
    r = CK.read_checkpoint(p)
    assert r["epoch"] == 7 and r["global_step"] == 1234 and r["stage"] == 1
    assert torch.equal(r["vertices_offsets"], ck["model"]["vertices_offsets"]) and torch.equal(r["grid_params"], ck["model"]["mlp_mat_opt.encoder.params"])
    assert all(torch.equal(a, ck["model"]["mlp_mat_opt.net.net.%d.weight" % i]) for a, i in zip(r["mlp_weights"], (0, 2, 4)))
    assert r["light_base"].dtype == torch.float32 and torch.equal(r["light_base"], ck["light_base"].float())
    voff, light = CK.apply_checkpoint(r, None, n_vertices=12, device="cpu")
    assert torch.equal(voff, r["vertices_offsets"]) and torch.equal(light, r["light_base"])
    with pytest.raises(ValueError, match="12 vertices"):
        CK.apply_checkpoint(r, None, n_vertices=13, device="cpu")
    # a bare state dict (load_checkpoint's first branch, :1942-1946): no light, no counters
    p2 = str(tmp_path / "bare.pth"); torch.save(ck["model"], p2)
    r2 = CK.read_checkpoint(p2)
    assert r2["light_base"] is None and r2["epoch"] is None and torch.equal(r2["grid_params"], r["grid_params"])
    # a material field with a layer missing is refused, a file without any material field reads as None
    broken = dict(ck["model"]); del broken["mlp_mat_opt.net.net.2.weight"]
    p3 = str(tmp_path / "broken.pth"); torch.save({"model": broken}, p3)
    with pytest.raises(KeyError, match="net.2"):
        CK.read_checkpoint(p3)
    p4 = str(tmp_path / "stage0.pth"); torch.save({"model": {"aabb_train": torch.zeros(6)}, "epoch": 1}, p4)
    r4 = CK.read_checkpoint(p4)
    assert r4["grid_params"] is None and r4["mlp_weights"] is None and r4["vertices_offsets"] is None
    with pytest.raises(KeyError):
        CK.apply_checkpoint(r4, mlp_mat=object())


def test_material_field_constants_follow_the_reference_cli():
    """nerf/network.py:119-125 + main.py:39,109-110,167-170: AABB = +-bound, min = (kd_min, 0, roughness_min, 0), max = (kd_max, 0, 1, me_max);
    cascades = 1 + ceil(log2(bound)) (nerf/renderer.py:97).  A reference checkpoint records none of them: evaluation warns and uses main.py's defaults;
    command-line values win over recorded ones and a mismatch is reported."""
    aabb, mn, mx = CK.material_field_args(CK.material_config())
    assert aabb.tolist() == [-2, -2, -2, 2, 2, 2] and mn.tolist() == pytest.approx([0, 0, 0, 0, 0.08, 0]) and mx.tolist() == [1, 1, 1, 0, 1, 0]
    aabb, mn, mx = CK.material_field_args(CK.material_config(bound=1, me_max=0.5, roughness_min=0.2))     # configs/OWL/gamepad.txt: --me_max 0.5
    assert aabb.tolist() == [-1, -1, -1, 1, 1, 1] and mn[4].item() == pytest.approx(0.2) and mx.tolist() == [1, 1, 1, 0, 1, 0.5]
    assert [CK.cascade_of_bound(b) for b in (0.5, 1, 1.5, 2, 3, 4, 16)] == [1, 1, 2, 2, 3, 3, 5]
    with pytest.raises(KeyError):
        CK.material_config(bond=1)
    msgs = []
    c = CK.resolve_material_config(None, warn=msgs.append, me_max=0.5)
    assert c["bound"] == 2.0 and c["me_max"] == 0.5 and len(msgs) == 1 and "--bound 2.0" in msgs[0] and "me_max" not in msgs[0]
    msgs.clear()
    c = CK.resolve_material_config(dict(bound=1.0, me_max=0.5), warn=msgs.append)
    assert c["bound"] == 1.0 and c["me_max"] == 0.5 and c["roughness_min"] == 0.08 and not msgs
    c = CK.resolve_material_config(dict(bound=1.0, me_max=0.5), warn=msgs.append, bound=2.0, me_max=0.5)
    assert c["bound"] == 2.0 and len(msgs) == 1 and "--bound" in msgs[0]


def test_read_checkpoint_returns_recorded_material_constants(tmp_path):
    ck = _reference_layout()
    p = str(tmp_path / "a.pth"); torch.save(ck, p)
    assert CK.read_checkpoint(p)["material_config"] is None                    # the reference's own files carry none
    ck["material_config"] = CK.material_config(bound=1.0, me_max=0.5)
    p = str(tmp_path / "b.pth"); torch.save(ck, p)
    assert CK.read_checkpoint(p)["material_config"]["me_max"] == 0.5


def test_reads_a_checkpoint_written_by_the_reference_code():
    """tests/golden/ref_checkpoint_stage1.pth was written by the reference's OWN Trainer.save_checkpoint (nerf/utils.py:1838-1883, executed from its AST
    over a model built from the reference's MLPTexture3D / _MLP classes; gen_reference_checkpoint.py): read_checkpoint must find every tensor under the
    names that code used, and the generator also checked the other direction (a file written by checkpoint.save_checkpoint loads through the
    reference's own load_checkpoint)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = np.load(os.path.join(here, "ref_checkpoint_stage1.npz"))
    r = CK.read_checkpoint(os.path.join(here, "ref_checkpoint_stage1.pth"))
    assert r["epoch"] == int(want["epoch"]) and r["global_step"] == int(want["global_step"]) and r["stage"] == 1 and r["material_config"] is None
    assert np.array_equal(r["vertices_offsets"].numpy(), want["voff"]) and np.array_equal(r["grid_params"].numpy(), want["grid"])
    assert all(np.array_equal(a.numpy(), want[k]) for a, k in zip(r["mlp_weights"], ("w0", "w1", "w2")))
    assert np.array_equal(r["light_base"].numpy(), want["light"]) and r["light_base"].dtype == torch.float32
    assert set(want["keys"].tolist()) >= {"vertices_offsets", "mlp_mat_opt.encoder.params", "mlp_mat_opt.net.net.0.weight", "mlp_mat_opt.net.net.2.weight", "mlp_mat_opt.net.net.4.weight"}
    assert bool(want["package_file_loads_in_reference"])
    voff, light = CK.apply_checkpoint(r, None, n_vertices=12, device="cpu")
    assert torch.equal(voff, r["vertices_offsets"]) and torch.equal(light, r["light_base"])


def test_train_state_round_trip_and_schedule_fast_forward(tmp_path):
    """A resumable checkpoint (Trainer.save_checkpoint full=True, nerf/utils.py:1856-1866) carries the three optimisers and their schedules under the reference's
    keys; read_checkpoint hands them back, and an optimiser restored from them takes the same next step as the one that was saved.  A checkpoint without them
    (the reference's default) still lets a resumed run continue the learning-rate SCHEDULE: it is a function of the step count (scripts/train_stage1.py)."""
    import types
    from mirres_restir_nerf_mesh_amd import checkpoint as CK
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(8)); env = torch.nn.Parameter(torch.rand(4, 8, 3))
    sched = lambda it: max(0.0, 10 ** (-it * 0.0002))
    def make():
        o = torch.optim.Adam([{"params": [w], "lr": 0.03}]); return o, torch.optim.lr_scheduler.LambdaLR(o, sched)
    o, s = make()
    for it in range(40):
        o.zero_grad(); (w * w).sum().backward(); o.step(); s.step()
    mlp = types.SimpleNamespace(encoder=types.SimpleNamespace(params=torch.zeros(16)), net=types.SimpleNamespace(net={i: types.SimpleNamespace(weight=torch.zeros(2, 2)) for i in (0, 2, 4)}))
    cfg = CK.material_config(bound=1.0, roughness_min=0.08, me_max=0.0)
    p = str(tmp_path / "resume.pth")
    CK.save_checkpoint(p, mlp, torch.zeros(5, 3), env, epoch=40, global_step=40, material_config=cfg, train_state={"optimizer_mat": o.state_dict(), "scheduler_mat": s.state_dict()})
    with pytest.raises(KeyError):
        CK.save_checkpoint(p + "x", mlp, torch.zeros(5, 3), env, material_config=cfg, train_state={"optimiser": {}})
    ck = CK.read_checkpoint(p)
    assert set(ck["train_state"]) == {"optimizer_mat", "scheduler_mat"} and ck["global_step"] == 40
    w_saved = w.detach().clone()
    o.zero_grad(); (w * w).sum().backward(); o.step(); s.step(); after = w.detach().clone(); lr_after = o.param_groups[0]["lr"]
    with torch.no_grad():
        w.copy_(w_saved)
    o2, s2 = make()
    o2.load_state_dict(ck["train_state"]["optimizer_mat"]); s2.load_state_dict(ck["train_state"]["scheduler_mat"])
    o2.zero_grad(); (w * w).sum().backward(); o2.step(); s2.step()
    assert torch.equal(w.detach(), after) and o2.param_groups[0]["lr"] == lr_after
    # no saved state: the schedule alone is fast-forwarded to the checkpoint's step
    o3, s3 = make()
    s3.last_epoch = 40
    for g_, base, fn in zip(s3.optimizer.param_groups, s3.base_lrs, s3.lr_lambdas):
        g_["lr"] = base * fn(40)
    assert o3.param_groups[0]["lr"] == pytest.approx(0.03 * sched(40)) and CK.read_checkpoint(p)["train_state"] is not None


def test_reference_full_checkpoint_train_state():
    """tests/golden/ref_checkpoint_stage1_full.pth was written by the reference's own Trainer.save_checkpoint(full=True) (nerf/utils.py:1856-1867) over real
    optimisers / LambdaLR schedules after nine steps.  read_checkpoint must hand back all six state dicts under the reference's key names — in particular
    `scheduler_mat` / `scheduler_light`, which load_checkpoint (:2003-2022) reads — and save_checkpoint must accept those names; the generator also verified
    that a package-written full file resumes in the reference's load_checkpoint (recorded in the npz).  Names this package wrote before round 4 are read too."""
    import types
    from mirres_restir_nerf_mesh_amd import checkpoint as CK
    g = np.load(os.path.join(GOLD, "ref_checkpoint_stage1.npz"))
    assert bool(g["package_full_file_resumes_in_reference"])
    assert {"optimizer", "lr_scheduler", "optimizer_mat", "scheduler_mat", "optimizer_light", "scheduler_light"} <= set(g["full_top_keys"].tolist())
    ck = CK.read_checkpoint(os.path.join(GOLD, "ref_checkpoint_stage1_full.pth"))
    ts = ck["train_state"]
    assert set(ts) == set(CK.TRAIN_STATE_KEYS)
    assert ts["scheduler_mat"]["last_epoch"] == int(g["full_sched_mat_last_epoch"]) == 9 and ts["scheduler_light"]["last_epoch"] == int(g["full_sched_light_last_epoch"])
    # a LambdaLR restored from it continues the reference's learning rate
    w = torch.nn.Parameter(torch.zeros(3)); o = torch.optim.Adam([{"params": [w], "lr": 1e-2}])
    s = torch.optim.lr_scheduler.LambdaLR(o, lambda it: max(0.0, 10 ** (-it * 0.0002)))
    s.load_state_dict(ts["scheduler_mat"])
    assert s.last_epoch == 9 and s.get_last_lr()[0] == pytest.approx(float(g["full_lr_mat"]), rel=1e-12)
    # legacy names (rounds 1-3 of this package) are mapped on read
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "old.pth")
        torch.save({"model": {"vertices_offsets": torch.zeros(4, 3)}, "epoch": 1, "global_step": 5, "stage": 1, "lr_scheduler_mat": {"last_epoch": 5}, "lr_scheduler_light": {"last_epoch": 5}}, p)
        old = CK.read_checkpoint(p)["train_state"]
        assert set(old) == {"scheduler_mat", "scheduler_light"} and old["scheduler_mat"]["last_epoch"] == 5
