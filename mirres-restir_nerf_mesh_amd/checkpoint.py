"""Loading what a reference run leaves on disk, so that `--test --spp N` evaluation can start from the reference's own files:

  * the stage-0 meshes `<workspace>/mesh_stage0/mesh_{cas}[_updated].ply` (nerf/renderer.py:150-171; the reference reads them with trimesh, which
    is not a dependency here: `read_ply` is a plain PLY reader for the vertex x/y/z + face-list files it writes);
  * a stage-1 checkpoint `*.pth` (Trainer.save_checkpoint / load_checkpoint, nerf/utils.py:1840-1964): `state['model']` = the model's state dict
    (`vertices_offsets`, `mlp_mat_opt.encoder.params`, `mlp_mat_opt.net.net.{0,2,4}.weight`, plus the stage-0 NeRF entries this path ignores) and
    the environment map under the top-level key `light_base` (EnvironmentLight is not an nn.Module, :1853-1854).

Nothing here touches the GPU except `apply_checkpoint`, which copies into an existing MLPTexture3D.  Shape or key mismatches raise — there is no
partial load of the material field.
"""
import os
import struct

import numpy as np
import torch

__all__ = ["read_ply", "write_ply", "load_stage0_mesh", "read_checkpoint", "apply_checkpoint", "save_checkpoint", "material_config",
           "resolve_material_config", "material_field_args", "cascade_of_bound"]

_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H", "int": "i", "int32": "i",
              "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply(path):
    """(vertices f32 [V, 3], faces i32 [F, 3]) of a PLY file: ascii or binary (either endianness), any extra vertex properties skipped, polygons
    with more than three corners fan-split."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError("%s: not a PLY file" % path)
    body = data.index(b"\n", end) + 1
    fmt = None; elements = []
    for line in data[:end].decode("ascii", "replace").splitlines():
        w = line.split()
        if not w:
            continue
        if w[0] == "format":
            fmt = w[1]
        elif w[0] == "element":
            elements.append({"name": w[1], "count": int(w[2]), "props": []})
        elif w[0] == "property":
            if w[1] == "list":
                elements[-1]["props"].append(("list", _PLY_TYPES[w[2]], _PLY_TYPES[w[3]], w[4]))
            else:
                elements[-1]["props"].append(("scalar", _PLY_TYPES[w[1]], None, w[2]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
    verts = None; faces = []
    if fmt == "ascii":
        tok = data[body:].split()
        pos = 0
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for kind, t0, t1, name in el["props"]:
                    if kind == "scalar":
                        row[name] = float(tok[pos]); pos += 1
                    else:
                        k = int(tok[pos]); pos += 1
                        row[name] = [int(float(x)) for x in tok[pos:pos + k]]; pos += k
                rows.append(row)
            verts, faces = _collect(el, rows, verts, faces)
    else:
        e = "<" if fmt == "binary_little_endian" else ">"
        pos = body
        for el in elements:
            if all(p[0] == "scalar" for p in el["props"]):      # fixed-size records: one frombuffer
                dt = np.dtype([(p[3], e + p[1]) for p in el["props"]])
                arr = np.frombuffer(data, dt, el["count"], pos); pos += dt.itemsize * el["count"]
                if el["name"] == "vertex":
                    verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float32)
                continue
            rows = []
            for _ in range(el["count"]):
                row = {}
                for kind, t0, t1, name in el["props"]:
                    if kind == "scalar":
                        row[name] = struct.unpack_from(e + t0, data, pos)[0]; pos += struct.calcsize(t0)
                    else:
                        k = struct.unpack_from(e + t0, data, pos)[0]; pos += struct.calcsize(t0)
                        row[name] = list(struct.unpack_from(e + str(k) + t1, data, pos)); pos += k * struct.calcsize(t1)
                rows.append(row)
            verts, faces = _collect(el, rows, verts, faces)
    if verts is None:
        raise ValueError("%s: no vertex element" % path)
    f = np.asarray(faces, np.int32).reshape(-1, 3)
    if f.size and (f.min() < 0 or f.max() >= verts.shape[0]):
        raise ValueError("%s: face index out of range" % path)
    return verts, f


def _collect(el, rows, verts, faces):
    if el["name"] == "vertex":
        verts = np.array([[r["x"], r["y"], r["z"]] for r in rows], np.float32).reshape(-1, 3)
    elif el["name"] == "face":
        key = next(p[3] for p in el["props"] if p[0] == "list")
        for r in rows:
            idx = r[key]
            for k in range(1, len(idx) - 1):
                faces.append((idx[0], idx[k], idx[k + 1]))
    return verts, faces


def write_ply(path, vertices, faces, binary=True):
    """Vertex x/y/z (float) + face list (uchar count, int indices) — the layout trimesh exports and the reference's stage-0 writes."""
    v = np.asarray(vertices, np.float32).reshape(-1, 3); f = np.asarray(faces, np.int32).reshape(-1, 3)
    head = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (
        "binary_little_endian" if binary else "ascii", v.shape[0], f.shape[0])
    with open(path, "wb") as fh:
        fh.write(head.encode("ascii"))
        if binary:
            fh.write(v.astype("<f4").tobytes())
            rec = np.empty(f.shape[0], np.dtype([("n", "u1"), ("i", "<i4", 3)])); rec["n"] = 3; rec["i"] = f
            fh.write(rec.tobytes())
        else:
            for r in v:
                fh.write(("%r %r %r\n" % (float(r[0]), float(r[1]), float(r[2]))).encode("ascii"))
            for r in f:
                fh.write(("3 %d %d %d\n" % (r[0], r[1], r[2])).encode("ascii"))


def load_stage0_mesh(workspace, cascade=1, mesh="", from_scratch=False):
    """nerf/renderer.py:146-171: per cascade `mesh_{cas}_updated.ply` when it exists (and --ckpt is not 'scratch'), else `mesh_{cas}.ply`; `--mesh`
    overrides the updated path; cascades are concatenated with their face indices shifted.  Returns (vertices f32, triangles i32, v_cumsum, f_cumsum)."""
    vs, ts = [], []; v_cumsum = [0]; f_cumsum = [0]
    for cas in range(cascade):
        updated = os.path.join(workspace, "mesh_stage0", "mesh_%d_updated.ply" % cas) if mesh == "" else mesh
        path = updated if (os.path.exists(updated) and not from_scratch) else os.path.join(workspace, "mesh_stage0", "mesh_%d.ply" % cas)
        v, f = read_ply(path)
        vs.append(v); ts.append(f + v_cumsum[-1])
        v_cumsum.append(v_cumsum[-1] + v.shape[0]); f_cumsum.append(f_cumsum[-1] + f.shape[0])
    return np.concatenate(vs, 0).astype(np.float32), np.concatenate(ts, 0).astype(np.int32), np.array(v_cumsum), np.array(f_cumsum)


_MAT = "mlp_mat_opt."

# The reference's CLI defaults for what fixes the decoding of a material field (main.py:39,109-110,167-170).  They are NOT stored in a reference
# checkpoint: an evaluation must be given the training run's values.
_MATERIAL_DEFAULTS = dict(bound=2.0, roughness_min=0.08, me_max=0.0, kd_min=(0.0, 0.0, 0.0), kd_max=(1.0, 1.0, 1.0))


def material_config(**kw):
    """The constants that decode a material field, main.py's defaults overridden by `kw` (bound, roughness_min, me_max, kd_min, kd_max)."""
    bad = set(kw) - set(_MATERIAL_DEFAULTS)
    if bad:
        raise KeyError("unknown material-field constant(s): %s" % ", ".join(sorted(bad)))
    c = dict(_MATERIAL_DEFAULTS); c.update({k: v for k, v in kw.items() if v is not None})
    c["bound"] = float(c["bound"]); c["roughness_min"] = float(c["roughness_min"]); c["me_max"] = float(c["me_max"])
    c["kd_min"] = tuple(float(x) for x in c["kd_min"]); c["kd_max"] = tuple(float(x) for x in c["kd_max"])
    if not c["bound"] > 0:
        raise ValueError("bound must be positive")
    return c


_material_config = material_config      # save_checkpoint has a parameter of the same name


def resolve_material_config(recorded=None, warn=None, **cli):
    """Command-line values win; otherwise what the checkpoint recorded; otherwise main.py's defaults — with a warning, because a reference checkpoint
    records none of them and decoding it with another run's `--bound` / `--me_max` / `--roughness_min` gives wrong materials without any error."""
    import warnings
    warn = warn or (lambda m: warnings.warn(m, stacklevel=3))
    given = {k: v for k, v in cli.items() if v is not None}
    if recorded is None:
        missing = [k for k in ("bound", "roughness_min", "me_max") if k not in given]
        if missing:
            warn("checkpoint records no material-field constants; using main.py's defaults for %s — pass the training run's values if they differ"
                 % ", ".join("--%s %s" % (k, _MATERIAL_DEFAULTS[k]) for k in missing))
        return material_config(**given)
    rec = material_config(**recorded)
    out = material_config(**{**rec, **given})
    for k in given:
        if out[k] != rec[k]:
            warn("--%s %s differs from the value the checkpoint was written with (%s)" % (k, out[k], rec[k]))
    return out


def cascade_of_bound(bound):
    """nerf/renderer.py:97."""
    import math
    return 1 + math.ceil(math.log2(bound)) if bound > 1 else 1


def material_field_args(cfg):
    """(AABB [6], mlp_min [6], mlp_max [6]) exactly as nerf/network.py:119-125 builds them from the options: AABB = +-bound;
    min = (kd_min, 0, roughness_min, 0), max = (kd_max, 0, 1, me_max)  (main.py:167-170)."""
    b = cfg["bound"]
    aabb = torch.tensor([-b, -b, -b, b, b, b], dtype=torch.float32)
    mn = torch.tensor(list(cfg["kd_min"]) + [0.0, cfg["roughness_min"], 0.0], dtype=torch.float32)
    mx = torch.tensor(list(cfg["kd_max"]) + [0.0, 1.0, cfg["me_max"]], dtype=torch.float32)
    return aabb, mn, mx


def read_checkpoint(path, map_location="cpu"):
    """A reference `.pth` -> dict(vertices_offsets, grid_params, mlp_weights (3 tensors), light_base, epoch, global_step, stage).  Accepts both forms
    load_checkpoint accepts (:1942-1946): {'model': state_dict, ...} or a bare state dict.  Entries the file lacks come back as None; the material
    field is all-or-nothing."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    f32 = lambda t: None if t is None else t.detach().to(torch.float32)
    grid = model.get(_MAT + "encoder.params")
    ws = [model.get(_MAT + "net.net.%d.weight" % i) for i in (0, 2, 4)]
    have = [x is not None for x in [grid] + ws]
    if any(have) and not all(have):
        raise KeyError("%s: incomplete material field (%s)" % (path, ", ".join(k for k, h in zip(("encoder.params", "net.0", "net.2", "net.4"), have) if not h)))
    top = ck if isinstance(ck, dict) and "model" in ck else {}
    return dict(vertices_offsets=f32(model.get("vertices_offsets")), grid_params=f32(grid), mlp_weights=[f32(w) for w in ws] if all(have) else None,
                light_base=f32(top.get("light_base")), epoch=top.get("epoch"), global_step=top.get("global_step"), stage=top.get("stage"),
                material_config=top.get("material_config"), train_state=_train_state_of(top))


def apply_checkpoint(ck, mlp_mat=None, n_vertices=None, device="cuda"):
    """Copy a read_checkpoint() result into an MLPTexture3D and return (vertices_offsets, light_base) on `device` (None where the file had none).
    `light_base` replaces the model's map as load_checkpoint does (:1962-1964); the caller clamps / trains it as before."""
    if mlp_mat is not None:
        if ck["grid_params"] is None:
            raise KeyError("checkpoint has no material field")
        dst = [mlp_mat.encoder.params] + [mlp_mat.net.net[i].weight for i in (0, 2, 4)]
        src = [ck["grid_params"].reshape(-1)] + list(ck["mlp_weights"])
        for name, d, s in zip(("encoder.params", "net.0.weight", "net.2.weight", "net.4.weight"), dst, src):
            if tuple(d.shape) != tuple(s.shape):
                raise ValueError("material field %s: checkpoint shape %s, expected %s" % (name, tuple(s.shape), tuple(d.shape)))
        with torch.no_grad():
            for d, s in zip(dst, src):
                d.copy_(s.to(d.device))
    voff = ck["vertices_offsets"]
    if voff is not None and n_vertices is not None and voff.shape[0] != n_vertices:
        raise ValueError("vertices_offsets: checkpoint has %d vertices, the mesh %d" % (voff.shape[0], n_vertices))
    to = lambda t: None if t is None else t.to(device).contiguous()
    return to(voff), to(ck["light_base"])


# Trainer.save_checkpoint(full=True) writes exactly these names (nerf/utils.py:1856-1867) and load_checkpoint reads them back (:1966-2022);
# `scaler` / `ema` are not produced here (no AMP scaler, no EMA on this path) and are ignored on read.
def _train_state_of(top):
    ts = {k: top[k] for k in TRAIN_STATE_KEYS if k in top}
    for old, new in _LEGACY_TRAIN_STATE_KEYS.items():
        if old in top and new not in ts:
            ts[new] = top[old]
    return ts


TRAIN_STATE_KEYS = ("optimizer", "lr_scheduler", "optimizer_mat", "optimizer_light", "scheduler_mat", "scheduler_light")
# names this package wrote before round 4 (never understood by the reference); accepted on read only
_LEGACY_TRAIN_STATE_KEYS = {"lr_scheduler_mat": "scheduler_mat", "lr_scheduler_light": "scheduler_light"}


def save_checkpoint(path, mlp_mat, vertices_offsets, light_base, epoch=0, global_step=0, stage=1, material_config=None, train_state=None):
    """The same file layout, written from this engine's objects (Trainer.save_checkpoint, :1843-1854, 1912-1920), plus one extra
    top-level key the reference's loader ignores: `material_config` (the AABB / output-range constants the field was trained with, read back from the
    module when not given) so that an evaluation of this file cannot silently decode it with another run's `--bound` / `--me_max`.
    `train_state` (a `full=True` checkpoint, :1856-1866): state dicts under the reference's keys `optimizer`, `lr_scheduler`, `optimizer_mat`,
    `optimizer_light`, `scheduler_mat`, `scheduler_light` (LambdaLR state dicts, :1863-1867) so that a resumed run — here or through the reference's own
    `load_checkpoint` (:2003-2022) — continues the Adam moments and the learning-rate schedules where they were."""
    if material_config is None:
        lo, hi = (t.detach().cpu().tolist() for t in mlp_mat.AABB)
        mn, mx = (t.detach().cpu().tolist() for t in mlp_mat.min_max)
        if not (lo[0] == lo[1] == lo[2] == -hi[0] and hi[0] == hi[1] == hi[2]):
            raise ValueError("material field AABB %s..%s is not +-bound: pass material_config explicitly" % (lo, hi))
        material_config = _material_config(bound=hi[0], roughness_min=mn[4], me_max=mx[5], kd_min=mn[:3], kd_max=mx[:3])
    model = {"vertices_offsets": vertices_offsets.detach().cpu(), _MAT + "encoder.params": mlp_mat.encoder.params.detach().cpu()}
    for i in (0, 2, 4):
        model[_MAT + "net.net.%d.weight" % i] = mlp_mat.net.net[i].weight.detach().cpu()
    state = {"epoch": epoch, "global_step": global_step, "stats": {}, "stage": stage, "light_base": light_base.detach().cpu(), "model": model,
             "material_config": dict(material_config)}
    for k, v in (train_state or {}).items():
        if k not in TRAIN_STATE_KEYS:
            raise KeyError("train_state: unknown entry %r (expected %s)" % (k, ", ".join(TRAIN_STATE_KEYS)))
        state[k] = v
    torch.save(state, path)
