"""Mirror of the material-field part of nerf/render_helper.py (reference): MLPTexture3D = hash grid + tiny MLP.

State-dict compatible with the reference: `encoder.params` (flat fp32 [12 599 920]) and `net.net.{0,2,4}.weight`."""
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import lib, check, stream_ptr

def generate_envir_map_dir(envmap_h, envmap_w, is_jittor=False):
    """nerf/render_helper.py:8-26: the fixed lat-long light set of the non-ReSTIR renderer (render_dump.py). Returns (solid-angle weights [H*W] summing
    to 4 pi, unit directions [H*W,3]): z up, row 0 nearest the zenith, latitude / longitude sampled at the cell centres, longitude running from +pi down.
    Set-up code (once per model, network.py:131), plain torch as in the reference; `is_jittor` perturbs every direction inside its cell."""
    dlat, dlng = np.pi / envmap_h, 2 * np.pi / envmap_w
    lat = torch.linspace(np.pi / 2 - 0.5 * dlat, -np.pi / 2 + 0.5 * dlat, envmap_h)[:, None].expand(envmap_h, envmap_w)
    lng = torch.linspace(np.pi - 0.5 * dlng, -np.pi + 0.5 * dlng, envmap_w)[None, :].expand(envmap_h, envmap_w)
    colat_sin = torch.sin(torch.pi / 2 - lat)
    weights = (4 * torch.pi * colat_sin / colat_sin.sum()).to(torch.float32).reshape(-1)
    if is_jittor:
        lat = lat + dlat * (torch.rand_like(lat) - 0.5); lng = lng + dlng * (torch.rand_like(lng) - 0.5)
    cl = torch.cos(lat)
    dirs = torch.stack((torch.cos(lng) * cl, torch.sin(lng) * cl, torch.sin(lat)), dim=-1).reshape(-1, 3)
    return weights, dirs


GRADIENT_SCALING = 128.0  # render_helper.py:77-80: MLP input-grad hook x128 (=> grid grads x128), encoder input-grad hook /128


class _HashGrid(torch.nn.Module):
    """Stands in for tcnn.Encoding(3, HashGrid 16x2, T=2^19, base 16, scale 1.447): owns the flat fp32 master `params`."""
    n_output_dims = 32

    def __init__(self, seed=None):
        super().__init__()
        n = lib().mirres_matnet_grid_entries() * 2
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        p = (torch.rand(n, generator=g, dtype=torch.float32) * 2 - 1) * 1e-4  # tcnn grid init: U(-1e-4, 1e-4)
        self.params = torch.nn.Parameter(p.cuda())
        self._f16 = None
        self._f16_version = None

    def packed(self):
        """fp16 copy of the table the kernels read (tcnn computes in fp16); refreshed when `params` changes."""
        v = self.params._version
        if self._f16 is None or self._f16_version != v or self._f16.device != self.params.device:
            if self._f16 is None:
                self._f16 = torch.empty(self.params.numel(), dtype=torch.int16, device=self.params.device)
            check(lib().mirres_matnet_pack_grid(self.params.data_ptr(), self._f16.data_ptr(), self.params.numel(), stream_ptr()), "mirres_matnet_pack_grid")
            self._f16_version = v
        return self._f16


class _MLP(torch.nn.Module):
    """render_helper.py:28-51: bias-free Linear/ReLU stack with kaiming-uniform init; evaluated by the engine, not by torch."""

    def __init__(self, cfg, loss_scale=1.0):
        super().__init__()
        self.loss_scale = loss_scale
        net = (torch.nn.Linear(cfg['n_input_dims'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        for _ in range(cfg['n_hidden_layers'] - 1):
            net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_output_dims'], bias=False),)
        self.net = torch.nn.Sequential(*net).cuda()
        self.net.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if type(m) == torch.nn.Linear:
            torch.nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')


class _MatNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, pos, params, w0, w1, w2):
        ctx.pos_dtype = pos.dtype
        pos = pos.detach().contiguous().float()
        n = pos.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=pos.device)
        st = owner._struct()
        check(lib().mirres_matnet_fwd(C.byref(st), pos.data_ptr(), n, out.data_ptr(), None, stream_ptr()), "mirres_matnet_fwd")
        ctx.owner = owner
        ctx.save_for_backward(pos)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (pos,) = ctx.saved_tensors
        owner = ctx.owner
        grad_out = grad_out.contiguous().float()
        gp = torch.zeros_like(owner.encoder.params)
        w = [owner.net.net[i].weight for i in (0, 2, 4)]
        gw = [torch.zeros_like(t) for t in w]
        st = owner._struct()
        # sample() is differentiable in its argument too (tcnn's HashGrid returns input gradients; the reference sends kd / ks loss and smoothness
        # gradients to `vertices_offsets` this way, nerf/renderer.py:1017-1018): the x128 / /128 hooks cancel on this route (render_helper.py:41,78-80)
        g_pos = torch.empty_like(pos) if ctx.needs_input_grad[1] else None
        check(lib().mirres_matnet_bwd(C.byref(st), pos.data_ptr(), pos.shape[0], grad_out.data_ptr(), gp.data_ptr(), gw[0].data_ptr(), gw[1].data_ptr(),
                                      gw[2].data_ptr(), g_pos.data_ptr() if g_pos is not None and pos.shape[0] else None, stream_ptr()), "mirres_matnet_bwd")
        if g_pos is not None and g_pos.dtype != ctx.pos_dtype:
            g_pos = g_pos.to(ctx.pos_dtype)
        # the reference's hooks scale the gradient that reaches the encoder's parameters by 128
        return None, g_pos, gp * GRADIENT_SCALING, gw[0], gw[1], gw[2]


class MLPTexture3D(torch.nn.Module):
    """render_helper.py:53-124."""

    def __init__(self, AABB, channels=3, internal_dims=32, hidden=2, min_max=None, seed=None):
        super().__init__()
        if channels != 6 or internal_dims != 32 or hidden != 2:
            raise ValueError("the engine implements the reference's material field: channels=6, internal_dims=32, hidden=2")
        self.channels = channels
        self.internal_dims = internal_dims
        self.AABB = AABB.cuda()
        self.AABB = (self.AABB[0:3], self.AABB[3:6])
        self.min_max = min_max
        self.encoder = _HashGrid(seed)
        mlp_cfg = {"n_input_dims": self.encoder.n_output_dims, "n_output_dims": self.channels, "n_hidden_layers": hidden, "n_neurons": self.internal_dims}
        self.net = _MLP(mlp_cfg, GRADIENT_SCALING)

    def _struct(self):
        st = _lib.MatNet()
        self._keep = [self.encoder.packed()] + [self.net.net[i].weight.detach().contiguous() for i in (0, 2, 4)]
        st.grid_f16 = self._keep[0].data_ptr()
        st.w0, st.w1, st.w2 = (t.data_ptr() for t in self._keep[1:])
        if getattr(self, "_host_consts", None) is None:   # constants of the module: read back once (a .cpu() per call would sync the stream)
            self._host_consts = (self.AABB[0].detach().cpu().tolist(), self.AABB[1].detach().cpu().tolist(),
                                 self.min_max[0].detach().cpu().tolist(), self.min_max[1].detach().cpu().tolist())
        lo, hi, mn, mx = self._host_consts
        st.aabb_min[:] = lo; st.aabb_max[:] = hi; st.out_min[:] = mn; st.out_max[:] = mx
        return st

    def sample(self, texc):
        flat = texc.reshape(-1, 3)
        out = _MatNetFn.apply(self, flat, self.encoder.params, *[self.net.net[i].weight for i in (0, 2, 4)])
        return out.view(*texc.shape[:-1], self.channels)

    @torch.no_grad()
    def sample_no_di(self, texc):
        flat = texc.reshape(-1, 3).contiguous().float()
        n = flat.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=flat.device)
        if n:
            st = self._struct()
            check(lib().mirres_matnet_fwd(C.byref(st), flat.data_ptr(), n, out.data_ptr(), None, stream_ptr()), "mirres_matnet_fwd")
        return out.view(*texc.shape[:-1], self.channels)

    @torch.no_grad()
    def mlp_on_encoding(self, enc_fp16):
        """`self.net.forward(p_enc)` + sigmoid/range (render_helper.py:100-104) on precomputed fp16 encodings [n,32]: the MFMA-tiled kernel."""
        enc = enc_fp16.contiguous()
        assert enc.dtype in (torch.float16, torch.int16) and enc.shape[-1] == 32
        n = enc.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=enc.device)
        if n:
            st = self._struct()
            check(lib().mirres_matnet_mlp(C.byref(st), enc.data_ptr(), n, out.data_ptr(), stream_ptr()), "mirres_matnet_mlp")
        return out

    @torch.no_grad()
    def encode(self, texc):
        """`self.encoder(_texc)` (render_helper.py:97): fp16 hash-grid features [n,32] of world-space positions."""
        flat = texc.reshape(-1, 3).contiguous().float()
        n = flat.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=flat.device)
        enc = torch.empty((n, 32), dtype=torch.float16, device=flat.device)
        if n:
            st = self._struct()
            check(lib().mirres_matnet_fwd(C.byref(st), flat.data_ptr(), n, out.data_ptr(), enc.data_ptr(), stream_ptr()), "mirres_matnet_fwd")
        return enc

    def clamp_(self):
        pass

    def cleanup(self):
        pass
