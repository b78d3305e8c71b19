"""Test infrastructure: float64 torch restatements (with autograd) of the path's differentiable stages — the same formulas as the CPU oracle's
process_FinalShading / EvaluateFinalSamples_di_ / process_EAWDenoise (oracle/orc_kernels.hpp, orc_brdf.hpp, orc_light.hpp, each following the Slang lines cited
there).  The tests first hold each restatement's FORWARD to the oracle's output on the same inputs (so it is the oracle's function, in double precision), then
compare the HIP adjoints with its autograd gradients element by element.  Every restatement runs on the device of its inputs."""
import math

import torch

LUM = (0.212671, 0.715160, 0.072169)


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def _lum(c):
    return c[..., 0:1] * LUM[0] + c[..., 1:2] * LUM[1] + c[..., 2:3] * LUM[2]


def _schlick3(f0, c):
    return f0 + (1.0 - f0) * torch.clamp(1.0 - c, min=0.0) ** 5


def _lambda_ggx(a2, c):
    cs = torch.where(c <= 0, torch.ones_like(c), c)          # the untaken branch must stay finite for autograd (0 * inf = NaN otherwise)
    c2 = cs * cs
    tan2 = torch.clamp(1.0 - c2, min=0.0) / c2
    return torch.where(c <= 0, torch.zeros_like(c), 0.5 * (-1.0 + torch.sqrt(1.0 + a2 * tan2)))


def final_shading(occ, normal, ray_dir, kd, rm, fdir, fdist, Li):
    """process_FinalShading (FinalShading.slang:14-109) on foreground pixels: returns (color, diffuse light, specular light), zero where the sample is invalid."""
    n = normal
    sign = torch.where(n[:, 2:3] > 0, 1.0, -1.0).to(n.dtype)
    a = -1.0 / (sign + n[:, 2:3])
    b = n[:, 0:1] * n[:, 1:2] * a
    fx = torch.cat((1.0 + sign * n[:, 0:1] * n[:, 0:1] * a, sign * b, -sign * n[:, 0:1]), 1)
    fy = torch.cat((b, sign + n[:, 1:2] * n[:, 1:2] * a, -n[:, 1:2]), 1)
    loc = lambda v: torch.cat((_dot(fx, v), _dot(fy, v), _dot(n, v)), 1)
    valid = (occ > 0.1) & (fdist > 0)
    fdir = torch.where(valid, fdir, n.detach())           # pixels without a sample: any finite direction (their outputs are masked to zero below)
    A, B = loc(-ray_dir), loc(fdir)                       # the view direction and the light direction in the shading frame
    rough, metal = rm[:, 0:1], rm[:, 1:2]
    spec_albedo = 0.04 * (1.0 - metal) + kd * metal
    alpha = rough * rough
    alpha = torch.where(alpha < 1e-4, torch.zeros_like(alpha), alpha)
    pD = _lum(kd) * (1.0 - metal)
    pS = _lum(_schlick3(spec_albedo, _dot(-ray_dir, n))) * (metal + (1.0 - metal))
    low = torch.minimum(A[:, 2:3], B[:, 2:3]) < 1e-6
    diff = torch.where(low | ~(pD > 0) | ~valid, torch.zeros_like(Li), torch.clamp(0.31830988 * B[:, 2:3], min=0.0) * Li)
    h = A + B
    h = h / torch.sqrt(torch.where(low, torch.ones_like(low, dtype=h.dtype), _dot(h, h)))
    a2 = alpha * alpha
    d = (h[:, 2:3] * a2 - h[:, 2:3]) * h[:, 2:3] + 1.0
    D = a2 / (d * d * math.pi)
    G = 1.0 / (1.0 + _lambda_ggx(a2, A[:, 2:3]) + _lambda_ggx(a2, B[:, 2:3]))
    F = _schlick3(spec_albedo, _dot(A, h))
    spec = torch.where(low | (alpha == 0) | ~(pS > 0) | ~valid, torch.zeros_like(Li), F * D * G * 0.25 / torch.where(low, torch.ones_like(A[:, 2:3]), A[:, 2:3]) * Li)
    return kd * (1.0 - metal) * diff + spec, diff, spec


def env_lookup(tex, W, H, d):
    """env_le (lightDi.slang:119-132) of directions d [n,3] (already in the lat-long frame) with the clamp-to-edge, truncating bilinear lookup of helper.slang:46-71:
    tex [H*W,3] -> [n,3]; zero at the poles."""
    theta = torch.acos(d[:, 1])
    s = torch.sin(theta)
    phi = torch.atan2(d[:, 2], d[:, 0])
    phi = torch.where(phi < 0, phi + 6.2831853, phi)
    u, v = phi * 0.1591549, 1.0 - theta * 0.31830988
    x, y = u * W - 0.5, v * H - 0.5
    x0, y0 = torch.trunc(x).long(), torch.trunc(y).long()
    fx_, fy_ = x - x0.to(x.dtype), y - y0.to(y.dtype)
    x1, y1 = (x0 + 1).clamp(0, W - 1), (y0 + 1).clamp(0, H - 1)
    x0, y0 = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    t = lambda yy, xx: tex[yy * W + xx]
    out = (t(y0, x0) * (1 - fx_)[:, None] + t(y0, x1) * fx_[:, None]) * (1 - fy_)[:, None] + (t(y1, x0) * (1 - fx_)[:, None] + t(y1, x1) * fx_[:, None]) * fy_[:, None]
    return torch.where((s.abs() < 1e-4)[:, None], torch.zeros_like(out), out)


def oct_decode(f):
    x, y = f[:, 0] * 2.0 - 1.0, f[:, 1] * 2.0 - 1.0
    z = 1.0 - x.abs() - y.abs()
    t = torch.clamp(-z, 0.0, 1.0)
    x = x + torch.where(x >= 0, -t, t); y = y + torch.where(y >= 0, -t, t)
    n = torch.stack((x, y, z), 1)
    return n / torch.sqrt((n * n).sum(1, keepdim=True))


def eval_final(tex, W, H, light_data, weight, vis):
    """process_EvaluateFinalSamples_di_ (EvaluateFinalSamples.slang:129-188): Li = W * Le(direction of the reservoir's sample), gated by validity and visibility."""
    L = oct_decode(light_data[:, 1:3])
    em = env_lookup(tex, W, H, torch.stack((-L[:, 0], L[:, 2], L[:, 1]), 1))          # ngp_dir
    on = ((light_data[:, 0] > 0.1) & (vis[:, 0] > 0))[:, None]
    return torch.where(on, weight * em, torch.zeros_like(em))


def eaw(fx, fy, step, c_phi, n_phi, p_phi, occ, color, normal, pos):
    """process_EAWDenoise (EAWDenoise.slang:50-302): one 25-tap a-trous pass; background pixels copy their colour."""
    k1 = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0], dtype=color.dtype, device=color.device)
    C = color.view(fy, fx, 3); Nn = normal.view(fy, fx, 3); P = pos.view(fy, fx, 3)
    ys, xs = torch.meshgrid(torch.arange(fy, device=color.device), torch.arange(fx, device=color.device), indexing="ij")
    total = torch.zeros_like(C); cum = torch.zeros((fy, fx, 1), dtype=color.dtype, device=color.device)
    for i in range(25):
        ox, oy = (i % 5) - 2, (i // 5) - 2
        ux, uy = xs + ox * step, ys + oy * step
        ok = ((ux >= 0) & (uy >= 0) & (ux < fx) & (uy < fy))[..., None]
        uxc, uyc = ux.clamp(0, fx - 1), uy.clamp(0, fy - 1)
        ct, nt, pt = C[uyc, uxc], Nn[uyc, uxc], P[uyc, uxc]
        w = torch.clamp(torch.exp(-_dot(C - ct, C - ct) / c_phi), max=1.0) * torch.clamp(torch.exp(-_dot(Nn - nt, Nn - nt) / n_phi), max=1.0) \
            * torch.clamp(torch.exp(-_dot(P - pt, P - pt) / p_phi), max=1.0) * (k1[i % 5] * k1[i // 5] / 256.0)
        w = torch.where(ok, w, torch.zeros_like(w))
        total = total + ct * w; cum = cum + w
    out = total / cum
    return torch.where((occ.view(fy, fx, 1) < 0.1), C, out).reshape(-1, 3)
