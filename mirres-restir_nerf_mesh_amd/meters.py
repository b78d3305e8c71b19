"""Evaluation meters and result writing of `Trainer.evaluate / test` (nerf/utils.py:477-592, 1319-1395) for the `--test --spp N` use.

PSNRMeter is checked against the reference's own class (tests/golden/gen_reference_losses.py).  SSIMMeter calls torchmetrics'
`structural_similarity_index_measure` in the reference; torchmetrics is not in the reference tree or this image, so `ssim` restates its published
defaults (Wang et al. 2004: 11 x 11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, data range = the larger of the two inputs' value ranges, mean over
the window positions that lie fully inside the image) — unpinned, checked against a direct per-window evaluation.  LPIPS needs its pretrained
network and is not provided.  Images are written as PNG with the standard library (cv2 / imageio are not dependencies).
"""
import os
import struct
import zlib

import numpy as np
import torch

__all__ = ["PSNRMeter", "SSIMMeter", "ssim", "to_uint8", "write_png", "write_test_frame"]


class PSNRMeter:
    """nerf/utils.py:477-513 (max pixel value 1): running mean of -10 log10(mse) per update."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.V = 0.0; self.N = 0

    def update(self, preds, truths):
        a, b = (x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x) for x in (preds, truths))
        v = -10 * np.log10(np.mean((a - b) ** 2))
        self.V += v; self.N += 1
        return v

    def measure(self):
        return self.V / self.N

    def report(self):
        return "PSNR = %.6f" % self.measure()


def ssim(preds, truths, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03, data_range=None):
    """Mean structural similarity of [B, C, H, W] images (see the module note for the definition)."""
    if preds.shape != truths.shape or preds.dim() != 4:
        raise ValueError("ssim: expected two [B, C, H, W] tensors of one shape")
    if min(preds.shape[2:]) < kernel_size:
        raise ValueError("ssim: image smaller than the %d x %d window" % (kernel_size, kernel_size))
    p, t = preds.to(torch.float32), truths.to(torch.float32)
    if data_range is None:
        data_range = torch.maximum(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    x = torch.arange(kernel_size, dtype=torch.float32, device=p.device) - (kernel_size - 1) / 2
    g = torch.exp(-(x / sigma) ** 2 / 2); g = g / g.sum()
    C = p.shape[1]
    win = (g[:, None] * g[None, :]).expand(C, 1, kernel_size, kernel_size).contiguous()
    f = lambda z: torch.nn.functional.conv2d(z, win, groups=C)          # window positions fully inside the image
    mp, mt = f(p), f(t)
    spp, stt, spt = f(p * p) - mp * mp, f(t * t) - mt * mt, f(p * t) - mp * mt
    m = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (spp + stt + c2))
    return m.reshape(m.shape[0], -1).mean(-1).mean()


class SSIMMeter:
    """nerf/utils.py:555-591: [B, H, W, 3] (or [H, W, 3]) inputs in [0, 1], running mean of the per-update SSIM."""

    def __init__(self, device=None):
        self.device = device
        self.clear()

    def clear(self):
        self.V = 0.0; self.N = 0

    def update(self, preds, truths):
        prep = lambda z: (z[None] if z.dim() == 3 else z).permute(0, 3, 1, 2).contiguous().to(self.device if self.device is not None else z.device)
        v = float(ssim(prep(preds), prep(truths)))
        self.V += v; self.N += 1
        return v

    def measure(self):
        return self.V / self.N

    def report(self):
        return "SSIM = %.6f" % self.measure()


def to_uint8(img):
    """`(pred * 255).astype(np.uint8)` of Trainer.test (:1353-1357): truncation, not rounding; the image is in [0, 1] by construction (clamped first
    here, where numpy's cast of an out-of-range float is undefined)."""
    a = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
    return (np.clip(a, 0.0, 1.0) * 255).astype(np.uint8)


def write_png(path, img_u8):
    """8-bit grey / RGB / RGBA PNG ([H, W], [H, W, 3] or [H, W, 4] uint8), no filtering."""
    a = np.ascontiguousarray(img_u8)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3, 4)):
        raise ValueError("write_png: expected uint8 [H, W], [H, W, 3] or [H, W, 4]")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    h, w = a.shape[:2]
    ctype = 0 if a.ndim == 2 else (2 if a.shape[2] == 3 else 6)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, -1)], axis=1).tobytes()
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_test_frame(save_path, name, index, image_brdf, depth=None):
    """The per-frame files Trainer.test writes for the BRDF branch with write_video off (:1361-1366): `<name>_<i>_rgb_brdf.png` and, when a depth
    map is given, `<name>_<i>_depth.png` (min-max normalised with the 1e-6 guard)."""
    os.makedirs(save_path, exist_ok=True)
    out = [os.path.join(save_path, "%s_%04d_rgb_brdf.png" % (name, index))]
    write_png(out[0], to_uint8(image_brdf))
    if depth is not None:
        d = depth.detach().cpu().numpy() if torch.is_tensor(depth) else np.asarray(depth)
        d = (d - d.min()) / (d.max() - d.min() + 1e-6)
        out.append(os.path.join(save_path, "%s_%04d_depth.png" % (name, index)))
        write_png(out[1], (d * 255).astype(np.uint8))
    return out
