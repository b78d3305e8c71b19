"""Evaluation meters and result writing of `Trainer.evaluate / test` (nerf/utils.py:477-592, 1319-1395) for the `--test --spp N` use.

PSNRMeter is checked against the reference's own class (tests/golden/gen_reference_losses.py).  SSIMMeter calls torchmetrics'
`structural_similarity_index_measure` in the reference; torchmetrics is not in the reference tree or this image, so `ssim` restates its published
defaults (Wang et al. 2004: 11 x 11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, data range = the larger of the two inputs' value ranges, mean over
the window positions that lie fully inside the image) — unpinned, checked against a direct per-window evaluation.  LPIPSMeter calls the `lpips`
package in the reference (net 'vgg'); neither that package, torchvision nor any pretrained weights are in this image, so `LPIPS` restates the published
network (Zhang et al. 2018, version 0.1: VGG-16 features at relu1_2 ... relu5_3, unit-normalised over channels, squared difference, one non-negative
1 x 1 head per tap, spatial mean, sum over the taps) and takes its weights from files the user supplies (torchvision's vgg16 state dict + the package's
vgg.pth, or one state dict of lpips.LPIPS): unpinned, checked against a plain numpy evaluation of the same formula; without weights it refuses to
construct instead of scoring with random ones.  Images are written as PNG with the standard library (cv2 / imageio are not dependencies).
"""
import os
import struct
import zlib

import numpy as np
import torch

__all__ = ["PSNRMeter", "SSIMMeter", "LPIPSMeter", "LPIPS", "ssim", "to_uint8", "write_png", "write_test_frame"]


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


# VGG-16 `features` indices of the 13 convolutions (torchvision numbering: ReLUs and the 2 x 2 max-pools take the indices between) and the five taps
_VGG_CONVS = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
              (17, 256, 512), (19, 512, 512), (21, 512, 512), (24, 512, 512), (26, 512, 512), (28, 512, 512))
_VGG_POOL_BEFORE = (5, 10, 17, 24)           # a max-pool precedes these convolutions
_VGG_TAPS = (2, 7, 14, 21, 28)               # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 follow these convolutions
_LPIPS_SHIFT, _LPIPS_SCALE = (-0.030, -0.088, -0.188), (0.458, 0.448, 0.450)


class LPIPS(torch.nn.Module):
    """lpips.LPIPS(net='vgg') (version 0.1, lpips=True, spatial=False) as the reference constructs it (nerf/utils.py:522, 796): forward(in0, in1,
    normalize=False) -> [B, 1, 1, 1]; inputs [B, 3, H, W] in [-1, 1], or in [0, 1] with normalize=True (what LPIPSMeter passes).

    Weights: `vgg` = torchvision's vgg16 state dict (keys features.<i>.weight / .bias; classifier.* ignored) and `lin` = the package's weights/v0.1/vgg.pth
    (keys lin<k>.model.1.weight, [1, C, 1, 1]) — each a path for torch.load or a dict —, or `vgg` alone = a state dict of lpips.LPIPS itself
    (net.slice<k>.<i>.weight ..., lin<k>.model.1.weight).  There are no defaults: a perceptual score from random weights would be a number without meaning."""

    def __init__(self, net="vgg", vgg=None, lin=None):
        super().__init__()
        if net != "vgg":
            raise NotImplementedError("LPIPS: the reference uses net='vgg' (nerf/utils.py:516, 796)")
        if vgg is None:
            raise RuntimeError("LPIPS needs pretrained weights (torchvision vgg16 + lpips v0.1 vgg.pth); none ship with this image — pass vgg= / lin=")
        def load(x):
            if not isinstance(x, (str, bytes, os.PathLike)):
                return dict(x)
            try:         # the files are plain state dicts: never unpickle arbitrary objects from a path that comes from the command line / environment
                return torch.load(x, map_location="cpu", weights_only=True)
            except TypeError:      # a torch without the argument
                return torch.load(x, map_location="cpu")
        sd = load(vgg)
        if lin is not None:
            sd = dict(sd); sd.update(load(lin))
        self.convs = torch.nn.ModuleList([torch.nn.Conv2d(ci, co, 3, padding=1) for _, ci, co in _VGG_CONVS])
        self.lins = torch.nn.ModuleList([torch.nn.Conv2d(c, 1, 1, bias=False) for c in (64, 128, 256, 512, 512)])
        packaged = any(k.startswith("net.slice") for k in sd)
        with torch.no_grad():
            for conv, (idx, ci, co) in zip(self.convs, _VGG_CONVS):
                if packaged:         # lpips keeps torchvision's layer numbers inside its five slices
                    sl = 1 + sum(idx >= b - 1 for b in _VGG_POOL_BEFORE)          # the max-pool (index b - 1) opens the next slice
                    key = "net.slice%d.%d" % (sl, idx)
                else:
                    key = "features.%d" % idx
                if key + ".weight" not in sd:
                    raise KeyError("LPIPS: %s.weight missing from the VGG-16 weights" % key)
                w, b = sd[key + ".weight"], sd[key + ".bias"]
                if tuple(w.shape) != (co, ci, 3, 3):
                    raise ValueError("LPIPS: %s.weight has shape %s, VGG-16 needs %s" % (key, tuple(w.shape), (co, ci, 3, 3)))
                conv.weight.copy_(w); conv.bias.copy_(b)
            for k, l in enumerate(self.lins):
                key = "lin%d.model.1.weight" % k
                if key not in sd:
                    raise KeyError("LPIPS: %s missing (the package's weights/v0.1/vgg.pth holds the five heads)" % key)
                l.weight.copy_(sd[key].reshape(l.weight.shape))
        self.register_buffer("shift", torch.tensor(_LPIPS_SHIFT).view(1, 3, 1, 1)); self.register_buffer("scale", torch.tensor(_LPIPS_SCALE).view(1, 3, 1, 1))
        for q in self.parameters():
            q.requires_grad_(False)

    def features(self, x):
        taps = []
        for conv, (idx, _, _) in zip(self.convs, _VGG_CONVS):
            if idx in _VGG_POOL_BEFORE:
                x = torch.nn.functional.max_pool2d(x, 2, 2)
            x = torch.relu(conv(x))
            if idx in _VGG_TAPS:
                taps.append(x)
        return taps

    def forward(self, in0, in1, normalize=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0, f1 = self.features((in0 - self.shift) / self.scale), self.features((in1 - self.shift) / self.scale)
        val = 0
        for a, b, l in zip(f0, f1, self.lins):
            a = a / (torch.sqrt(torch.sum(a * a, dim=1, keepdim=True)) + 1e-10); b = b / (torch.sqrt(torch.sum(b * b, dim=1, keepdim=True)) + 1e-10)
            val = val + l((a - b) ** 2).mean(dim=(2, 3), keepdim=True)
        return val


class LPIPSMeter:
    """nerf/utils.py:515-553: [B, H, W, 3] (or [H, W, 3]) inputs in [0, 1]; update() scores fn(truths, preds, normalize=True) and keeps the running mean.
    `vgg` / `lin` as for LPIPS; when both are None they are taken from MIRRES_LPIPS_VGG / MIRRES_LPIPS_LIN (paths), and without those the meter refuses."""

    def __init__(self, net="vgg", device=None, vgg=None, lin=None):
        self.net = net
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if vgg is None:
            vgg, lin = os.environ.get("MIRRES_LPIPS_VGG"), os.environ.get("MIRRES_LPIPS_LIN")
        self.fn = LPIPS(net=net, vgg=vgg, lin=lin).eval().to(self.device)
        self.clear()

    def clear(self):
        self.V = 0.0; self.N = 0

    def update(self, preds, truths):
        prep = lambda z: (z[None] if z.dim() == 3 else z).permute(0, 3, 1, 2).contiguous().to(self.device, torch.float32)
        with torch.no_grad():
            v = self.fn(prep(truths), prep(preds), normalize=True).item()
        self.V += v; self.N += 1
        return v

    def measure(self):
        return self.V / self.N

    def report(self):
        return "LPIPS (%s) = %.6f" % (self.net, self.measure())


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


# ---------------------------------------------------------------------------------------------- OpenEXR (the material / light maps of Trainer.test)
def write_exr(path, img):
    """What `pyexr.write(path, array)` produces for the maps Trainer.test saves next to the BRDF image (nerf/utils.py:1372-1377: kd, ks, normal, env_map and
    the diffuse / specular light, float32): a single-part scan-line OpenEXR file, channels R, G, B (A) — or Y for one channel — stored as 32-bit floats, no
    compression (pyexr compresses; any EXR reader decodes either).  img: [H, W], [H, W, 1], [H, W, 3] or [H, W, 4]."""
    a = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
    a = np.ascontiguousarray(a, np.float32)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError("write_exr: expected [H, W], [H, W, 1], [H, W, 3] or [H, W, 4]")
    h, w, c = a.shape
    names = {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}[c]
    order = sorted(range(c), key=lambda k: names[k])                     # channels are stored in alphabetical order, per scan line
    attr = lambda name, typ, data: name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(names[k].encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for k in order) + b"\0"   # pixel type 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = (struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box)
            + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
            + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    line_bytes = 4 * w * c
    table_at = len(head)
    first = table_at + 8 * h
    offsets = struct.pack("<%dQ" % h, *[first + y * (8 + line_bytes) for y in range(h)])
    planar = np.ascontiguousarray(a[:, :, order].transpose(0, 2, 1))     # [H, C, W]: one scan line = its channels one after the other
    with open(path, "wb") as fh:
        fh.write(head + offsets)
        for y in range(h):
            fh.write(struct.pack("<ii", y, line_bytes) + planar[y].tobytes())
    return path


def read_exr(path):
    """Reads back what write_exr wrote (uncompressed scan lines, FLOAT channels) -> float32 [H, W, C] in R, G, B (A) / Y order."""
    b = open(path, "rb").read()
    if struct.unpack_from("<i", b, 0)[0] != 20000630:
        raise ValueError("%s: not an OpenEXR file" % path)
    p, attrs = 8, {}
    while b[p] != 0:
        e = b.index(b"\0", p); name = b[p:e].decode(); p = e + 1
        e = b.index(b"\0", p); typ = b[p:e].decode(); p = e + 1
        n = struct.unpack_from("<i", b, p)[0]; p += 4
        attrs[name] = (typ, b[p:p + n]); p += n
    p += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("%s: only uncompressed files (as write_exr makes them)" % path)
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1]); w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, q, cl = [], 0, attrs["channels"][1]
    while cl[q] != 0:
        e = cl.index(b"\0", q); nm = cl[q:e].decode(); q = e + 1
        if struct.unpack_from("<i", cl, q)[0] != 2:
            raise ValueError("%s: only FLOAT channels" % path)
        ch.append(nm); q += 16
    offs = struct.unpack_from("<%dQ" % h, b, p)
    out = np.zeros((h, w, len(ch)), np.float32)
    for y in range(h):
        yy, nb = struct.unpack_from("<ii", b, offs[y])
        out[yy - y0] = np.frombuffer(b, np.float32, w * len(ch), offs[y] + 8).reshape(len(ch), w).T
    want = [n for n in ("R", "G", "B", "A", "Y") if n in ch]
    return out[:, :, [ch.index(n) for n in want]]


def write_test_maps(save_path, name, index, maps):
    """The float maps of Trainer.test's BRDF branch (:1372-1377): `<name>_<i>_{kd,ks,normal,env_map,netrgb_diffuse,netrgb_specular}.png` — OpenEXR content
    under a .png name, as the reference writes them; `normal` is stored as n / 2 + 1 / 2.  maps: dict with any of kd, ks, normal, env_map, rgb_diffuse_light,
    rgb_specular_light ([H, W, 3] each)."""
    os.makedirs(save_path, exist_ok=True)
    files = []
    for key, stem, f in (("kd", "kd", None), ("ks", "ks", None), ("normal", "normal", lambda x: x * 0.5 + 0.5), ("env_map", "env_map", None),
                         ("rgb_diffuse_light", "netrgb_diffuse", None), ("rgb_specular_light", "netrgb_specular", None)):
        if key not in maps or maps[key] is None:
            continue
        m = maps[key]
        m = m.detach().cpu().numpy() if torch.is_tensor(m) else np.asarray(m)
        files.append(write_exr(os.path.join(save_path, "%s_%04d_%s.png" % (name, index, stem)), f(m) if f else m))
    return files
