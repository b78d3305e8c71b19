"""CPU: meters.LPIPS / LPIPSMeter (nerf/utils.py:515-553, 794-796) — the published LPIPS v0.1 'vgg' formula on user-supplied weights.
No pretrained weights exist in this image, so the network is checked on RANDOM weights against a plain float64 numpy evaluation of the formula written
here from the paper's definition (3 x 3 convolutions as sums over shifted windows), in both state-dict layouts it accepts; what it can never be is pinned to
the package's numbers."""
import os

import numpy as np
import pytest
import torch

from mirres_restir_nerf_mesh_amd import meters

CONVS = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256), (17, 256, 512), (19, 512, 512), (21, 512, 512),
         (24, 512, 512), (26, 512, 512), (28, 512, 512))


def _weights(seed=0):
    g = torch.Generator().manual_seed(seed)
    vgg = {}
    for idx, ci, co in CONVS:
        vgg["features.%d.weight" % idx] = torch.randn((co, ci, 3, 3), generator=g) * (2.0 / (9 * ci)) ** 0.5
        vgg["features.%d.bias" % idx] = torch.randn((co,), generator=g) * 0.05
    vgg["classifier.0.weight"] = torch.zeros(4, 4)                                       # present in torchvision's file, ignored
    lin = {"lin%d.model.1.weight" % k: torch.rand((1, c, 1, 1), generator=g) for k, c in enumerate((64, 128, 256, 512, 512))}
    return vgg, lin


def _numpy_lpips(vgg, lin, x0, x1):
    """float64, straight from the definition; x [3, H, W] in [-1, 1]."""
    shift = np.array([-0.030, -0.088, -0.188])[:, None, None]; scale = np.array([0.458, 0.448, 0.450])[:, None, None]
    def conv(x, w, b):
        C, H, W = x.shape
        xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
        out = np.zeros((w.shape[0], H, W))
        for dy in range(3):
            for dx in range(3):
                out += np.einsum("oc,chw->ohw", w[:, :, dy, dx], xp[:, dy:dy + H, dx:dx + W])
        return out + b[:, None, None]
    def feats(x):
        x = (x - shift) / scale
        taps = []
        for idx, _, _ in CONVS:
            if idx in (5, 10, 17, 24):
                C, H, W = x.shape
                x = x.reshape(C, H // 2, 2, W // 2, 2).max(axis=(2, 4))
            x = np.maximum(conv(x, vgg["features.%d.weight" % idx].double().numpy(), vgg["features.%d.bias" % idx].double().numpy()), 0)
            if idx in (2, 7, 14, 21, 28):
                taps.append(x)
        return taps
    total = 0.0
    for k, (a, b) in enumerate(zip(feats(x0), feats(x1))):
        a = a / (np.sqrt((a * a).sum(0, keepdims=True)) + 1e-10); b = b / (np.sqrt((b * b).sum(0, keepdims=True)) + 1e-10)
        w = lin["lin%d.model.1.weight" % k].double().numpy().reshape(-1)
        total += float(np.mean(np.einsum("c,chw->hw", w, (a - b) ** 2)))
    return total


def test_lpips_matches_the_definition_on_random_weights(tmp_path):
    vgg, lin = _weights()
    fn = meters.LPIPS(vgg=vgg, lin=lin).eval()
    g = torch.Generator().manual_seed(3)
    a = torch.rand((2, 3, 16, 32), generator=g); b = (a + 0.2 * torch.randn((2, 3, 16, 32), generator=g)).clamp(0, 1)
    got = fn(a, b, normalize=True)
    assert got.shape == (2, 1, 1, 1)
    for i in range(2):
        want = _numpy_lpips(vgg, lin, 2 * a[i].double().numpy() - 1, 2 * b[i].double().numpy() - 1)
        assert abs(float(got[i]) - want) < 2e-5 * max(1.0, abs(want)), (float(got[i]), want)
    assert float(got.min()) > 1e-4                                                                       # the images differ: the score is not trivially zero
    assert float(fn(a, a, normalize=True).abs().max()) == 0.0                                            # identical images
    assert torch.allclose(fn(a, b, normalize=True), fn(b, a, normalize=True), rtol=1e-6, atol=1e-8)      # symmetric
    assert torch.allclose(fn(2 * a - 1, 2 * b - 1), got, rtol=1e-6, atol=1e-8)                           # normalize=True is the [0,1] -> [-1,1] map
    # the two file layouts: torchvision vgg16 + the package's vgg.pth, and one state dict of lpips.LPIPS (net.slice<k>.<i>...)
    pv, pl = str(tmp_path / "vgg16.pth"), str(tmp_path / "vgg.pth")
    torch.save(vgg, pv); torch.save(lin, pl)
    assert torch.equal(meters.LPIPS(vgg=pv, lin=pl).eval()(a, b, normalize=True), got)
    packaged = dict(lin)
    for idx, _, _ in CONVS:
        sl = 1 if idx < 4 else 2 if idx < 9 else 3 if idx < 16 else 4 if idx < 23 else 5
        for part in ("weight", "bias"):
            packaged["net.slice%d.%d.%s" % (sl, idx, part)] = vgg["features.%d.%s" % (idx, part)]
    packaged["scaling_layer.shift"] = torch.zeros(1, 3, 1, 1)
    assert torch.equal(meters.LPIPS(vgg=packaged).eval()(a, b, normalize=True), got)


def test_lpips_meter_interface_and_refusals(tmp_path, monkeypatch):
    vgg, lin = _weights(1)
    m = meters.LPIPSMeter(device=torch.device("cpu"), vgg=vgg, lin=lin)
    g = torch.Generator().manual_seed(4)
    pred = torch.rand((16, 16, 3), generator=g); truth = torch.rand((16, 16, 3), generator=g)
    v1 = m.update(pred, truth); v2 = m.update(pred[None], pred[None])
    assert v1 > 0 and v2 == 0.0 and abs(m.measure() - v1 / 2) < 1e-12 and m.report().startswith("LPIPS (vgg) = ")
    want = float(meters.LPIPS(vgg=vgg, lin=lin)(truth.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None], normalize=True))
    assert abs(v1 - want) < 1e-7                                                                          # fn(truths, preds, normalize=True), nerf/utils.py:540
    m.clear(); assert m.N == 0
    monkeypatch.delenv("MIRRES_LPIPS_VGG", raising=False); monkeypatch.delenv("MIRRES_LPIPS_LIN", raising=False)
    with pytest.raises(RuntimeError, match="pretrained"):
        meters.LPIPSMeter(device=torch.device("cpu"))                                                     # no weights: no score
    with pytest.raises(NotImplementedError):
        meters.LPIPS(net="alex", vgg=vgg, lin=lin)
    with pytest.raises(KeyError):
        meters.LPIPS(vgg=vgg)                                                                             # the five heads are missing
    bad = dict(vgg); bad["features.5.weight"] = torch.zeros(64, 64, 3, 3)
    with pytest.raises(ValueError):
        meters.LPIPS(vgg=bad, lin=lin)
    pv, pl = str(tmp_path / "a.pth"), str(tmp_path / "b.pth"); torch.save(vgg, pv); torch.save(lin, pl)
    monkeypatch.setenv("MIRRES_LPIPS_VGG", pv); monkeypatch.setenv("MIRRES_LPIPS_LIN", pl)
    m2 = meters.LPIPSMeter(device=torch.device("cpu"))
    assert abs(m2.update(pred, truth) - v1) < 1e-9


def test_stage1_loss_perceptual_term():
    """nerf/utils.py:1079-1082: lambda_lpips x criterion(pred, gt) for the BRDF image (and the NeRF image when present), on the [0, 1] images as they are."""
    from types import SimpleNamespace
    from mirres_restir_nerf_mesh_amd import losses
    vgg, lin = _weights(2)
    crit = meters.LPIPS(vgg=vgg, lin=lin)
    H, W = 16, 16
    g = torch.Generator().manual_seed(5)
    out = {"image_brdf": torch.rand((H * W, 3), generator=g).requires_grad_(True), "diffuse_light": torch.rand((H * W, 3), generator=g),
           "specular_light": torch.rand((H * W, 3), generator=g), "img_brdf_indirect": torch.zeros(H * W, 3),
           "kd_grad": torch.rand((H * W, 3), generator=g), "ks_grad": torch.rand((H * W, 3), generator=g), "normal_grad": torch.rand((H * W, 3), generator=g)}
    gt = torch.rand((H * W, 3), generator=g)
    base = losses.stage1_loss(out, gt, gt, SimpleNamespace(use_brdf=True))
    with pytest.raises(ValueError):
        losses.stage1_loss(out, gt, gt, SimpleNamespace(use_brdf=True, lambda_lpips=0.5))
    full = losses.stage1_loss(out, gt, gt, SimpleNamespace(use_brdf=True, lambda_lpips=0.5), criterion_lpips=crit, frame_hw=(H, W))
    want = 0.5 * crit(out["image_brdf"].view(1, H, W, 3).permute(0, 3, 1, 2), gt.view(1, H, W, 3).permute(0, 3, 1, 2))
    assert full.dim() == 0 and abs(float((full - base).detach()) - float(want.detach())) < 1e-6 and float(want.detach()) > 0
    full.backward()
    assert out["image_brdf"].grad is not None and float(out["image_brdf"].grad.abs().sum()) > 0           # the term trains the image, the network's weights stay fixed
    assert all(not q.requires_grad for q in crit.parameters())
