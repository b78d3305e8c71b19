"""Evaluation meters / result files.  PSNRMeter against the reference's own class (ref_losses.npz); SSIM (torchmetrics-defined,
unpinned) against a direct per-window evaluation and its invariants; PNG files read back with PIL."""
import os

import numpy as np
import pytest
import torch

from mirres_restir_nerf_mesh_amd import meters

G = os.path.join(os.path.dirname(__file__), "golden", "ref_losses.npz")


def test_psnr_meter_against_the_reference_class():
    g = np.load(G)
    m = meters.PSNRMeter()
    vals = [m.update(torch.from_numpy(a), torch.from_numpy(b)) for a, b in zip(g["psnr_pred"], g["psnr_truth"])]
    np.testing.assert_allclose(vals, g["psnr_each"], rtol=1e-12)
    assert abs(m.measure() - float(g["psnr_mean"])) < 1e-12 and m.report() == str(g["psnr_report"])
    m.clear()
    assert m.N == 0


def _ssim_direct(p, t, K=11, sigma=1.5):
    x = np.arange(K) - (K - 1) / 2
    g = np.exp(-(x / sigma) ** 2 / 2); g /= g.sum(); w = np.outer(g, g)
    L = max(p.max() - p.min(), t.max() - t.min()); c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    B, C, H, W = p.shape
    acc = []
    for b in range(B):
        one = []
        for c in range(C):
            for y in range(H - K + 1):
                for xx in range(W - K + 1):
                    a, d = p[b, c, y:y + K, xx:xx + K], t[b, c, y:y + K, xx:xx + K]
                    ma, md = (w * a).sum(), (w * d).sum()
                    va, vd, cv = (w * a * a).sum() - ma * ma, (w * d * d).sum() - md * md, (w * a * d).sum() - ma * md
                    one.append((2 * ma * md + c1) * (2 * cv + c2) / ((ma * ma + md * md + c1) * (va + vd + c2)))
        acc.append(np.mean(one))
    return float(np.mean(acc))


def test_ssim_against_a_direct_evaluation_and_invariants():
    rng = np.random.default_rng(5)
    t = rng.random((2, 3, 20, 17)); p = np.clip(t + 0.1 * rng.standard_normal(t.shape), 0, 1)
    got = float(meters.ssim(torch.from_numpy(p), torch.from_numpy(t)))
    assert abs(got - _ssim_direct(p.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64))) < 2e-5
    x = torch.from_numpy(t)
    assert abs(float(meters.ssim(x, x)) - 1.0) < 1e-6
    assert abs(float(meters.ssim(torch.from_numpy(p), x)) - float(meters.ssim(x, torch.from_numpy(p)))) < 1e-7      # symmetric
    assert float(meters.ssim(1 - x, x)) < 0.0 < got < 1.0                                                              # anti-correlated structure
    with pytest.raises(ValueError):
        meters.ssim(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))
    m = meters.SSIMMeter()
    v = m.update(torch.from_numpy(p[0]).permute(1, 2, 0), torch.from_numpy(t[0]).permute(1, 2, 0))                     # [H, W, 3] as Trainer.evaluate passes
    m.update(x.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1))
    assert abs(m.measure() - (v + 1.0) / 2) < 1e-6 and m.report().startswith("SSIM = ")


def test_png_files_read_back(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    img = rng.random((13, 21, 3)).astype(np.float32); img[0, 0] = (1.0, 0.0, 0.999)
    depth = rng.random((13, 21)).astype(np.float32) * 5 + 1
    files = meters.write_test_frame(str(tmp_path / "results_brdf"), "ngp_ep0003", 4, torch.from_numpy(img), depth)
    assert [os.path.basename(f) for f in files] == ["ngp_ep0003_0004_rgb_brdf.png", "ngp_ep0003_0004_depth.png"]
    back = np.asarray(Image.open(files[0]))
    assert back.shape == (13, 21, 3) and np.array_equal(back, (img * 255).astype(np.uint8)) and back[0, 0].tolist() == [255, 0, 254]   # truncation
    d = np.asarray(Image.open(files[1]))
    assert d.shape == (13, 21) and d.min() == 0 and d.max() == 254          # (max - min) / (max - min + 1e-6) stays just below 1
    rgba = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    meters.write_png(str(tmp_path / "a.png"), rgba)
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "a.png"))), rgba)
    with pytest.raises(ValueError):
        meters.write_png(str(tmp_path / "b.png"), rgba.astype(np.float32))


def test_exr_maps_round_trip(tmp_path):
    """meters.write_exr / read_exr / write_test_maps: the float maps Trainer.test saves with pyexr (nerf/utils.py:1372-1377) — header fields of a scan-line
    OpenEXR file, channels in alphabetical order on disk and R, G, B in memory, values bit-exact through the round trip, the reference's file names."""
    import struct
    from mirres_restir_nerf_mesh_amd import meters
    rng = np.random.default_rng(3)
    img = (rng.random((7, 11, 3)) * 4 - 1).astype(np.float32); img[0, 0] = [np.inf, -0.0, 1e-40]
    p = meters.write_exr(str(tmp_path / "a.exr"), img)
    raw = open(p, "rb").read()
    assert struct.unpack_from("<ii", raw, 0) == (20000630, 2) and b"channels\0chlist\0" in raw and raw.index(b"B\0") < raw.index(b"G\0") < raw.index(b"R\0")
    back = meters.read_exr(p)
    assert back.shape == img.shape and np.array_equal(back.view(np.uint32), img.view(np.uint32))
    for c in (1, 4):
        a = rng.random((5, 6, c)).astype(np.float32)
        assert np.array_equal(meters.read_exr(meters.write_exr(str(tmp_path / ("c%d.exr" % c)), a)), a)
    files = meters.write_test_maps(str(tmp_path / "brdf"), "ngp_ep0010", 3, {"kd": img, "normal": np.zeros((7, 11, 3), np.float32), "env_map": img[:4, :8]})
    assert [os.path.basename(f) for f in files] == ["ngp_ep0010_0003_kd.png", "ngp_ep0010_0003_normal.png", "ngp_ep0010_0003_env_map.png"]
    assert np.array_equal(meters.read_exr(files[1]), np.full((7, 11, 3), 0.5, np.float32))        # normal * 0.5 + 0.5
    with pytest.raises(ValueError):
        meters.write_exr(str(tmp_path / "bad.exr"), np.zeros((4, 4, 2), np.float32))
