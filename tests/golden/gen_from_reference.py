"""Generates tests/golden/ref_python.npz by IMPORTING the reference's own Python (this container only: /root/reference is absent on the GPU box).

What can be imported of the hot path: everything else is Slang->CUDA or needs slangpy/tinycudann/nvdiffrast.
  * nerf/render_dump.py:safe_l2_normalize            — used by run_restir_di_with_pt (renderer_restir.py:486)
  * nerf/ScreenSpaceReSTIR/GenerateLightTiles.py:make_sampleable — the torch half (cumsum / sum / normalisation / forced 1.0 entries) of the
    importance tables; its two kernel launches are served by a fake module `m` that runs this repo's oracle restatement of those kernels.
  * nerf/ScreenSpaceReSTIR/Denoising.py:EAWDenoise_use_phi / EAWDenoise_use_phi_no_di — the a-trous driver (iteration count, stepWidth /= 2
    with int() at the launch, ping-pong of the colour buffer); its kernel launches are served the same way, so the fixture pins the DRIVER
    (what mirres_render's finish and mirres-restir_nerf_mesh_amd/Denoising.py restate), not the kernel.
The fixtures are data (inputs + outputs); no reference source text is stored.

    python tests/golden/gen_from_reference.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    from oracle import oracle as O
    rd = load("nerf/render_dump.py", "ref_render_dump")
    glt = load("nerf/ScreenSpaceReSTIR/GenerateLightTiles.py", "ref_glt")
    rng = np.random.default_rng(11)
    x = rng.normal(size=(64, 3)).astype(np.float32)
    x[0] = 0.0; x[1] = [1e-9, 0, 0]; x[2] = [3, -4, 0]
    norm_out = rd.safe_l2_normalize(torch.from_numpy(x), dim=-1).numpy()

    # make_sampleable: fake slang module whose kernels are the oracle's restatement; torch.zeros(device='cuda') redirected to CPU
    Hc, Wc = 12, 20
    env = (rng.random((Hc, Wc, 3)) * 2).astype(np.float32)
    env[2:5] = 0.0                                 # three adjacent black rows -> the middle one takes the uniform fallback (row_weight < 1e-4)
    tex = O.flip_env(env)

    class Launch:
        def __init__(self, fn): self.fn = fn
        def launchRaw(self, blockSize=None, gridSize=None): self.fn()

    class FakeM:
        def make_sampleable(self, env_tex, weight, width, height):
            def run():
                pdf = np.zeros(width * height, np.float32); cdf = np.zeros((width + 1) * height, np.float32)
                mp = np.zeros(height, np.float32); mc = np.zeros(height + 1, np.float32)
                # oracle kernel part only: un-normalised weights are what the Slang kernel writes (make_sampleable.slang:34-60)
                w = O.env_weights(env_tex.numpy(), width, height)
                weight.copy_(torch.from_numpy(w).reshape(-1, 1))
            return Launch(run)
        def Distribution2D(self, w, h, pdf_, cdf_):
            def run():
                p, c = O.distribution2d(pdf_.numpy().ravel().copy(), cdf_.numpy().ravel().copy(), w, h)
                pdf_.copy_(torch.from_numpy(p).reshape(pdf_.shape)); cdf_.copy_(torch.from_numpy(c).reshape(cdf_.shape))
            return Launch(run)

    real_zeros = torch.zeros
    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)
    torch.zeros = cpu_zeros
    try:
        pdf_, cdf_, mpdf_, mcdf_ = glt.make_sampleable(FakeM(), torch.from_numpy(tex), Wc, Hc)
    finally:
        torch.zeros = real_zeros
    # ---- EAW driver
    den = load("nerf/ScreenSpaceReSTIR/Denoising.py", "ref_denoising")
    fx, fy = 24, 20; Np = fx * fy
    occ = (rng.random((Np, 1)) > 0.2).astype(np.float32)
    col = rng.random((Np, 3)).astype(np.float32)
    # a gently curved sheet (the filter's normal / position weights must not vanish: phi = (2.0, 0.1, 0.001))
    yy, xx = np.meshgrid(np.arange(fy, dtype=np.float32), np.arange(fx, dtype=np.float32), indexing="ij")
    pos = np.stack([xx * 0.01, yy * 0.01, 0.02 * np.sin(xx * 0.3) * np.cos(yy * 0.2)], -1).reshape(Np, 3).astype(np.float32)
    nrm = np.stack([-0.3 * np.cos(xx * 0.3) * 0.2, 0.2 * np.sin(yy * 0.2) * 0.2, np.ones_like(xx)], -1).reshape(Np, 3).astype(np.float32)
    nrm += rng.normal(size=(Np, 3)).astype(np.float32) * 0.02; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    launches = []

    class FakeDen:
        def _run(self, PHI, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map, out_color):
            def run():
                launches.append(int(stepWidth))
                o = O.eaw(framedim_x, framedim_y, stepWidth, PHI[0], PHI[1], PHI[2], occ_map.numpy(), color.numpy(), normal_map.numpy(), pos_map.numpy())
                out_color.copy_(torch.from_numpy(np.asarray(o)).reshape(out_color.shape))
            return Launch(run)
        def process_EAWDenoise(self, **k): return self._run(**k)
        def process_EAWDenoise_no_di(self, **k): return self._run(**k)

    torch.zeros = cpu_zeros
    try:
        t = lambda a_: torch.from_numpy(a_.copy())
        eaw_di = den.EAWDenoise_use_phi(FakeDen(), 2.0, 0.1, 0.001, 2, 2, fx, fy, t(occ), t(col), t(nrm), t(pos)).numpy()
        eaw_nodi = den.EAWDenoise_use_phi_no_di(FakeDen(), 2.0, 0.1, 0.001, 4, 3, fx, fy, t(occ), t(col), t(nrm), t(pos)).numpy()
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_python.npz"), norm_in=x, norm_out=norm_out, env=env,
                        pdf=pdf_.numpy().ravel(), cdf=cdf_.numpy().ravel(), mpdf=mpdf_.numpy().ravel(), mcdf=mcdf_.numpy().ravel(),
                        eaw_dims=np.array([fx, fy], np.int32), eaw_occ=occ, eaw_col=col, eaw_nrm=nrm, eaw_pos=pos, eaw_di=eaw_di, eaw_nodi=eaw_nodi,
                        eaw_steps=np.array(launches, np.int32))
    print("wrote ref_python.npz")


if __name__ == "__main__":
    main()
