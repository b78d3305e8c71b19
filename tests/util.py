"""Shared fixtures for the parity tests: a small synthetic frame whose every stage is produced by the CPU oracle."""
import numpy as np

CONFIG = dict(light_tile_count=128, light_tile_size=1024, neighbor_offset_count=8192)


class SmallFrame:
    """Mesh + G-buffer + env tables built with the oracle only (no GPU). fx x fy pixels, constant or textured materials."""

    def __init__(self, O, S, fx=48, fy=40, subdiv=3, ground=16, env_hw=(32, 64), seed=0, rough=0.45, metal=0.0, varied=True):
        import os
        sweep = int(os.environ.get("MIRRES_TEST_SEED", "0"))      # robustness sweeps: another view, other materials, same tests
        seed += sweep
        self.O, self.fx, self.fy = O, fx, fy
        N = fx * fy
        self.N = N
        self.vert, self.tri = S.make_mesh(subdiv, ground)
        self.info, self.aabb, self.sorted, self.height = O.bvh_build(self.vert, self.tri)
        eye, rd = S.camera_rays(fy, fx, 30.0 + 47.0 * sweep, 30.0 - 7.0 * (sweep % 5)) if sweep else S.camera_rays(fy, fx)
        self.eye = eye
        self.ray_dir_raw = rd
        rays = O.make_rays(np.repeat(eye[None], N, 0), rd)
        r = O.trace(self.info, self.aabb, self.vert, self.tri, rays, True)
        self.occ = r["hit"].astype(np.float32)
        self.pos = r["pos"].copy()
        self.normal = np.where(self.occ[:, None] > 0, r["normal"], 0).astype(np.float32)
        self.depth = np.linalg.norm(self.pos - eye, axis=1).astype(np.float32)
        rng = np.random.default_rng(seed)
        if varied:
            self.kd = (0.25 + 0.6 * rng.random((N, 3))).astype(np.float32)
            self.rm = np.stack([0.15 + 0.7 * rng.random(N), metal + 0.3 * rng.random(N)], 1).astype(np.float32)
        else:
            self.kd = np.full((N, 3), 0.6, np.float32)
            self.rm = np.stack([np.full(N, rough), np.full(N, metal)], 1).astype(np.float32)
        self.env = S.make_env(env_hw[0], env_hw[1], sun=25.0)
        self.Hc, self.Wc = env_hw
        self.tex = O.flip_env(self.env)
        self.tables = O.make_sampleable(self.tex, self.Wc, self.Hc)
        # derived maps of restir_di_with_pt (:279-287) and run_restir_di_with_pt (:484-486)
        self.ray_dir = (rd / np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-6)).astype(np.float32)
        self.normal_depth = np.concatenate([self.normal, self.depth[:, None]], 1).astype(np.float32)
        k, m, r_ = self.kd, self.rm[:, 1], self.rm[:, 0]
        b0 = (k[:, 0] * np.float32(0.2126) + k[:, 1] * np.float32(0.7152)) + k[:, 2] * np.float32(0.0722)
        b1 = (m * np.float32(0.2126) + m * np.float32(0.7152)) + m * np.float32(0.0722)
        a = np.clip(r_, np.float32(0.01), np.float32(1.0))
        self.brdf = np.stack([b0, b1, a * a], 1).astype(np.float32)
        self.noff = O.neighbor_offsets(8192)
        self.keep = O.Keep()
        self.frame = O.make_frame(self.keep, fx, fy, self.occ, self.pos, self.normal_depth, self.brdf, self.ray_dir, (self.info, self.aabb), self.vert,
                                  self.tri, self.tex, self.Wc, self.Hc, self.tables)


def match_fraction(a, b, rtol=1e-4, atol=1e-6):
    """Fraction of rows whose every component agrees within tolerance."""
    a = np.asarray(a, np.float64).reshape(len(a), -1); b = np.asarray(b, np.float64).reshape(len(b), -1)
    ok = np.all(np.abs(a - b) <= atol + rtol * np.abs(b), axis=1)
    return float(ok.mean()), ok


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)
