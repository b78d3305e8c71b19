"""Mirror of nerf/ScreenSpaceReSTIR/GenerateLightTiles.py (reference) on the MI355X engine."""
import torch

from ._lib import lib, check, stream_ptr
from ._ops import _f32


def make_sampleable(m, env_map, width, height):
    """GenerateLightTiles.py:4-29 -> (pdf_ [Hc*Wc,1], cdf_ [Hc*(Wc+1),1], mpdf_ [Hc,1], mcdf_ [Hc+1,1]).
    env_map: vertically flipped, flattened [Hc*Wc,3] (renderer_restir.py:305-311)."""
    env_map = _f32(env_map.contiguous())
    width, height = int(width), int(height)
    dev = env_map.device
    pdf_ = torch.empty((width * height, 1), dtype=torch.float32, device=dev)
    cdf_ = torch.empty(((width + 1) * height, 1), dtype=torch.float32, device=dev)
    mpdf_ = torch.empty((height, 1), dtype=torch.float32, device=dev)
    mcdf_ = torch.empty((height + 1, 1), dtype=torch.float32, device=dev)
    check(lib().mirres_env_make_sampleable(env_map.data_ptr(), width, height, pdf_.data_ptr(), cdf_.data_ptr(), mpdf_.data_ptr(), mcdf_.data_ptr(),
                                           stream_ptr()), "mirres_env_make_sampleable")
    return pdf_, cdf_, mpdf_, mcdf_


def GenerateLightTiles(m, debug_out, env_tex, pdf_, cdf_, mpdf_, mcdf_, width, height, frameIndex, light_data, light_uv, light_inv_pdf,
                       light_tile_count=128, light_tile_size=1024):
    """GenerateLightTiles.py:31-52. Mutates light_data / light_uv / light_inv_pdf in place."""
    cfg = m.ctx.cfg
    if (int(light_tile_count), int(light_tile_size)) != (cfg.light_tile_count, cfg.light_tile_size):
        raise ValueError("light tile shape differs from the context's configuration")
    check(lib().mirres_light_tiles(m.ctx.h, _f32(env_tex).data_ptr(), int(width), int(height), _f32(pdf_).data_ptr(), _f32(cdf_).data_ptr(),
                                   _f32(mpdf_).data_ptr(), _f32(mcdf_).data_ptr(), int(frameIndex) & 0xffffffff, light_data.data_ptr(),
                                   light_uv.data_ptr() if light_uv is not None else None, light_inv_pdf.data_ptr(), stream_ptr()), "mirres_light_tiles")
    return 'hello'
