"""mirres-mi355x: MI355X-native engine behind the reference's ReSTIR path-tracing operator surface.

Host side mirrors nerf/renderer_restir.py, nerf/ScreenSpaceReSTIR/{Resampling,GenerateLightTiles,Denoising}.py and
nerf/render_helper.py:MLPTexture3D of brabbitdousha/MIRReS-ReSTIR_Nerf_mesh; all compute goes through the C ABI of
libmirres.so (include/mirres.h). There is no CPU fallback: without the built library every operator raises.
"""
from . import _lib  # noqa: F401
from . import scene  # noqa: F401

__all__ = ["_lib", "scene"]


def __getattr__(name):
    # torch-facing modules are imported lazily so that `import mirres_restir_nerf_mesh_amd` stays cheap
    import importlib
    if name in ("renderer_restir", "Resampling", "Denoising", "GenerateLightTiles", "render_helper", "dist", "harness", "losses", "raster", "render_dump", "checkpoint", "meters"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
