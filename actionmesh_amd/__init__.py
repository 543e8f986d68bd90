"""actionmesh_amd: MI355X-native (gfx950) implementation of ONE hot path of
facebookresearch/actionmesh - the Stage-I temporal-3D flow-matching denoise loop
(`actionmesh/model/temporal_denoiser.py` + `actionmesh/scheduler/*`) - behind the
reference's own plug-in seams.  The arithmetic lives in libactionmesh_amd.so
(hand-written HIP, C-ABI in include/actionmesh_amd.h); this package is the
host-side mirror of the reference interface.  No CPU fallback.
"""
from ._lib import LIB_PATH, HipLibraryMissing  # noqa: F401
from .autoencoder import HipAutoencoder  # noqa: F401
from .image_encoder import HipImageEncoder  # noqa: F401
from .denoiser import HipDenoiser, HipEngine, WindowCache  # noqa: F401
from .mesh_io import create_animated_glb, load_glb, save_deformation, save_meshes  # noqa: F401
from . import actionbench  # noqa: F401
from .dropin import install, uninstall  # noqa: F401
from .scheduler import ClassifierFreeGuidance, HipSchedulerFlow  # noqa: F401
from .sharding import FrameShardPlan  # noqa: F401
from .windows import LatentBank, chunk_from, denoise_window, generate_3d_latents, generate_vertex_animation  # noqa: F401

__all__ = ["HipAutoencoder", "HipImageEncoder", "HipDenoiser", "HipEngine", "HipSchedulerFlow", "ClassifierFreeGuidance",
           "FrameShardPlan", "WindowCache", "HipLibraryMissing", "LIB_PATH",
           "LatentBank", "chunk_from", "denoise_window", "generate_3d_latents", "generate_vertex_animation", "save_deformation",
           "save_meshes", "create_animated_glb", "load_glb", "actionbench", "install", "uninstall"]
