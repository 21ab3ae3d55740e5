"""rtgs -- MI355X-native drop-in for the render path of fangjunzhou/rt-gaussian-splat-renderer.

Same module and class names as the reference's ``rtgs`` package (ray, gaussian, camera, scene,
ray_tracer, utils); the device work is done by hand-written HIP kernels for gfx950 behind the C
ABI of include/rtgs_hip.h.  The reference exports nothing at top level (its __init__.py:1 only
imports rtgs.utils); the main classes are re-exported here for convenience.
"""

from . import utils  # noqa: F401
from .camera import Camera  # noqa: F401
from .gaussian import BOUNDING_THRESHOLD, Gaussian, GaussianField, new_gaussian  # noqa: F401
from .ray import Ray, RayField, new_ray  # noqa: F401
from .ray_tracer import RayTracer  # noqa: F401
from .scene import Scene, SceneHit  # noqa: F401

__all__ = ["Camera", "Gaussian", "GaussianField", "new_gaussian", "BOUNDING_THRESHOLD", "Ray",
           "RayField", "new_ray", "RayTracer", "Scene", "SceneHit", "utils"]
