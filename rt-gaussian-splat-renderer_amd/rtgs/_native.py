"""ctypes binding of the C ABI in include/rtgs_hip.h (librtgs_hip.so, hipcc --offload-arch=gfx950).

There is deliberately no fallback: if the shared library is missing or a call fails, the caller
gets an exception.  torch is imported first so that the library binds to the same libamdhip64
(soname libamdhip64.so.7) that torch already loaded -- streams and device pointers are only
meaningful inside one HIP runtime instance.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_size_t, c_uint32, c_void_p

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

ABI_VERSION = 7
MAX_DEPTH = 16
RAY_FLOATS = 8
SH_COEFFS = 15

TRACE_RESUME = 1
TRACE_ACCUMULATE = 2
TRACE_COMPACT = 4
# (8 and 16 were the two-waves-per-tile form's switches up to ABI v6: retired, the bits are ignored)
TRACE_TILE_LIST = 32   # the shard is RtgsTraceParams.tile_list / tile_count (an empty list renders nothing)
TRACE_REGION = 64      # the shard lies inside RtgsTraceParams.region: its projection skips what cannot reach it
TRACE_BINS_ON = 128    # (ignored since round 5: bins are the default for every shard)
TRACE_BINS_OFF = 256   # experiments: no per-frame slab bins for the call's own projection (exact ties may resolve differently)
PEEL_FRESH = 1

LIB_PATH = os.environ.get("RTGS_HIP_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "_lib", "librtgs_hip.so")


class RtgsScene(Structure):
    _fields_ = [("n", c_int32), ("position", c_void_p), ("rotation", c_void_p), ("scale", c_void_p),
                ("color_opacity", c_void_p), ("sh", c_void_p)]


class RtgsCamera(Structure):
    _fields_ = [("position", c_float * 3), ("rotation", c_float * 4), ("width", c_int32),
                ("height", c_int32), ("fx", c_float), ("fy", c_float), ("device_pose", c_void_p)]


def _camera_dtype():
    """RtgsCamera as a NumPy record (HOST arrays of poses for rtgs_shard_frames are filled through it)."""
    import numpy as np
    dt = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("width", "<i4"), ("height", "<i4"),
                   ("fx", "<f4"), ("fy", "<f4"), ("device_pose", "<u8")], align=True)
    assert dt.itemsize == ctypes.sizeof(RtgsCamera) and dt.fields["device_pose"][1] == RtgsCamera.device_pose.offset
    return dt


CAMERA_DTYPE = _camera_dtype()


class RtgsTraceParams(Structure):
    _fields_ = [("depth", c_int32), ("t_cutoff", c_float), ("tile_first", c_int32),
                ("tile_step", c_int32), ("flags", c_uint32), ("tile_order", c_void_p),
                ("tile_list", c_void_p), ("tile_count", c_int32), ("region", c_float * 4)]


# name -> (restype, argtypes); mirrors include/rtgs_hip.h one to one.
PROTOTYPES = {
    "rtgs_abi_version": (c_int32, []),
    "rtgs_strerror": (c_char_p, [c_int32]),
    "rtgs_covariance": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_generate_rays": (c_int32, [POINTER(RtgsCamera), c_void_p, c_void_p]),
    "rtgs_workspace_bytes": (c_size_t, [c_int32]),
    "rtgs_workspace_bytes_frame": (c_size_t, [c_int32, c_int32, c_int32]),
    "rtgs_workspace_bytes_frames": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "rtgs_max_frames_per_launch": (c_int32, []),
    "rtgs_pose_bytes": (c_size_t, []),
    "rtgs_pose_pack": (c_int32, [POINTER(RtgsCamera), c_void_p]),
    "rtgs_tile_size": (c_int32, []),
    "rtgs_num_tiles": (c_int32, [c_int32, c_int32]),
    "rtgs_num_local_tiles": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "rtgs_launch_order": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rtgs_launch_order_list": (c_int32, [c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "rtgs_trace": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), POINTER(RtgsTraceParams),
                             c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_int32,
                             c_void_p]),
    "rtgs_project": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), c_void_p, c_size_t,
                               c_void_p]),
    "rtgs_project_shard": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), c_int32, c_int32, c_void_p,
                                     c_size_t, c_void_p]),
    "rtgs_peel_list_bytes": (c_size_t, [c_int32, c_int32]),
    "rtgs_trace_lists": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), c_void_p, c_void_p, c_size_t, c_void_p,
                                   c_size_t, c_int32, c_void_p]),
    "rtgs_peel": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), c_void_p, c_size_t, c_int32, c_uint32, c_void_p,
                            c_void_p, c_void_p]),
    "rtgs_peel_start": (c_int32, [POINTER(RtgsCamera), c_void_p, c_size_t, c_int32, c_void_p, c_int32, c_void_p]),
    "rtgs_untile": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "rtgs_event_create": (c_int32, [POINTER(c_void_p)]),
    "rtgs_event_destroy": (c_int32, [c_void_p]),
    "rtgs_event_query": (c_int32, [c_void_p]),
    "rtgs_shard_frame": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), POINTER(RtgsTraceParams), c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int32]),
    "rtgs_shard_frames": (c_int32, [POINTER(RtgsScene), c_void_p, c_int32, POINTER(RtgsTraceParams), c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                    c_void_p, c_int32]),
    "rtgs_frames_land": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                   c_void_p, c_void_p]),
    "rtgs_frames_land_map": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                       c_void_p, c_void_p]),
    "rtgs_untile_map": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "rtgs_closest_hit": (c_int32, [POINTER(RtgsScene), c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "rtgs_accel_bytes": (c_size_t, [c_int32]),
    "rtgs_accel_build": (c_int32, [POINTER(RtgsScene), c_void_p, c_size_t, c_void_p]),
    "rtgs_closest_hit_accel": (c_int32, [POINTER(RtgsScene), c_void_p, c_size_t, c_int32, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "rtgs_sample_step_rays_accel": (c_int32, [POINTER(RtgsScene), c_void_p, c_size_t, c_int32, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "rtgs_precompute": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_render": (c_int32, [POINTER(RtgsScene), POINTER(RtgsCamera), c_int32, c_float, c_int32, c_int32,
                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rtgs_sample_step_rays": (c_int32, [POINTER(RtgsScene), c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_gaussian_hit": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "rtgs_gaussian_eval": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_quat_op": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_ray_get": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rtgs_disp_buffer": (c_int32, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                   c_void_p]),
    "rtgs_fill": (c_int32, [c_void_p, c_size_t, c_float, c_void_p]),
}

_lib = None


class RtgsError(RuntimeError):
    """A negative return code from the C ABI."""


def load_library(path: str, extra: dict | None = None) -> ctypes.CDLL:
    """dlopen a build of the library and attach the prototypes (every symbol of include/rtgs_hip.h
    must resolve; `extra` adds build-specific ones, e.g. the counting build's rtgs_debug_counters)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    handle = ctypes.CDLL(path)
    for name, (restype, argtypes) in {**PROTOTYPES, **(extra or {})}.items():
        fn = getattr(handle, name)  # AttributeError = missing export
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.rtgs_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI {got} != binding ABI {ABI_VERSION}")
    return handle


def lib() -> ctypes.CDLL:
    """Load librtgs_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = load_library(LIB_PATH)
    return _lib


def tile_size() -> int:
    """Macro-tile edge (pixels) the library was built with: the unit of sharding."""
    return lib().rtgs_tile_size()


def check(code: int, what: str = "rtgs") -> None:
    """Map the ABI's integer error convention onto Python exceptions."""
    if code == 0:
        return
    msg = lib().rtgs_strerror(code).decode()
    if code in (-1, -2, -3):
        raise ValueError(f"{what}: {msg} (code {code})")
    raise RtgsError(f"{what}: {msg} (code {code})")


def require_device(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a torch tensor on the HIP device (got "
                           f"{type(t).__name__} on {getattr(t, 'device', '?')}); "
                           "this path has no CPU implementation")
    if t.dtype not in (torch.float32, torch.int32) or not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous float32/int32")
    return t


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


# torch.cuda.current_stream() costs ~8 us per call (device-index parsing, availability checks):
# three of them per frame were a third of the tile-parallel loop's host time.  The two private
# accessors below are what it ends in; fall back to the public path where they are missing.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_stream = getattr(torch._C, "_cuda_getCurrentStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """The hipStream_t of torch's current stream, as the void* the ABI takes."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def current_stream(device_index: int) -> "torch.cuda.Stream":
    """torch.cuda.current_stream(device_index) without its argument handling."""
    if _cur_stream is not None:
        sid, didx, dtype = _cur_stream(device_index)
        return torch.cuda.Stream(stream_id=sid, device_index=didx, device_type=dtype)
    return torch.cuda.current_stream(device_index)


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible to torch; the rtgs render path needs an MI355X")
    return torch.device("cuda", torch.cuda.current_device())


def make_camera(position, rotation, width, height, fx, fy, device_pose=None) -> RtgsCamera:
    c = RtgsCamera()
    c.device_pose = device_pose
    for k in range(3):
        c.position[k] = float(position[k])
    for k in range(4):
        c.rotation[k] = float(rotation[k])
    c.width, c.height = int(width), int(height)
    c.fx, c.fy = float(fx), float(fy)
    return c
