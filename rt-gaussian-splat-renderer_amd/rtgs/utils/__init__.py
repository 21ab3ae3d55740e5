"""Utility modules mirroring the reference's rtgs.utils package (src/rtgs/utils/)."""
from . import math, quaternion, types  # noqa: F401
