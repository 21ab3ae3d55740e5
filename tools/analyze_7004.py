import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
from fuzz_case import make
from oracle import rtgs_ref as ref
import importlib
seed, pi, pj = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
