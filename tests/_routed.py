"""N library contexts on one device standing for the N ranks of a multi-GPU run (test
infrastructure shared by tests/test_gpu_parity.py and tests/fuzz_gpu.py)."""

import numpy as np

from compairr_amd import HipOverlap



