"""ggnn_amd -- MI355X-native GGNN query + build engine (drop-in for the `ggnn` module surface).

The compute path is hand-written HIP for gfx950 in ggnn_amd/csrc (libggnn_amd.so) behind the
C-ABI of include/ggnn_c.h; this package is the thin host side.  There is no CPU fallback.
"""
from .api import (GGNN, DistanceMeasure, Evaluation, Evaluator, FloatDataset, Graph, IntDataset,
                  UCharDataset, set_log_level)

__all__ = ["GGNN", "DistanceMeasure", "Evaluation", "Evaluator", "FloatDataset", "Graph",
           "IntDataset", "UCharDataset", "set_log_level"]
__version__ = "0.1.0"
