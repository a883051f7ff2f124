"""MI355X-native MeshGraphNets Encode-Process-Decode engine (hot path of una-auxme/MeshGraphNets.jl).

The directory name carries a dot, so import it through the repo-root shim:  `import mgn_amd`.
"""
from . import synth  # noqa: F401
from ._capi import LIB_PATH, MGN_DEVICE_NONE, PROTOTYPES, load  # noqa: F401
from .engine import (Engine, FeatureGraph, GraphNetwork, MgnError, edge_features_native,  # noqa: F401
                     run_forward_staged, run_processor_staged, step, triangles_to_edges_native, world_edges_native)
