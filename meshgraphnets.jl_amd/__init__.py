"""MI355X-native MeshGraphNets Encode-Process-Decode engine (hot path of una-auxme/MeshGraphNets.jl).

The directory name carries a dot, so import it through the repo-root shim:  `import mgn_amd`.
"""
from . import synth  # noqa: F401
from ._capi import (LIB_PATH, MGN_COMM_HOST, MGN_COMM_ID_BYTES, MGN_COMM_RCCL, MGN_DEVICE_NONE, MGN_E_RCCL, PROTOTYPES,  # noqa: F401
                    load)
from .engine import (Engine, FeatureGraph, GraphNetwork, MgnError, edge_features_native,  # noqa: F401
                     run_forward_staged, run_processor_staged, step,
                     load as load_network, save as save_network,  # GraphNetCore's load / save! (`load` here is the library loader)
                     triangles_to_edges_native, world_edges_native)
