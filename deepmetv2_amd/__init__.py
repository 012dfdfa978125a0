"""deepmetv2_amd -- MI355X-native operators for the DeepMETv2 DynamicEdgeConv -> MET hot path.

Drop-in names (same spelling and argument meaning as the third-party operators the reference imports at
/root/reference/model/graph_met_network.py:7,9, model/net.py:8, train.py:10):

    from deepmetv2_amd import EdgeConv, DynamicEdgeConv      # torch_geometric.nn
    from deepmetv2_amd import knn_graph, radius_graph, knn   # torch_cluster
    from deepmetv2_amd import scatter_add, scatter_max       # torch_scatter

All of them run hand-written HIP kernels for gfx950 through the C ABI in include/dmet.h
(deepmetv2_amd/libdmet_hip.so, built by `python -m deepmetv2_amd.build`).  There is no CPU implementation:
calling an operator without the library or with non-GPU tensors raises.
"""
from .cluster import knn, knn_graph, knn_table, radius_graph, radius_table
from .conv import DynamicEdgeConv, EdgeConv
from .data import Batch, DeviceLoader, EventLoader, collate, events_from_padded
from .graph import GraphFuture, NeighborTable, build_async, raise_deferred_errors, register_batch, to_undirected
from .metrics import metrics, resolution, u_perp_par_loss
from .scatter import met_reduce, scatter_add, scatter_max
from .nn import accelerate

__all__ = [
    "EdgeConv", "DynamicEdgeConv", "knn", "knn_graph", "knn_table", "radius_graph", "radius_table",
    "scatter_add", "scatter_max", "met_reduce", "NeighborTable", "register_batch", "metrics", "resolution",
    "u_perp_par_loss", "to_undirected", "raise_deferred_errors", "accelerate", "build_async", "GraphFuture", "Batch", "EventLoader", "DeviceLoader", "collate", "events_from_padded",
]
__version__ = "0.1.0"
