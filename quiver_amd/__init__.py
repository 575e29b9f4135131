"""quiver_amd — MI355X-native similarity-search hot path for Quiver.

The product is ``libqv.so`` (HIP kernels behind the C ABI of include/qv.h).  This
package is the thin Python host side used by the tests and the benchmark: a
ctypes binding (``_lib``), a row-numbered device index (``DeviceIndex``) and the
mirrors of the reference's host interfaces for this path (``hybrid``, ``hnsw``,
``core``, ``vectortypes``) — same names, argument meaning and error behaviour as
the Go packages they stand in for.
"""
from ._lib import QvError, lib, load_library, METRICS, metric_id  # noqa: F401
from .device_index import DeviceIndex, DeviceGraph, GraphReplicas, ShardedIndex  # noqa: F401

__all__ = ["QvError", "lib", "load_library", "METRICS", "metric_id", "DeviceIndex", "DeviceGraph", "GraphReplicas", "ShardedIndex"]
