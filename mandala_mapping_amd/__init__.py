"""mandala_mapping_amd — MI355X-native scan-registration engine for the m3d pipeline (host side).

Only what the hot path needs: the ctypes binding of libm3dreg.so (csrc/), the PointCloud2 payload
codec, the seeded synthetic workloads, and the pair-sharding helper for multi-GPU batches.
The HIP library is loaded lazily by `mandala_mapping_amd.binding`; importing this package never
needs a GPU.
"""
from . import abi  # noqa: F401
