"""embedding_amd — MI355X-native random-walk + SGNS engine (drop-in for the hot path of thekingofkings/embedding).

The package is a thin host layer over libdge.so (hand-written HIP for gfx950, C ABI in include/dge.h).
Importing it loads the library; there is no CPU fallback.
"""
from ._native import DgeError, TrainConfig, TrainStats, lib, LIB_PATH  # noqa: F401
from .engine import DeviceGraph, SgnsModel, WalkCorpus, deepwalk_config, host_sync_count, make_config, tuning  # noqa: F401
