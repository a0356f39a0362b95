"""deplex_amd — MI355X-native drop-in for prime-slam/deplex's PlaneExtractor::process().

`Config`, `PlaneExtractor` mirror `deplex.plane_extraction`; `deplex_amd.utils.DepthImage` mirrors
`deplex.utils.DepthImage`.  All compute happens in the HIP library behind include/deplex_amd.h.
"""
from . import utils  # noqa: F401
from ._capi import DeplexError  # noqa: F401
from .plane_extraction import Config, Pipeline, PlaneExtractor  # noqa: F401
from .sharding import ShardedExtractor, pinned_empty, shard_range  # noqa: F401

__all__ = ["Config", "PlaneExtractor", "Pipeline", "ShardedExtractor", "pinned_empty", "shard_range", "DeplexError", "utils"]
