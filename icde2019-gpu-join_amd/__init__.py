"""icde2019-gpu-join_amd — MI355X-native radix-partitioned hash join (drop-in for the in-GPU join
path of psiul/ICDE2019-GPU-Join).  The product is csrc/ (gfx950 HIP kernels behind the C ABI of
include/hj.h); this package is the thin host-side mirror used by tests, bench.py and the multi-GPU
driver.  The directory name is not a Python identifier: load it with `__graft_entry__.load_package()`."""
from . import _lib, generator  # noqa: F401
from .join import (ECAPACITY, EHIP, EINVAL, ENOMEM, HJError, HashJoin, PAYLOAD_GIVEN, PAYLOAD_ONES, PAYLOAD_ROWID, REL_R, REL_S,  # noqa: F401
                   hashJoinClusteredProbe, host_join, host_split, host_split_blocks, shard_of)

build = _lib.build
