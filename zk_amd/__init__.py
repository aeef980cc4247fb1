"""zk_amd -- MI355X (gfx950) implementation of the iammadab/zk sumcheck / MLE-fold / NTT hot path.

`zk_amd.api` mirrors the reference's public Rust API for that path over the C ABI in include/zk_amd.h
(libzk_amd.so: hand-written HIP kernels + host protocol logic).  No CPU fallback exists.
"""
from .api import *  # noqa: F401,F403
from . import gkr  # noqa: F401,E402
