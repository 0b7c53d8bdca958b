"""sortmerna_amd -- MI355X (gfx950) engine for SortMeRNA's per-read hot path.

The product is libsmr_hip.so (hand-written HIP kernels behind the C ABI of include/smr_hip.h); this package is
its thin host-side mirror of the reference driver (processor.cpp:align()).  There is no CPU fallback: importing
the engine without hipcc-built code or without a GPU raises.
"""
from .engine import Engine, Index, Reads, SmrError, align, align_resident, default_params, minimal_score, pigeonhole_layout  # noqa: F401
