"""ampligraph_amd -- MI355X-native engine behind AmpliGraph's ScoringBasedEmbeddingModel
fit()/predict()/evaluate() surface.  Compute lives in libamdkge.so (hand-written HIP for gfx950,
C ABI in include/amdkge.h); this package is the Python host that mirrors the reference's API."""
__version__ = "0.1.0"

from . import _ffi  # noqa: F401
