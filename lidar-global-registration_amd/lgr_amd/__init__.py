"""lgr_amd -- Python host side of the MI355X-native global-registration hot path.

Only plumbing lives here (ctypes binding of the C ABI, synthetic scan-pair generator, pair sharding across ranks);
all computation happens in liblgr_hip.so (hand-written HIP for gfx950).
"""
from . import synthetic  # noqa: F401  (numpy only)


def load_capi():
    """Import the C-ABI binding lazily (raises ImportError when liblgr_hip.so has not been built)."""
    import importlib
    return importlib.import_module(".capi", __name__)
