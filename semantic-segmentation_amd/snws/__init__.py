"""MI355X-native normalized-conv hot path: host-side Python over the libsnws C ABI."""
from . import _lib  # noqa: F401
