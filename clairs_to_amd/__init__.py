"""clairs_to_amd: MI355X-native hot path of ClairS-TO (pileup tensor -> AFF/NEG inference -> posterior).

Importing the package loads the HIP extension (libclairsto_amd.so) and raises ImportError when it has not
been built; nothing in here falls back to a CPU implementation."""
from . import _lib  # noqa: F401  (fails loudly if the extension is missing)

__all__ = ["_lib"]
