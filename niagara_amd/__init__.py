"""niagara_amd — MI355X-native visibility front-end with niagara's drawcull -> tasksubmit -> clustercull dispatch
contract.  The product is the C-ABI library (include/niagara_vis.h, niagara_amd/csrc); this package is the thin
Python host layer used by the tests and the benchmark."""
from . import layouts  # noqa: F401
from ._lib import EXPORTS, NvError, PyramidDesc, SO_PATH, lib  # noqa: F401
from . import host, synth  # noqa: F401

__all__ = ["layouts", "host", "synth", "lib", "NvError", "PyramidDesc", "EXPORTS", "SO_PATH"]
