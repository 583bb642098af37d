"""lamsa_amd -- MI355X-native hot path of the LAMSA long-read aligner.

The package holds the HIP kernels + C-ABI library (csrc/ -> lib/liblamsa_hp.so), the host
side of `lamsa aln` (host/) and thin ctypes bindings used by tests and bench (hp.py).
There is no CPU fallback: importing works anywhere, computing needs the built library and
a gfx950 device.
"""
from .hp import LamsaHp, HpPara, load_library, LIB_PATH  # noqa: F401
