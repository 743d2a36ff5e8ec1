"""sgaligner_amd -- MI355X-native (gfx950) implementation of SGAligner's node-embedding + matching hot path.

Importing the package is side-effect free; the HIP C-ABI library (csrc/ -> libsga_hip.so) is loaded on
first use by `sgaligner_amd._lib` and its absence is a hard error (there is no CPU fallback).
"""
__version__ = '0.1.0'
