"""peakachu_amd -- MI355X-native scoring hot path of Peakachu (v2.3).

Window gather -> distance-normalise -> Gaussian blur -> min-max scale ->
Random-Forest predict_proba -> threshold, behind the reference's own
Chromosome / CLI / bedpe boundary (peakachu/scoreUtils.py:9-135).  Host code
is Python; the compute is hand-written HIP for gfx950 reached through the
C ABI declared in include/peakachu_hip.h (ctypes only).
"""
__version__ = "0.1.0"
