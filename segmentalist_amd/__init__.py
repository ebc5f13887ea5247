"""
segmentalist_amd -- MI355X-native implementation of the segmentalist hot path behind the
reference's own Python surface (`import segmentalist_amd as segmentalist`).

Module names, class names, method names, keyword arguments and record-dict keys follow
kamperh/segmentalist; the arithmetic of the hot path runs in hand-written HIP kernels
(libsegk.so, include/segk.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
