"""scaledreamer_amd — MI355X-native Asynchronous-Score-Distillation inner loop.

Python host code (this package) mirrors ScaleDreamer's plugin surface and calls hand-written HIP for
gfx950 through the C ABI of include/asd_hip.h (scaledreamer_amd/libasd_hip.so).
"""
__version__ = "0.1.0"
