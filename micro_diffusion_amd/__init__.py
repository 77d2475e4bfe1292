"""MI355X-native MicroDiT training path (hand-written gfx950 HIP kernels behind a C ABI)."""
__version__ = "0.1.0"
