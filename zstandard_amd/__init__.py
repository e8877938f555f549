"""MI355X-native Zstandard block codec: HIP kernels (csrc/) behind a C ABI (include/zsmi.h), with a host-side
mirror of the reference's public API (api.py)."""
from .api import ZStdDecompress, ZstdDecompressor, ZstdCompressor, BatchCodec   # noqa: F401
