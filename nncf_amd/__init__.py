"""nncf_amd -- MI355X (gfx950) implementation of NNCF's weight-compression lowering hot path.

Host side = this package (Python, mirrors the reference's function seam); device side =
libnncf_amd_wc.so (hand-written HIP kernels behind the C ABI in include/nncf_amd_wc.h)."""
from nncf_amd.parameters import CompressedWeight, CompressWeightsMode, WeightCompressionConfig  # noqa: F401

__version__ = "0.1.0"
