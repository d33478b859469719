"""nncf_amd -- MI355X (gfx950) implementation of NNCF's weight-compression lowering hot path.

Host side = this package (Python, mirrors the reference's function seam); device side =
libnncf_amd_wc.so (hand-written HIP kernels behind the C ABI in include/nncf_amd_wc.h)."""
from nncf_amd.parameters import (AdvancedAWQParameters, AdvancedCompressionParameters, AdvancedGPTQParameters,  # noqa: F401
                                 AdvancedScaleEstimationParameters, BackupMode, CompressedWeight, CompressWeightsMode,
                                 Dataset, GroupSizeFallbackMode, IgnoredScope, SensitivityMetric,
                                 WeightCompressionConfig)


def compress_weights(model, **kwargs):
    """nncf.compress_weights for a torch model on an AMD GPU: see nncf_amd.quantize_model.compress_weights."""
    from nncf_amd.quantize_model import compress_weights as _impl

    return _impl(model, **kwargs)

__version__ = "0.1.0"
