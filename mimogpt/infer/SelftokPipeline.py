"""Drop-in module path of the reference pipeline; the implementation is selftoktokenizer_amd.pipeline."""
from selftoktokenizer_amd.pipeline import NormalizeToTensor, SelftokPipeline, norm_ip  # noqa: F401
