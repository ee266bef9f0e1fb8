"""selftoktokenizer_amd: MI355X-native Selftok encode/decode hot path.

Drop-in host API: `mimogpt.infer.SelftokPipeline` (shim package at the repo root) ->
`selftoktokenizer_amd.pipeline.SelftokPipeline`.  The compute path is hand-written
gfx950 HIP behind the C ABI in include/selftok_hip.h (libselftok_hip.so), plus
PyTorch-ROCm GEMMs/convs.  There is no CPU fallback: ops raise if the HIP library
is missing.
"""
__version__ = "0.1.0"
