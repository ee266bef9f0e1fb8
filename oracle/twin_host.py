"""TEST INFRASTRUCTURE: run the product's own host modules (selftoktokenizer_amd.encoder / mmdit / pipeline._Flow) on the CPU, with every
C-ABI call going to oracle/libselftok_cpu.so (the header's symbols compiled for the CPU, host pointers) and every GEMM to torch-CPU --
the "build's own CPU restatement via libselftok_cpu.so + torch-CPU GEMMs" of SURVEY.md section 8d.  Used by bench.py's cpu_baseline leg
and by tests/test_cpu_twin_host.py; the product never imports this (it has no CPU path: ops.* refuse CPU tensors)."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_twin():
    from selftoktokenizer_amd import _lib
    path = os.path.join(_HERE, "libselftok_cpu.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


@contextlib.contextmanager
def on_cpu_twin():
    """inside the block `ops.*` accept CPU tensors and call the CPU twin (stream = NULL); everything is restored on exit"""
    from selftoktokenizer_amd import _lib, ops
    twin = load_twin()
    saved = (_lib.load, _lib._lib, ops._need_cuda, ops._stream)
    _lib.load = lambda: twin
    _lib._lib = twin
    ops._need_cuda = lambda *a: None
    ops._stream = lambda: None
    try:
        yield twin
    finally:
        _lib.load, _lib._lib, ops._need_cuda, ops._stream = saved


def build_encoder(sd_cpu, K: int = 512):
    import torch
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    return QformerEncoderGPU(sd_cpu, torch.device("cpu"), K)


def build(sd_cpu, K: int = 512):
    """(encoder, MMDiT, flow, k_table) of the product's host classes on CPU tensors; call inside `on_cpu_twin()`"""
    import torch
    from selftoktokenizer_amd.config import default_config
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    from selftoktokenizer_amd.pipeline import _Flow
    from selftoktokenizer_amd.schedule import DiTiCont
    cpu = torch.device("cpu")
    p = default_config(K).tokenizer.params
    diti = DiTiCont(1000, K, p.stages, p.k_per_stage)
    flow = _Flow(50, 1.0, cpu)
    return QformerEncoderGPU(sd_cpu, cpu, K), MMDiTGPU(sd_cpu, cpu, K), flow, diti.to_indices(flow.t_long)
