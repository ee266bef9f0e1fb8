"""Host-side schedule of the rectified-flow decode (product code; numpy fp32, no torch kernels involved).

Reproduces bit-for-bit what the reference computes with torch-CPU:
  RectifiedFlow.make_schedule('uniform')  sd3/rectified_flow.py:66-80   (torch.linspace fp32 on CPU)
  timestep_map[i] -> .long()              sd3/rectified_flow.py:203 ; SelftokPipeline.py:243
  DiTi_cont.to_indices / get_position     diti_utils.py:73-110
  encoder mask  arange(K) <= k            models_ours.py:345-353   (here: just the visible count k+1)
The fragile part is float: linspace(1,0,51)*1000 truncates to 459, 399, ... not 460, 400;
tests/test_schedule.py pins this module to tests/golden/schedule.npz (captured from the reference).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _fma32(a, b, c):
    # a*b is exact in float64 for fp32 inputs and the sum below fits 53 bits for the magnitudes used here,
    # so one rounding to fp32 == fmaf(a, b, c)
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def linspace32(start: float, end: float, steps: int) -> np.ndarray:
    """ATen CPU linspace for fp32: step=(end-start)/(steps-1); i < steps/2 counts up with an fma,
    the rest counts down from `end`."""
    start, end = f32(start), f32(end)
    step = f32((end - start) / f32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = _fma32(step, f32(i), start)
        else:
            out[i] = f32(end - f32(step * f32(steps - 1 - i)))
    return out


class FlowSchedule:
    """RectifiedFlow(num_steps, start=1.0, val_schedule='uniform', shift=1.0) buffers."""

    def __init__(self, num_steps: int = 50, start: float = 1.0):
        base = linspace32(start, 0.0, num_steps + 1)
        self.num_timesteps = num_steps
        self.scheduled_t = base[:-1].copy()
        self.scheduled_t_prev = base[1:].copy()
        self.timestep_map = (self.scheduled_t * f32(1000.0)).astype(np.float32)
        self.one_minus_scheduled_t = (f32(1.0) - self.scheduled_t).astype(np.float32)
        self.t_long = self.timestep_map.astype(np.int64)          # .long(): truncation toward zero
        self.dt = (self.scheduled_t - self.scheduled_t_prev).astype(np.float32)   # a_t - a_prev in fp32


class DiTiCont:
    """DiTi_cont: piecewise-linear timestep -> index of the last visible token."""

    def __init__(self, n_timesteps: int, K: int, stages: str, k_per_stage: str):
        assert stages and k_per_stage
        self.K = K
        self.k_per_stage = [int(k) for k in k_per_stage.split(",")]
        self.stages = [0] + [int(s) for s in stages.split(",")]
        self.segments = []
        acc = 0
        for i, kp in enumerate(self.k_per_stage):
            slope = float(kp) / (self.stages[i + 1] - self.stages[i])
            self.segments.append((self.stages[i], slope, acc))
            acc += kp

    def to_indices(self, t_long) -> np.ndarray:
        """t_long: integer timesteps.  int64 tensor * python float -> fp32 product, truncated."""
        t = np.asarray(t_long, dtype=np.int64)
        ind = np.zeros_like(t)
        for low, slope, base in self.segments:
            xp = t - low
            val = (xp.astype(np.float32) * f32(slope)).astype(np.int64) + base
            ind = np.where(xp >= 0, val, ind)
        return np.clip(ind, 0, self.K - 1)

    @staticmethod
    def get_position(k):
        return 1000 + k * 8


def decode_plan(num_steps: int, diti: DiTiCont):
    """-> (FlowSchedule, k[num_steps]) : everything the decode loop needs from the host."""
    flow = FlowSchedule(num_steps)
    return flow, diti.to_indices(flow.t_long)
