"""oracle/schedule.py -- TEST INFRASTRUCTURE.  Host-side schedule of the 50-step rectified-flow decode.

Restates (numpy fp32, via the C helpers in selftok_oracle.c):
  * RectifiedFlow.make_schedule 'uniform'  (reference sd3/rectified_flow.py:66-80)
  * `timestep_map[i] ... .long()`          (sd3/rectified_flow.py:203, SelftokPipeline.py:243)
  * DiTi_cont.to_indices / get_position    (diti_utils.py:73-110)
Pinned against the reference by tools/oracle/gen_golden.py -> tests/golden/schedule_*.npz.
"""
from __future__ import annotations

import numpy as np

from . import clib


def make_schedule(num_steps: int = 50, start: float = 1.0):
    """-> dict(scheduled_t, scheduled_t_prev, timestep_map fp32[num_steps], t_long int64[num_steps])"""
    base = clib.linspace(start, 0.0, num_steps + 1)
    scheduled_t = base[:-1].copy()
    scheduled_t_prev = base[1:].copy()
    timestep_map = (scheduled_t * np.float32(1000.0)).astype(np.float32)
    t_long = timestep_map.astype(np.int64)  # .long() truncates toward zero
    return dict(scheduled_t=scheduled_t, scheduled_t_prev=scheduled_t_prev, timestep_map=timestep_map, t_long=t_long)


def parse_stages(stages: str, k_per_stage: str):
    return [int(s) for s in stages.split(",")], [int(s) for s in k_per_stage.split(",")]


def diti_indices(t_long, stages, k_per_stage, K: int) -> np.ndarray:
    return np.array([clib.diti_index(int(t), stages, k_per_stage, K) for t in np.asarray(t_long).reshape(-1)], dtype=np.int64)


def k_table(num_steps: int, stages, k_per_stage, K: int) -> np.ndarray:
    """number-of-visible-tokens index k for every decode step."""
    return diti_indices(make_schedule(num_steps)["t_long"], stages, k_per_stage, K)


def get_position(k):
    return 1000 + k * 8
