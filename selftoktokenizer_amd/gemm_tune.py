"""Picking hipBLASLt's fp32 GEMM kernels for the MMDiT block Linears by measurement.

The fp32 headline is 87 % library GEMM (PyTorch-ROCm -> hipBLASLt), and hipBLASLt's own heuristic leaves 3 - 8 % of it on the table at
this model's shapes: [B (k+1), 1536] x {[1536, 4608], [1536, 1536], [1536, 6144], [6144, 1536]} runs at 0.84 / 0.87 / 0.92 / 0.95 of the
fp32 matrix peak with the default choice and at 0.94 / 0.89 / 0.95 / 0.96 with the kernel PyTorch's TunableOp finds by exhaustive search
(tools/bench_fp32_gemm_shapes.py, profiles/r3_fp32_gemm_default_vs_tuned.txt).  The exhaustive search costs ~10 s per shape and a 50-step
decode has 50 distinct context lengths -- 204 shapes, half an hour -- so it is not run.  Instead:

  * CANDIDATES = the solutions that search returned on four row counts (16 shapes);
  * at the first decode of a batch size, every candidate (and the default) is timed on two representative row counts of each of the four
    (N, K) families -- a probe is a TunableOp results file that maps a not-otherwise-used row count to the candidate, read back with
    tuning disabled, two passes, the faster one counts -- about 4 s in total;
  * the winner of each family is written for the row counts of the step (TunableOp file, read back) -- except the small ones at which it loses to the default
    (three extra probes at the low end of the row list, round 6: a kernel chosen at 16384 rows can lose 3x at 1280); a family whose default wins gets no
    entry.  Nothing here can be slower than the default by more than the timing noise.

The candidate names are solution indices of ONE hipBLASLt build (`FOUND_WITH`).  On any other build an index may not exist -- TunableOp
then raises inside the probing `F.linear` (`TORCH_CHECK(iter != ops_.end())`, not "ignored") -- or may name a kernel that does not
support the shape.  So: when `torch.cuda.tunable.get_validators()` differs from `FOUND_WITH`, `autotune_linears` does nothing and says
so; and every probe runs under try/except, a candidate that fails is dropped (and its key withdrawn by a default-kernel entry).

OPT-IN (`SelftokPipeline(..., tune_gemm=True)` or `pipe.tune_linears(batch)`; bench.py asks for it).  Process state:
TunableOp is only enabled (tuning OFF) inside `enabled()` blocks -- the pipeline wraps its own sampler calls in one -- and the caller's
`torch.cuda.tunable` enabled / tuning flags are restored on exit; TunableOp's results file name points into the temp dir unless the caller had set
one ($PYTORCH_TUNABLEOP_FILENAME).  The entries themselves stay in TunableOp's in-memory table for the process: keys of THIS model's Linear shapes only.  The
results are fp32 GEMMs either way: a different summation order inside the same tolerance as the default kernel's (parity gates unchanged).
"""
from __future__ import annotations

import contextlib
import os
import tempfile
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn.functional as F

# solutions TunableOp's exhaustive search selected on this image's hipBLASLt for the four Linear families at
# M in {6400, 16384, 22976, 32000} (profiles/r3_fp32_gemm_default_vs_tuned.txt), and the library build they are indices of
FOUND_WITH = {"HIPBLASLT_VERSION": "100000-20250912-42-1199-g2584e35062", "GCN_ARCH_NAME": "gfx950:sramecc+:xnack-"}
CANDIDATES = tuple(f"Gemm_Hipblaslt_{i}" for i in (627292, 627296, 627311, 627324, 627325, 627372, 627376, 627391, 627404, 627408, 627436))
OP = "GemmAndBiasTunableOp_float_TN"
_H = 1536                                                                    # weights.DIT_HIDDEN
FAMILIES = ((3 * _H, _H), (_H, _H), (4 * _H, _H), (_H, 4 * _H))              # (N, K) of qkv, proj / out, fc1, fc2

_done: Dict[Tuple[int, Tuple[int, ...]], Dict] = {}
LAST_CUTS: Dict[Tuple[int, int], int] = {}          # (N, K) -> row counts up to this one were left on hipBLASLt's own choice by the last autotune (0: none)


def _key(N: int, M: int, K: int) -> str:
    return f"tn_{N}_{M}_{K}_ld_{K}_{K}_{N}"          # TunableOp's GemmAndBiasParams signature for F.linear(x [M,K], W [N,K], b)


def _write(path: str, entries: Iterable[Tuple[str, str]]) -> None:
    with open(path, "w") as f:
        for name, val in torch.cuda.tunable.get_validators():
            f.write(f"Validator,{name},{val}\n")
        for key, sol in entries:
            f.write(f"{OP},{key},{sol},0.0\n")


def validators_match(found_with: Dict[str, str] = None) -> Tuple[bool, str]:
    """(do the candidates belong to the hipBLASLt build / GPU of this process, the reason if not)"""
    found_with = FOUND_WITH if found_with is None else found_with
    try:
        have = dict(torch.cuda.tunable.get_validators())
    except Exception as e:                                # a PyTorch build without TunableOp
        return False, f"TunableOp unavailable ({type(e).__name__})"
    for k, v in found_with.items():
        if have.get(k) != v:
            return False, f"{k} is {have.get(k)!r}, the candidate kernels were found with {v!r}"
    return True, ""


@contextlib.contextmanager
def enabled():
    """TunableOp on, tuning off, for the duration of the block; the caller's flags are restored afterwards (a drop-in library must not
    leave torch's global switches changed).  Installed kernels take effect inside such a block only."""
    tun = getattr(torch.cuda, "tunable", None)
    if tun is None or not torch.cuda.is_available():
        yield False
        return
    try:
        was_on, was_tuning = tun.is_enabled(), tun.tuning_is_enabled()
        tun.enable(True)
        tun.tuning_enable(False)
    except Exception:
        yield False
        return
    try:
        yield True
    finally:
        tun.tuning_enable(was_tuning)
        tun.enable(was_on)


def _time(fn, n: int = 8, warm: int = 2) -> float:
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def autotune_linears(row_counts: Iterable[int], device: torch.device, reps: Optional[Iterable[int]] = None, families=FAMILIES, candidates=CANDIDATES,
                     verbose: bool = False, found_with: Dict[str, str] = None) -> Optional[Dict]:
    """row_counts: every M = rows of a block-Linear input the step will issue (context stream per step + image stream); reps: the row
    counts the candidates are timed on (default: the median and the largest).  Returns {(N, K): (winner or None, ms default, ms winner)}
    or None when TunableOp is unavailable or the candidates belong to another hipBLASLt build (`found_with`, default FOUND_WITH).
    The installed kernels are used by `F.linear` calls inside an `enabled()` block."""
    rows = sorted({int(m) for m in row_counts if int(m) > 0})
    if not rows or not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return None
    sig = (torch.cuda.current_device(), tuple(rows))
    if sig in _done:
        return _done[sig]
    ok, why = validators_match(found_with)
    if not ok:
        print(f"[gemm_tune] not tuning, hipBLASLt keeps its own kernel choice: {why}", flush=True)
        _done[sig] = None
        return None
    tun = torch.cuda.tunable
    if "PYTORCH_TUNABLEOP_FILENAME" not in os.environ and not (tun.get_filename() or "").startswith(tempfile.gettempdir()):
        try:                                                # torch 2.10 has no write-on-exit switch: whatever TunableOp may write goes
            tun.set_filename(os.path.join(tempfile.gettempdir(), f"selftok_tunableop_{os.getpid()}.csv"))    # to the temp dir, never the caller's cwd
        except Exception:
            pass
    reps = sorted({int(m) for m in reps}) if reps else sorted({rows[-1], rows[len(rows) // 2]})
    report: Dict = {}
    final: List[Tuple[str, str]] = []
    with enabled() as on, tempfile.TemporaryDirectory() as td:
        if not on:
            return None
        probe_id = 0
        for (N, K) in families:
            w = torch.randn(N, K, device=device) * 0.02
            bias = torch.randn(N, device=device)
            # two passes over the candidates (the chip's clock drifts over the first launches of a shape): each candidate keeps its
            # faster pass
            times = {}
            failed = set()
            ncand = len(candidates) + 1
            pool = {M: torch.randn(M + 2 + 2 * ncand, K, device=device) for M in reps}      # one buffer per representative row count; a probe is a row prefix
            for rnd in range(2):
                for ci, cand in enumerate((None,) + tuple(candidates)):
                    if cand in failed:
                        continue
                    total = 0.0
                    for M in reps:
                        Mp = M + 1 + 2 * ci if (M + 1 + 2 * ci) not in rows else M + 2 + 2 * ci      # a row count no real call uses: its key is ours alone
                        a = pool[M][:Mp]
                        try:
                            if cand is not None and rnd == 0:
                                path = os.path.join(td, f"probe{probe_id}.csv")
                                probe_id += 1
                                _write(path, [(_key(N, Mp, K), cand)])
                                tun.read_file(path)
                            total += _time(lambda: F.linear(a, w, bias))
                        except Exception as e:              # unknown solution name / kernel that rejects the shape: drop the candidate
                            failed.add(cand)
                            _write(os.path.join(td, "withdraw.csv"), [(_key(N, Mp, K), "Default")])
                            tun.read_file(os.path.join(td, "withdraw.csv"))
                            if verbose:
                                print(f"[gemm_tune] {cand} dropped for ({N}, {K}): {type(e).__name__}", flush=True)
                            break
                    if cand not in failed:
                        times[cand] = min(times.get(cand, float("inf")), total)
            pool.clear()
            for c in failed:
                times.pop(c, None)
            best = min(times, key=times.get)
            if best is not None and times[best] > 0.99 * times[None]:          # below the timing noise: keep the default
                best = None
            report[(N, K)] = (best, round(times[None], 4), round(times[best], 4))
            cut = 0
            if best is not None and len(rows) > 8:
                # the winner was chosen on the median and the largest row count; a 50-step decode also issues a few SMALL ones (the last steps' context: B (k + 1) rows,
                # k < 64), where a kernel picked for 16384 rows can lose 3x to the library's own small-M choice (round 6: proj at 1280 rows 175 vs 60 us).  Three probes
                # at the low end: rows up to the largest probe the winner loses at keep the default.
                for Mq in sorted({rows[0], rows[len(rows) // 8], rows[len(rows) // 4]}):
                    if Mq >= reps[0]:
                        break
                    free = [m for m in range(Mq + 1, Mq + 9) if m not in rows]           # two row counts next to Mq that no real call uses: their keys are ours alone
                    if len(free) < 2:
                        continue
                    Md, Mb = free[0], free[-1]
                    buf = torch.randn(Mb, K, device=device)
                    try:
                        path = os.path.join(td, f"probe{probe_id}.csv")
                        probe_id += 1
                        _write(path, [(_key(N, Mb, K), best)])
                        tun.read_file(path)
                        t_def = min(_time(lambda: F.linear(buf[:Md], w, bias)) for _ in range(2))
                        t_best = min(_time(lambda: F.linear(buf[:Mb], w, bias)) for _ in range(2))
                        if t_best > t_def:
                            cut = Mq
                    except Exception:                   # a probe must never take the tuning down: leave the small rows on the default
                        cut = max(cut, Mq)
                    del buf
            LAST_CUTS[(N, K)] = cut
            if best is not None:
                final += [(_key(N, M, K), best) for M in rows if M > cut]
            del w, bias
        if final:
            path = os.path.join(td, "selftok_linears.csv")
            _write(path, final)
            tun.read_file(path)
    if verbose:
        for fam, (best, t0, t1) in report.items():
            print(f"[gemm_tune] {fam}: default {t0:.3f} ms -> {best or 'default'} {t1:.3f} ms (two representative row counts)", flush=True)
    _done[sig] = report
    return report


def step_row_counts(B: int, k_table, image_tokens: int) -> Tuple[List[int], List[int]]:
    """(all row counts, the two the candidates are timed on) of the block-Linear inputs of a decode at batch B: context stream
    B (k + 1) rows at each step (timed at the median), image stream B x image_tokens rows at every step"""
    ctx = sorted(B * (int(k) + 1) for k in k_table)
    return sorted(set(ctx) | {B * image_tokens}), [B * image_tokens, ctx[len(ctx) // 2]]
