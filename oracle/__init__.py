"""oracle/ -- CPU restatement of the Selftok hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / the reported CPU baseline.  The product
(selftoktokenizer_amd/) never imports it and has no CPU fallback.

Pinned against the reference itself (imported read-only in the build container by
tools/oracle/gen_golden.py) through the vectors committed under tests/golden/.
The VAE arithmetic lives in a third-party dependency that is absent offline
(diffusers==0.32.2 AutoencoderKL + stabilityai SD3 VAE weights): that part is restated from
the in-repo architectural mirror (mimogpt/models/selftok/sd3/sd3_impls.py:215-474) and pinned
against that mirror only -- "parity unpinned" w.r.t. diffusers.
"""
