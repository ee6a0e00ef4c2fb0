"""CPU oracle for the TokensGen denoising hot path -- TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU) restatement of the reference algorithm (Vicky0522/TokensGen, Python).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package `tokensgen_amd` never does.  Every function cites the reference file:line it
follows.  Pinned against the reference itself: `tools/make_golden.py` imports the reference's
modules in the build container (through `tools/ref_shim`), runs them on seeded inputs and commits
the outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks this oracle against them.
"""
