"""CPU oracle for the deep-prior interpolation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is part of the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the
checker / reported CPU baseline — never as the thing measured as the GPU path or shipped.

Parity status: PINNED.  `oracle/make_golden.py` runs the reference itself (imported from
/root/reference through `oracle/ref_shim.py`, in the build container) and commits its outputs
under `tests/golden/`; `tests/test_oracle_golden.py` checks every oracle function against
those vectors on CPU.
"""
