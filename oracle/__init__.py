"""CPU oracle for the QUiNN hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU in float64, the algorithm of the reference's
data-parallel hot path (log-posterior / gradient, the MH chain stepper and its
samplers, the mean-field ELBO estimator, the `nnfit` trainer used by ensemble
members).  It exists to *check* the HIP path; it is never the thing shipped or
measured.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` may import it.  Nothing under `quinn_amd/` imports it.

Where the arithmetic lives.  The reference is pure Python; every number it
produces comes out of third-party primitives that are NOT under
/root/reference: torch (F.linear/addmm, tanh, autograd, optim.Adam,
distributions.Normal, randperm; de-facto pin torch 2.10.0) and numpy (legacy
RandomState: rand, randn, random_sample, permutation, multivariate_normal ->
LAPACK SVD; de-facto pin numpy 2.2.6) -- `pyproject.toml:32-37` lists bare
names, no versions.  The oracle therefore restates the reference's *control
flow and formulas* and calls those same primitives in the same order, so its
outputs can be (and are) compared bit-for-bit with the reference's.

Parity pinning.  The reference's own tests hold no numerical golden vectors for
this path, so the oracle is pinned against fixtures
generated in the build container by importing the reference itself:
`tests/golden/gen_golden.py` (committed) -> `tests/golden/*.npz`;
`tests/test_oracle_golden.py` asserts equality.  The reference's formula-level
known answers (Gaussian / GMM log-density closed forms, numpar == 21, flatten
round trip, alphas[0] == 0) are re-asserted in `tests/test_oracle_formulas.py`.
"""
