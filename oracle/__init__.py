"""
oracle/ -- CPU restatement of the reference's parameterization path. TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (large-steps-pytorch_amd/) imports this directory. The only
allowed users are tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py, and
there only as the checker / the timed CPU baseline, never as the thing shipped.

What is restated (reference = rgl-epfl/large-steps-pytorch @ 0.2.2, paths relative to its root):

  laplacian.uniform_laplacian   largesteps/geometry.py:65-94
  laplacian.cot_laplacian       largesteps/geometry.py:3-63     (fp32, same operation order)
  laplacian.compute_matrix      largesteps/geometry.py:96-133
  solve.to_differential         largesteps/parameterize.py:19-30
  solve.from_differential       largesteps/parameterize.py:32-61 + solvers.py:26-39 ('Cholesky')
  solve.reference_cg            largesteps/solvers.py:58-126     ('CG', fp32, abs 1e-5 stop)
  solve.jacobi_pcg              fp64 statement of the algorithm the HIP PCG implements
  normals.face_normals / vertex_normals (+ *_backward)   scripts/geometry.py:91-110, :115-147 and their analytic gradients
  step.run / step.AdamUniform   the loop body of scripts/main.py:172-208 (no renderer) with largesteps/optimize.py:18-41

Third-party dependency holding the default solver's arithmetic: `cholespy` (requirements.txt:1,
`cholespy>=0.1.4`, unpinned, a nanobind wrapper of SuiteSparse CHOLMOD). It is absent from
/root/reference and not installable here. Its published behaviour -- x = M^-1 b through a sparse
LL^T factorisation computed in double precision -- is restated by an fp64 sparse direct solve
(scipy SuperLU in symmetric mode), which agrees with any correct Cholesky solve to fp64 round-off.

Parity pinning: geometry.py / parameterize.py / solvers.py (CG + autograd) of the reference were
executed in the dev container (CPU tensors; tests/golden/make_golden.py) and their outputs are
committed under tests/golden/. The oracle is checked against every one of them, and against the
hand-checked vectors G1-G7 of SURVEY.md §8c, in tests/test_oracle.py; scripts/geometry.py was executed the same way
(tests/golden/make_golden_normals.py: outputs and torch-autograd gradients) and oracle.normals is checked against it in
tests/test_normals.py. The 'Cholesky' arithmetic
itself (cholespy) could not be executed anywhere: for that single call the parity is pinned only by
the mathematical definition (residual of the fp64 solve), i.e. "parity unpinned" at the cholespy
boundary.
"""
from . import laplacian, normals, solve, step  # noqa: F401
