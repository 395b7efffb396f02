"""
largesteps -- MI355X-native implementation of the Laplacian parameterization hot path of
"Large Steps in Inverse Rendering of Geometry" (reference package: rgl-epfl/large-steps-pytorch,
largesteps/ 0.2.2). Same modules, symbols and call forms as the reference:

    from largesteps.geometry import compute_matrix, laplacian_uniform, laplacian_cot
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.solvers import CholeskySolver, ConjugateGradientSolver, solve
    from largesteps.optimize import AdamUniform
    from largesteps.normals import compute_face_normals, compute_vertex_normals     (reference: scripts/geometry.py)

Device work is done by hand-written HIP kernels for gfx950 in lib/liblargesteps_hip.so (C ABI in
include/largesteps_hip.h); there is no CPU or stock-PyTorch fallback.
"""

__version__ = "0.2.2+mi355x.2"
__author__ = "largesteps-mi355x contributors (API after Baptiste Nicolet's largesteps)"
