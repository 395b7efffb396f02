"""
CPU-only checks of the boundary: the C-ABI library builds/loads and exports every symbol that
include/largesteps_hip.h declares, the ctypes table matches the header, and the host-side logic that
needs no device (argument validation, error strings, cache keys) behaves like the reference.
No kernel is launched here.
"""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "largesteps_hip.h")


@pytest.fixture(scope="module")
def native():
    from largesteps import _native
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _native


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ls_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(native):
    decl = _declared()
    assert len(decl) >= 12
    out = subprocess.run(["nm", "-D", "--defined-only", native.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in decl if s not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert sorted(native.EXPORTED_SYMBOLS) == decl, "ctypes table and header drifted apart"
    lib = native.lib()
    assert lib.ls_version() == 110


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C with no torch / HIP types."""
    c = tmp_path / "t.c"
    c.write_text('#include "largesteps_hip.h"\nint main(void){ls_solve_info i; i.iterations=0; return i.iterations + (LS_VERSION>0?0:1);}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")],
                   check=True)


def test_host_only_entry_points(native):
    lib = native.lib()
    n = ctypes.c_size_t(0)
    assert lib.ls_assemble_workspace_bytes(1000000, 1996002, ctypes.byref(n)) == 0
    # 5 int arrays of V+1, flags, scan sums, 2 x 6F slots
    # four V-sized int arrays, the corner offsets (3 F), the slot arrays (col, value: 6 F each) and the compact final rows (8 bytes x (6 F + V))
    # + tile tables and padding
    need = 4 * 4 * 1000001 + 4 + 4 * 3 * 1996002 + 2 * 4 * 6 * 1996002 + 8 * (6 * 1996002 + 1000000)
    assert need <= n.value <= need + 128 * 1024
    assert lib.ls_assemble_workspace_bytes(-1, 0, ctypes.byref(n)) == native.LS_E_INVALID
    assert lib.ls_assemble_workspace_bytes(3 * 10 ** 9, 0, ctypes.byref(n)) == native.LS_E_OVERFLOW
    assert "too large" in native.last_error()
    assert lib.ls_solver_set(None, b"variant", 1) == native.LS_E_INVALID
    with pytest.raises(ValueError):
        native.check(native.LS_E_INVALID)
    with pytest.raises(IndexError):
        native.check(native.LS_E_INDEX)
    assert lib.ls_solver_destroy(None) == 0


def test_reference_error_strings_without_device(golden):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    from largesteps.solvers import Solver
    e = golden.errors()
    v = torch.from_numpy(golden["quad/verts"])
    f = torch.from_numpy(golden["quad/faces"])
    for a in (1.0, -0.1, 1.5):
        with pytest.raises(ValueError) as ei:
            compute_matrix(v, f, 1.0, alpha=a)
        assert str(ei.value) == e[f"alpha={a}"]
    with pytest.raises(RuntimeError, match="no CPU path"):
        compute_matrix(v, f, 1.0)
    M = torch.sparse_coo_tensor(torch.tensor([[0, 1], [0, 1]]), torch.ones(2), (2, 2)).coalesce()
    with pytest.raises(ValueError) as ei:
        from_differential(M, v, "LU")
    assert str(ei.value) == e["method"]
    with pytest.raises(RuntimeError, match="no CPU path"):
        from_differential(M, torch.ones(2, 3), "Cholesky")
    with pytest.raises(NotImplementedError):
        Solver(M).solve(v)


def test_api_surface_matches_reference():
    import inspect
    import largesteps
    from largesteps import geometry, parameterize, solvers, optimize
    assert largesteps.__version__.startswith("0.2.2")
    assert list(inspect.signature(geometry.compute_matrix).parameters) == ["verts", "faces", "lambda_", "alpha", "cotan"]
    assert inspect.signature(geometry.compute_matrix).parameters["alpha"].default is None
    assert inspect.signature(geometry.compute_matrix).parameters["cotan"].default is False
    assert list(inspect.signature(geometry.laplacian_uniform).parameters) == ["verts", "faces"]
    assert list(inspect.signature(geometry.laplacian_cot).parameters) == ["verts", "faces"]
    assert list(inspect.signature(parameterize.to_differential).parameters) == ["L", "v"]
    sig = inspect.signature(parameterize.from_differential)
    assert list(sig.parameters) == ["L", "u", "method"] and sig.parameters["method"].default == "Cholesky"
    for name in ("Solver", "CholeskySolver", "ConjugateGradientSolver", "DifferentiableSolve", "solve"):
        assert hasattr(solvers, name)
    assert list(inspect.signature(solvers.Solver.solve).parameters) == ["self", "b", "backward"]
    sig = inspect.signature(optimize.AdamUniform.__init__)
    assert sig.parameters["lr"].default == 0.1 and sig.parameters["betas"].default == (0.9, 0.999)
    assert hasattr(parameterize, "_cache") and hasattr(parameterize, "cache_put")


def test_product_does_not_import_oracle():
    """The shipped package must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "large-steps-pytorch_amd")
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{fn} imports the oracle"
                assert "scipy" not in src, f"{fn} uses scipy"


def test_synthetic_meshes():
    from largesteps import synthetic
    v, f = synthetic.plane(1000)
    assert v.shape == (1000000, 3) and f.shape == (1996002, 3) and v.dtype == np.float32 and f.dtype == np.int64
    assert np.array_equal(f[0], [0, 1, 1001]) and np.array_equal(f[1], [0, 1001, 1000])
    for n in (1, 2, 16):
        v, f = synthetic.icosphere(n)
        assert v.shape[0] == 10 * n * n + 2 and f.shape[0] == 20 * n * n
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        _, cnt = np.unique(e, axis=0, return_counts=True)
        assert (cnt == 2).all() and cnt.shape[0] == 30 * n * n
    a = synthetic.config_mesh("cfg2_bunny70k")
    b = synthetic.config_mesh("cfg2_bunny70k")
    assert a[0].shape[0] == 70562 and np.array_equal(a[0], b[0])


def test_adam_uniform_capturable_state_dict_round_trip_keeps_an_int32_counter():
    """torch's load_state_dict casts a tensor "step" of a capturable group to float32; the kernel reads int32 (advisor finding,
    round 2): the optimizer converts it back by value."""
    from largesteps.optimize import AdamUniform
    p = torch.zeros(6, 3, requires_grad=True)
    opt = AdamUniform([p], lr=0.05, capturable=True)
    opt.state[p] = dict(step=torch.tensor([100, 0], dtype=torch.int32), g1=torch.ones_like(p), g2=torch.ones_like(p))
    sd = opt.state_dict()
    q = torch.zeros(6, 3, requires_grad=True)
    opt2 = AdamUniform([q], lr=0.05, capturable=True)
    opt2.load_state_dict(sd)
    step = opt2.state[q]["step"]
    assert step.dtype == torch.int32 and step.tolist() == [100, 0]
    assert opt2.state[q]["g1"].dtype == torch.float32
    import copy
    opt3 = copy.deepcopy(opt2)                       # __setstate__ path
    assert next(iter(opt3.state.values()))["step"].dtype == torch.int32


def test_picked_tree_by_system_size():
    """ls_direct_pick_tree (host only): the tree ls_direct_factor uses when leaf size / arity are left to the library -- one dense node
    for very small systems, shallow arity-4 trees up to 12k unknowns, arity 8 between 12k and 300k (three levels / four levels with dense
    leaves / five levels with sparse leaves), the arity-4 tree with 64-vertex leaves beyond; explicit values are kept."""
    import ctypes
    from largesteps import _native
    lib = _native.lib()

    def pick(V, leaf=0, arity=0):
        l, a = ctypes.c_int(leaf), ctypes.c_int(arity)
        _native.check(lib.ls_direct_pick_tree(V, ctypes.byref(l), ctypes.byref(a)))
        return l.value, a.value

    assert pick(4) == (4, 4) and pick(1280) == (1280, 4)
    assert pick(1281) == (1024, 4) and pick(12000) == (1024, 4)
    assert pick(12001) == (1024, 8) and pick(36000) == (1024, 8)
    assert pick(36001) == (256, 8) and pick(70562) == (256, 8) and pick(105000) == (256, 8)
    assert pick(105001) == (64, 8) and pick(249642) == (64, 8)
    assert pick(300000) == (128, 8)                       # 64-vertex leaves would round down to 9 vertices: one level less
    assert pick(300001) == (64, 4) and pick(1000000) == (64, 4) and pick(4000000) == (64, 4)
    assert pick(70000, leaf=64) == (64, 8) and pick(1000000, leaf=32, arity=2) == (32, 2)
    assert pick(70225, arity=4) == (128, 4) and pick(40000, arity=4) == (64, 4)      # the arity-4 rules for callers that fix the arity
    with pytest.raises(Exception):
        _native.check(lib.ls_direct_pick_tree(0, ctypes.byref(ctypes.c_int(0)), ctypes.byref(ctypes.c_int(0))))


def test_no_kernel_of_the_product_library_spills():
    """Every kernel of the built library, read from its code objects' metadata (tools/kernel_resources.py: .hip_fatbin unbundled,
    `llvm-readelf --notes`): no VGPR spill, no scratch. Round 3 shipped k_nd_down<1..4> with 42-114 spilled registers (a prefetch depth
    tuned on another kernel) and k_patch_cheb<3, 1024, 8> with 16 (hoisted unpacking of its packed neighbour ids) -- unnoticed, because
    nothing looked. An entry in ALLOWED needs a measured reason next to it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.kernels()
    assert len(ks) > 200 and any("k_nd_tier<3, true, 4, false>" in n for n in ks) and any("k_nd_tier<3, false, 4, true>" in n for n in ks) and any("k_nd_down<3, false>" in n for n in ks)
    ALLOWED = {}          # kernel name -> why its spills are accepted (none)
    bad = {n: (k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0)) for n, k in ks.items()
           if (k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)) and n not in ALLOWED}
    assert not bad, f"kernels with (spilled VGPRs, scratch bytes): {bad}"
    # the experiment of round 3 (one persistent launch for the upper levels) is archived source, not a kernel of the library
    assert not any("k_nd_span" in n for n in ks)


def test_host_planners_refuse_malformed_patterns():
    """The native planners index arrays with the caller's column ids: a malformed CSR pattern or a NaN position must come back as
    LS_E_INVALID with a message, not as an out-of-bounds write (the numpy statements they replaced raised IndexError)."""
    import ctypes
    import numpy as np
    from largesteps import _native
    lib = _native.lib()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    rowptr = np.array([0, 2, 4, 6], np.int32)
    good = np.array([0, 1, 0, 1, 1, 2], np.int32)
    pos = np.zeros((3, 3), np.float32)
    diag = np.ones(3, np.float32)
    val = np.ones(6, np.float32)
    cases = [(rowptr, np.array([0, 1, 0, 7, 1, 2], np.int32), pos, "column index"),
             (rowptr, np.array([0, -1, 0, 1, 1, 2], np.int32), pos, "column index"),
             (np.array([0, 4, 2, 6], np.int32), good, pos, "monotone"),
             (np.array([1, 2, 4, 6], np.int32), good, pos, "rowptr[0]")]
    bad_pos = pos.copy()
    bad_pos[1, 2] = np.nan
    for rp, col, ps, what in cases + [(rowptr, good, bad_pos, "finite")]:
        h = ctypes.c_void_p()
        assert lib.ls_nd_plan_create(3, p(rp), p(col), p(ps), 2, 2, 0, ctypes.byref(h)) == _native.LS_E_INVALID and what in _native.last_error()
        assert lib.ls_patch_plan_create(3, p(rp), p(col), p(diag), p(ps), 2, 2, 100, 1, 100, ctypes.byref(h)) == _native.LS_E_INVALID
        assert what in _native.last_error()
    for rp, col, ps, what in cases:
        h = ctypes.c_void_p()
        assert lib.ls_shard_plan_create(3, p(rp), p(col), p(val), 1, 0, 1, ctypes.byref(h)) == _native.LS_E_INVALID and what in _native.last_error()
        sizes = np.zeros(1, np.int64)
        assert lib.ls_shard_layer_sizes(3, p(rp), p(col), 0, 1, 1, p(sizes)) == _native.LS_E_INVALID and what in _native.last_error()
    h = ctypes.c_void_p()
    assert lib.ls_nd_plan_create(3, p(rowptr), p(good), p(pos), 2, 2, 0, ctypes.byref(h)) == 0
    lib.ls_nd_plan_destroy(h)


def test_sorting_network_of_the_assembler_sorts():
    """csrc/assemble.hip sorts a row's (column, weight) slots in registers with a fixed comparator network; by the 0-1 principle it sorts
    every input iff it sorts all 2^16 sequences of zeros and ones (tools/check_sort16.py reads the comparators out of the source)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sort16.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "60 comparators" in r.stdout and "True" in r.stdout, r.stdout + r.stderr


def test_out_of_memory_in_torch_empties_the_pool_and_retries_once(native):
    """The direct solver's buffer pool is invisible to torch's caching allocator: the package's allocating entry points release it
    (and torch's cache) and repeat the call ONCE when torch reports out of memory (advisor's finding, round 4)."""
    import torch
    calls = []

    @native.retry_on_oom
    def flaky(x):
        calls.append(x)
        if len(calls) == 1:
            raise torch.cuda.OutOfMemoryError("simulated")
        return x + 1
    assert flaky(1) == 2 and calls == [1, 1]

    @native.retry_on_oom
    def hopeless():
        calls.append("h")
        raise torch.cuda.OutOfMemoryError("simulated")
    with pytest.raises(torch.cuda.OutOfMemoryError):
        hopeless()
    assert calls.count("h") == 2
    # decorated calls nest (CholeskySolver.solve -> NestedDissectionSolver.solve): one failure = ONE repetition, by the outermost call
    # (advisor's finding, round 5: up to 4 attempts before)
    inner_calls, outer_calls = [], []

    @native.retry_on_oom
    def inner():
        inner_calls.append(1)
        raise torch.cuda.OutOfMemoryError("simulated")

    @native.retry_on_oom
    def outer():
        outer_calls.append(1)
        return inner()
    with pytest.raises(torch.cuda.OutOfMemoryError):
        outer()
    assert len(outer_calls) == 2 and len(inner_calls) == 2
    assert flaky(1) == 2                                     # the guard is released after a failure
    from largesteps import geometry, normals, solvers
    for fn in (geometry.compute_matrix, normals.compute_vertex_normals, solvers.NestedDissectionSolver.solve, solvers.ConjugateGradientSolver.solve):
        assert hasattr(fn, "__wrapped__"), fn


def test_direct_options_struct_and_argument_checks_without_a_device(native):
    """ls_direct_options (round 6): the defaults, the size field, and the argument checks that come before any device call -- the constructor's
    choices are arguments of the C ABI, not environment variables."""
    import ctypes
    lib = native.lib()
    opt = native.DirectOptions()
    native.check(lib.ls_direct_options_default(ctypes.byref(opt)))
    assert ctypes.sizeof(native.DirectOptions) == 36 and opt.struct_bytes == 36
    assert (opt.leaf_size, opt.arity, opt.tier_levels, opt.sparse_leaves, opt.shard_rank, opt.shard_count, opt.ordering, opt.tier_waves) == (0, 0, -1, 1, 0, 1, -1, 0)
    with pytest.raises(ValueError, match="null"):
        native.check(lib.ls_direct_options_default(None))
    h = ctypes.c_void_p()
    dummy = ctypes.c_void_p(16)          # never dereferenced: the checks below fail first
    for field, value, what in (("ordering", 7, "ordering"), ("tier_waves", 5, "tier_waves"), ("struct_bytes", 0, "struct_bytes")):
        bad = native.DirectOptions()
        native.check(lib.ls_direct_options_default(ctypes.byref(bad)))
        setattr(bad, field, value)
        with pytest.raises(ValueError, match=what):
            native.check(lib.ls_direct_factor_ex(dummy, dummy, dummy, 10, 10, None, ctypes.byref(bad), 0, None, ctypes.byref(h)))
    with pytest.raises(ValueError, match="bad argument"):           # null matrix: refused before the device is touched
        native.check(lib.ls_direct_factor_ex(None, None, None, 10, 10, None, None, 0, None, ctypes.byref(h)))
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "largesteps_hip.h")).read()
    assert "typedef struct ls_direct_options" in hdr and "int ls_direct_factor_ex(" in hdr and "#define LS_VERSION 110" in hdr
