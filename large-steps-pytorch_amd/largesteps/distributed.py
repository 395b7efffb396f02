"""
Vertex-block sharded from_differential: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the MI355X node; "gloo" in the CPU tests).

The reference is strictly single process / single GPU (no distributed call site exists), so
this module has no reference counterpart; it extends the solve of largesteps/solvers.py to the partitioning
BASELINE.json's north star names: the mesh is cut into P contiguous vertex blocks, rank r owns the rows of
block r of M and the matching rows of u / x. Per PCG iteration the ranks exchange

    * the halo rows of the search direction p (neighbour-only isend/irecv; 1000 vertices x 12 B per
      neighbour for the 1M-vertex plane), and
    * two fused all-reduces of the dot-product partials (p.Ap ; r.z and r.r for all columns at once).

Everything else is the single-GPU kernels of csrc/pcg.hip, launched one at a time on the shard's rectangular
matrix (owned rows x [owned | halo] columns) through the C ABI (ls_solver_create_ext / ls_solver_phase).

Layers (so that the host logic is testable without a GPU):
    ShardPlan      native analysis (csrc/shard_plan.cpp): row block, local column ids, halo / send lists.
    ShardedPCG     the iteration driver: collectives + a LocalOps object.           (CPU tests with gloo)
    HipShardOps    LocalOps on the HIP kernels -- the only implementation shipped.  (GPU tests)
"""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _native

_KMAX = 4
_PART_SLOTS = 4


def block_bounds(V, P):
    """Contiguous vertex blocks: rank r owns [bounds[r], bounds[r+1])."""
    return np.array([(r * V) // P for r in range(P + 1)], dtype=np.int64)


class ShardPlan:
    """Everything rank `rank` of `P` needs to know about its block of a V x V CSR matrix. The analysis is native
    (csrc/shard_plan.cpp behind ls_shard_plan_*: breadth-first ghost layers, local column ids, receive / send lists); this class
    only owns the arrays. numpy statement of the same plan: tests/shard_plan_statement.py.

    depth = 1 (PCG): the shard computes its owned rows; columns are [owned | halo layer 1].
    depth = s > 1 (Chebyshev, one halo exchange per s iterations): the shard ALSO computes the ghost layers
    1..s-1 redundantly (layer j = vertices at graph distance j from the block) and reads layer s, so that s
    iterations can run between two exchanges -- after j iterations the layers > s-j are stale, the owned rows never are.

    Local ids: [owned (n_own) | ghost layers 1..s-1 (computed) | ghost layer s (read only)]; the first `n_rows`
    local ids are the rows of the local matrix. Ghosts are ordered by (owner rank, global id) inside each of the
    two ghost groups, so every (owner, group) pair is one contiguous range:
    recv : [(src_rank, offset_in_ghost_region, count)]   in local order
    send : [(dst_rank, local_row_ids int32)]             the matching owned rows, message for message
    """

    def __init__(self, rank, P, lo, hi, depth, rowptr, col, val, ghost_global, n_ghost_rows, recv, send):
        self.rank, self.P, self.lo, self.hi, self.depth = rank, P, int(lo), int(hi), int(depth)
        self.n_own = int(hi - lo)
        self.n_halo = int(ghost_global.shape[0])
        self.n_rows = self.n_own + int(n_ghost_rows)          # rows of the local matrix (owned + computed ghosts)
        self.n_cols = self.n_own + self.n_halo
        self.rowptr, self.col, self.val = rowptr, col, val
        self.halo_global, self.recv, self.send = ghost_global, recv, send

    @staticmethod
    def build(rowptr, col, val, V, P, rank, depth=1):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        lib = _native.lib()
        as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        h = ctypes.c_void_p(None)
        _native.check(lib.ls_shard_plan_create(int(V), as_p(rowptr), as_p(col), as_p(val), int(P), int(rank), int(depth), ctypes.byref(h)))
        try:
            lo, hi, n_inner, n_ghosts, n_ent, n_ids = (ctypes.c_int64(0) for _ in range(6))
            n_recv, n_send = ctypes.c_int(0), ctypes.c_int(0)
            _native.check(lib.ls_shard_plan_info(h, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(n_inner), ctypes.byref(n_ghosts),
                                                 ctypes.byref(n_ent), ctypes.byref(n_recv), ctypes.byref(n_send), ctypes.byref(n_ids)))
            n_rows = hi.value - lo.value + n_inner.value
            l_rowptr = np.empty(n_rows + 1, dtype=np.int32)
            l_col = np.empty(n_ent.value, dtype=np.int32)
            l_val = np.empty(n_ent.value, dtype=np.float32)
            ghosts = np.empty(n_ghosts.value, dtype=np.int32)
            recv3 = np.empty((n_recv.value, 3), dtype=np.int32)
            s_ptr = np.empty(n_send.value + 1, dtype=np.int32)
            s_dst = np.empty(n_send.value, dtype=np.int32)
            s_ids = np.empty(n_ids.value, dtype=np.int32)
            _native.check(lib.ls_shard_plan_arrays(h, as_p(l_rowptr), as_p(l_col), as_p(l_val), as_p(ghosts), as_p(recv3), as_p(s_ptr),
                                                   as_p(s_dst), as_p(s_ids)))
        finally:
            lib.ls_shard_plan_destroy(h)
        recv = [(int(q), int(o), int(c)) for q, o, c in recv3]
        send = [(int(s_dst[i]), s_ids[s_ptr[i]:s_ptr[i + 1]].copy()) for i in range(n_send.value)]
        return ShardPlan(rank, P, lo.value, hi.value, depth, l_rowptr, l_col, l_val, ghosts.astype(np.int64), n_inner.value, recv, send)


class HipShardOps:
    """LocalOps on the MI355X: the kernels of csrc/pcg.hip on this rank's shard (through the C ABI)."""

    def __init__(self, plan, device, grid=None, block=None):
        self.plan, self.device = plan, torch.device(device)
        _native.require_device(torch.empty(0, device=self.device), "the shard device")
        dev = self.device
        self.rowptr = torch.from_numpy(plan.rowptr).to(dev)
        self.col = torch.from_numpy(plan.col).to(dev)
        self.val = torch.from_numpy(plan.val).to(dev)
        self._handle = ctypes.c_void_p(None)
        lib = _native.lib()
        with torch.cuda.device(dev):
            _native.check(lib.ls_solver_create_ext(_native.ptr(self.rowptr), _native.ptr(self.col), _native.ptr(self.val),
                                                   plan.n_rows, plan.n_cols, plan.col.shape[0], _KMAX, dev.index,
                                                   _native.stream_of(dev), ctypes.byref(self._handle)))
        if block is not None:
            _native.check(lib.ls_solver_set(self._handle, b"block", int(block)))
        if grid is not None:
            _native.check(lib.ls_solver_set(self._handle, b"grid", int(grid)))
        p_ptr, part_ptr, g, stride = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        _native.check(lib.ls_solver_buffers(self._handle, ctypes.byref(p_ptr), ctypes.byref(part_ptr), ctypes.byref(g),
                                            ctypes.byref(stride)))
        self.grid, self.part_stride = g.value, stride.value
        # torch owns the two buffers the collectives touch; the handle is re-pointed at them
        self._p_flat = torch.zeros(max(plan.n_cols, 1) * _KMAX, dtype=torch.float32, device=dev)
        self.part = torch.zeros((_PART_SLOTS, _KMAX, self.part_stride), dtype=torch.float64, device=dev)
        _native.check(lib.ls_solver_bind(self._handle, _native.ptr(self._p_flat), _native.ptr(self.part)))
        self.send_idx = [(q, torch.from_numpy(idx).to(dev)) for q, idx in plan.send]
        self.k = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _native.lib().ls_solver_destroy(h)
            except Exception:      # interpreter shutdown
                pass
            self._handle = None

    # -- LocalOps protocol ---------------------------------------------------------------------------
    def new_vector(self, k):
        return torch.empty((self.plan.n_own, k), dtype=torch.float32, device=self.device)

    def p_ext(self, k):
        """(n_cols, k) view of the search direction: rows [0, n_own) owned, the rest halo."""
        return self._p_flat[: self.plan.n_cols * k].view(self.plan.n_cols, k)

    def phase(self, phase, b, x, k, rtol, atol, it):
        _native.check(_native.lib().ls_solver_phase(self._handle, phase, _native.ptr(b), _native.ptr(x), k, rtol, atol, it,
                                                    _native.stream_of(self.device)))

    def pack(self, src, idx, k, out):
        """out[t,:] = src[idx[t],:] (send-buffer packing)."""
        _native.check(_native.lib().ls_gather_rows(_native.ptr(src), _native.ptr(idx), idx.shape[0], k,
                                                   _native.ptr(out), self.device.index, _native.stream_of(self.device)))

    def new_ext(self, k):
        """zero (n_cols, k) vector: owned rows, then the ghost layers."""
        return torch.zeros((self.plan.n_cols, k), dtype=torch.float32, device=self.device)

    def local_spectrum(self):
        """(Gershgorin bound of spec(D^-1 M), max diagonal) over this shard's rows."""
        lo, hi = ctypes.c_double(), ctypes.c_double()
        lib = _native.lib()
        _native.check(lib.ls_solver_set_spectrum(self._handle, 1.0))
        _native.check(lib.ls_solver_spectrum(self._handle, ctypes.byref(lo), ctypes.byref(hi)))
        return hi.value, (1.0 / lo.value if lo.value > 0 else 0.0)

    def cheb_steps(self, b, xa, xb, k, it0, c1, c2, n_rows):
        n = len(c1)
        a1 = (ctypes.c_float * n)(*c1)
        a2 = (ctypes.c_float * n)(*c2)
        _native.check(_native.lib().ls_shard_cheb_steps(self._handle, _native.ptr(b), _native.ptr(xa), _native.ptr(xb), k, it0, n,
                                                        a1, a2, n_rows, _native.stream_of(self.device)))

    def resnorm(self, b, x, k, n_rows):
        _native.check(_native.lib().ls_shard_resnorm(self._handle, _native.ptr(b), _native.ptr(x), k, n_rows,
                                                     _native.stream_of(self.device)))

    def poll(self, k, n):
        info = _native.SolveInfo()
        rc = _native.lib().ls_solver_poll(self._handle, k, n, ctypes.byref(info), _native.stream_of(self.device))
        if rc not in (0, _native.LS_E_NOT_CONVERGED):
            _native.check(rc)
        return dict(iterations=info.iterations, converged=bool(info.converged), rnorm=list(info.rnorm)[:k],
                    bnorm=list(info.bnorm)[:k], breakdown=rc != 0)


class ShardedPCG:
    """Iteration driver of the vertex-block sharded Jacobi-PCG. `ops` is this rank's LocalOps (HipShardOps in
    the product; the CPU tests inject a numpy statement of the same kernels to exercise the collectives on
    gloo). All ranks of `group` must call solve() together."""

    def __init__(self, plan, ops, group=None, rtol=1e-6, atol=0.0, max_iter=10000, check_every=16):
        self.plan, self.ops, self.group = plan, ops, group
        self.rtol, self.atol, self.max_iter, self.check_every = float(rtol), float(atol), int(max_iter), int(check_every)
        self.last_info = None
        self._sendbuf = {}

    def _allreduce(self, slot0, nslots):
        if self.plan.P > 1:
            dist.all_reduce(self.ops.part[slot0:slot0 + nslots], op=dist.ReduceOp.SUM, group=self.group)

    def _exchange_halo(self, k):
        self._exchange(self.ops.p_ext(k))

    def _exchange(self, *fulls):
        """Refresh the ghost rows of every `full` ((n_cols, k): owned rows first) from their owners, all tensors in
        ONE batch of neighbour-only isend/irecv (one RCCL group launch)."""
        plan, ops = self.plan, self.ops
        if plan.P == 1 or (not plan.recv and not plan.send):
            return
        # Loopback transport for single-GPU test boxes: gloo cannot isend/irecv device tensors,
        # so with backend gloo + device tensors the halo rows are staged through host memory. RCCL ("nccl")
        # moves the device buffers directly.
        stage = fulls[0].is_cuda and dist.get_backend(self.group) == "gloo"
        reqs, landed = [], []
        for t, full in enumerate(fulls):
            k = full.shape[1]
            for n, (q, idx) in enumerate(ops.send_idx):
                buf = self._sendbuf.get((t, n, k))
                if buf is None:
                    buf = self._sendbuf[(t, n, k)] = torch.empty((idx.shape[0], k), dtype=torch.float32, device=full.device)
                ops.pack(full, idx, k, buf)
                reqs.append(dist.P2POp(dist.isend, buf.cpu() if stage else buf, self._peer(q), group=self.group))
            for q, off, cnt in plan.recv:
                dst = full[plan.n_own + off: plan.n_own + off + cnt]
                if stage:
                    host = torch.empty(dst.shape, dtype=dst.dtype)
                    landed.append((dst, host))
                    dst = host
                reqs.append(dist.P2POp(dist.irecv, dst, self._peer(q), group=self.group))
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
        for dst, host in landed:
            dst.copy_(host)

    def _peer(self, q):
        return q if self.group is None else dist.get_global_rank(self.group, q)

    def solve(self, b):
        """b: this rank's (n_own, k) block of the right-hand side. Returns this rank's block of x."""
        plan, ops = self.plan, self.ops
        if b.dim() != 2 or b.shape[0] != plan.n_own or not (1 <= b.shape[1] <= _KMAX):
            raise ValueError(f"expected a ({plan.n_own}, k<=4) block of the right-hand side, got {tuple(b.shape)}")
        b = b.detach().contiguous()
        k = b.shape[1]
        x = ops.new_vector(k)
        ops.phase(0, b, x, k, self.rtol, self.atol, 0)          # r = b, p = D^-1 r, x = 0 ; partials r.z, r.r, b.b
        self._allreduce(1, 3)
        ops.phase(1, b, x, k, self.rtol, self.atol, 0)          # thresholds, column mask (identical on every rank)
        self._exchange_halo(k)
        n, info = 0, None
        while n < self.max_iter:
            ops.phase(2, b, x, k, self.rtol, self.atol, n)      # K1: Ap = M p_ext ; partial p.Ap
            self._allreduce(0, 1)
            ops.phase(3, b, x, k, self.rtol, self.atol, n)      # K2: x, r update ; partials r.z, r.r
            self._allreduce(1, 2)
            ops.phase(4, b, x, k, self.rtol, self.atol, n)      # K3: new p, stop flag
            self._exchange_halo(k)
            n += 1
            if n % self.check_every == 0 or n == self.max_iter:
                info = ops.poll(k, n)                             # same answer on every rank
                if info["iterations"] >= 0 or info["breakdown"]:
                    break
        if info is None:
            info = ops.poll(k, n)
        if info["iterations"] < 0:
            info["iterations"] = n
        self.last_info = info
        if info["breakdown"]:
            raise RuntimeError("largesteps: sharded PCG broke down (non-finite residual or matrix not SPD)")
        return x


class ShardedChebyshev(ShardedPCG):
    """Vertex-block sharded Chebyshev-accelerated Jacobi iteration (the single-GPU default of csrc/pcg.hip, k_cheb).

    No dot products => no all-reduce inside the iteration; the only exchange is the ghost rows of the two
    iterates, and with a depth-s ShardPlan (ghost layers 1..s-1 recomputed redundantly) only once per s
    iterations: n/s neighbour exchanges per solve instead of 2n all-reduces + n halo exchanges for PCG. The
    iteration count n and the coefficients follow from the global spectral enclosure (one MAX all-reduce at
    construction), so every rank runs the same schedule without talking.
    """

    def __init__(self, plan, ops, a_min, group=None, rtol=1e-6, max_iter=10000):
        super().__init__(plan, ops, group=group, rtol=rtol, atol=0.0, max_iter=max_iter)
        gersh, dmax = ops.local_spectrum()
        t = torch.tensor([gersh, dmax], dtype=torch.float64)
        if plan.P > 1:
            dev_t = t.to(ops.new_ext(1).device) if dist.get_backend(group) == "nccl" else t
            dist.all_reduce(dev_t, op=dist.ReduceOp.MAX, group=group)
            t = dev_t.cpu()
        self.dmax = float(t[1])
        self.lmax = float(t[0]) * (1.0 + 1e-5)
        self.lmin = 0.98 * float(a_min) / self.dmax

    def schedule(self, reduction):
        """(n, c1[], c2[]) of the Chebyshev recurrence for a residual reduction `reduction` (same on every rank)."""
        import math
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        sigma1 = theta / delta
        sk = math.sqrt(self.lmax / self.lmin)
        rate = (sk - 1.0) / (sk + 1.0)
        n = 0 if reduction >= 1.0 else int(math.ceil(math.log(2.0 / reduction) / -math.log(rate)))
        n = min(n, self.max_iter)
        c1, c2, rho = [], [], 1.0 / sigma1
        for it in range(n):
            if it == 0:
                c1.append(0.0)
                c2.append(1.0 / theta)
            else:
                rho_new = 1.0 / (2.0 * sigma1 - rho)
                c1.append(rho_new * rho)
                c2.append(2.0 * rho_new / delta)
                rho = rho_new
        return n, c1, c2

    def solve(self, b):
        plan, ops = self.plan, self.ops
        if b.dim() != 2 or b.shape[0] != plan.n_own or not (1 <= b.shape[1] <= _KMAX):
            raise ValueError(f"expected a ({plan.n_own}, k<=4) block of the right-hand side, got {tuple(b.shape)}")
        k = b.shape[1]
        n, c1, c2 = self.schedule(self.rtol)                   # cold start: ||r0|| = ||b||
        b_ext = ops.new_ext(k)
        b_ext[: plan.n_own] = b.detach()
        self._exchange(b_ext)                                   # the computed ghost rows need their b
        xa, xb = ops.new_ext(k), ops.new_ext(k)
        s = plan.depth
        for it0 in range(0, n, s):
            if it0 > 0:                                         # all ghost layers of both iterates are refreshed
                self._exchange(xa, xb)
            steps = min(s, n - it0)
            ops.cheb_steps(b_ext, xa, xb, k, it0, c1[it0:it0 + steps], c2[it0:it0 + steps], plan.n_rows)
        x_ext = xa if n % 2 == 0 else xb
        # true residual of the result over the owned rows (needs ghost layer 1 of x), summed over ranks; accepted when
        # it is at the requested level or at the fp32 backward-stable level 8 eps ||M|| ||x|| (as in csrc/pcg.hip)
        self._exchange(x_ext)
        ops.resnorm(b_ext, x_ext, k, plan.n_own)
        self._allreduce(1, 3)
        sums = ops.part[1:4, :k].sum(dim=2).cpu().numpy()      # rows: ||r||^2, ||x||^2, ||b||^2 per column
        rr, xx, bb = sums[0], sums[1], sums[2]
        mnorm = self.lmax * self.dmax
        lim = np.maximum(self.rtol ** 2 * bb, 64.0 * 3.6e-15 * mnorm * mnorm * xx)
        ok = bool(np.all(rr == rr) and np.all(rr <= lim))
        self.last_info = dict(iterations=n, converged=ok, rnorm=list(np.sqrt(rr)), bnorm=list(np.sqrt(bb)), breakdown=False,
                              method="chebyshev", exchanges=max(0, (n - 1) // s) + 2)
        if not ok:
            raise RuntimeError("largesteps: sharded Chebyshev failed its residual check (spectral enclosure violated?)")
        return x_ext[: plan.n_own].clone()


class ColumnSharded:
    """Right-hand-side sharding: the k columns of u are independent systems that share M, so rank r solves the columns
    {c : c mod min(P, k) == r} on its own GPU with the single-GPU solver and the ranks only meet in ONE all-gather of
    the solution columns -- no halo exchange, no all-reduce, nothing per iteration.

    Why it exists next to the vertex-block shards: one MI355X solves the 1M-vertex system in ~1.3 ms (23 kernel
    launches); cutting that across GPUs by vertex blocks adds several neighbour exchanges of tens of microseconds each
    (plus their host-side launch cost) to every solve and cannot win at this size, whereas a one-column solve moves a
    third of the vector bytes and needs no communication at all. Vertex-block sharding (ShardedChebyshev / ShardedPCG)
    remains the mode for meshes that do not fit one GPU.

    `solve_columns(b_cols) -> x_cols` is the local solver ((V, k_r) -> (V, k_r)); the product passes a PCGSolver's
    solve, the CPU tests a numpy one.
    """

    def __init__(self, solve_columns, k, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.k = int(k)
        self.active = min(self.P, self.k)
        self.columns = [c for c in range(self.k) if self.rank < self.active and c % self.active == self.rank]
        self.max_cols = -(-self.k // self.active)
        self.solve_columns = solve_columns

    def solve(self, b):
        """b: the full (V, k) right-hand side, resident on every rank. Returns the full (V, k) solution on every rank."""
        if b.dim() != 2 or b.shape[1] != self.k:
            raise ValueError(f"expected a (V, {self.k}) right-hand side, got {tuple(b.shape)}")
        V = b.shape[0]
        key = (V, b.dtype, b.device)
        if getattr(self, "_key", None) != key:        # staging buffers, reused across solves (padding rows stay zero)
            self._key = key
            self._mine = torch.zeros((self.max_cols, V), dtype=b.dtype, device=b.device)
            self._flat = torch.empty((self.P * self.max_cols, V), dtype=b.dtype, device=b.device)
            self._rows = torch.tensor([(c % self.active) * self.max_cols + c // self.active for c in range(self.k)],
                                      dtype=torch.int64, device=b.device)
        mine = self._mine
        if self.columns:
            x = self.solve_columns(b[:, self.columns].contiguous())
            mine[: len(self.columns)].copy_(x.t())
        if self.P == 1:
            return mine[: self.k].t().contiguous()
        # every rank contributes a (max_cols, V) block (idle / short ranks: zero rows): one all-gather
        if mine.is_cuda and dist.get_backend(self.group) == "gloo":      # loopback smoke mode: stage through the host
            host = torch.empty((self.P * self.max_cols, V), dtype=b.dtype)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            flat = host.to(b.device)
        else:
            flat = self._flat
            dist.all_gather_into_tensor(flat, mine, group=self.group)
        return flat.index_select(0, self._rows).t().contiguous()


class ShardedDirect:
    """The nested-dissection direct solver sharded by vertex blocks = SUBTREES of its elimination tree, one process per GPU.

    The tree's first level with at least P nodes is the cut: rank r runs a contiguous share of the subtrees rooted there
    (their vertices are its vertex block), every rank runs the few levels above the cut redundantly (their factor is a few
    tens of MB). A solve is  [own subtrees upwards] -> ONE all-reduce (sum) of the updates the subtrees hand to level
    cut - 1 (a few hundred KB; every entry has exactly one non-zero contributor, so the result is exact and independent of
    the reduction order) -> [levels above the cut up and down, own subtrees downwards].  No halo exchange, no dot products,
    nothing per level. Every rank holds the full right-hand side; it returns x on ITS rows (`owned` marks them) and, with
    gather=True, the full x on every rank (one more all-reduce, of V x k floats -- tests and small meshes).

    Every rank factorises the whole matrix (no communication in the constructor; the sharded part of the factor is what it
    reads per solve). M: the matrix as compute_matrix returns it, resident on this rank's GPU.
    """

    def __init__(self, M, group=None, leaf_size=64, arity=4):
        from .solvers import NestedDissectionSolver
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local = NestedDissectionSolver(M, leaf_size=leaf_size, arity=arity, shard=(self.rank, self.P))
        csr = _native.csr_of(M)
        self.device, self.V = csr.device, csr.V
        h = self.local._direct._h
        rk, cnt, cut, per_col = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        mask = np.zeros(self.V, dtype=np.uint8)
        _native.check(_native.lib().ls_direct_shard_info(h, ctypes.byref(rk), ctypes.byref(cnt), ctypes.byref(cut), ctypes.byref(per_col),
                                                         mask.ctypes.data_as(ctypes.c_void_p)))
        self.cut_level, self.exchange_floats_per_column = cut.value, per_col.value
        self.owned = torch.from_numpy(mask.astype(bool)).to(self.device)
        self._exchange = {}
        self.method = "nested-dissection"
        self.last_info = dict(iterations=0, converged=True, method="nested-dissection", exchanges=1 if self.P > 1 else 0)
        # the exchange through the library's own RCCL communicator (ls_dist_*: part 0, all-reduce in place, part 1 as ONE native
        # call on the solve's stream, no host round trip) when the job runs on RCCL; torch.distributed only ships the 128-byte id
        self._comm = ctypes.c_void_p(None)
        want = os.environ.get("LARGESTEPS_NATIVE_COLLECTIVE", "1") != "0"
        if want and self.P > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl":
            self._native_collective()

    def _native_collective(self):
        lib = _native.lib()
        ident = torch.zeros(128, dtype=torch.uint8)
        ok = torch.ones(1, dtype=torch.int32, device=self.device)
        if self.rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            if lib.ls_dist_unique_id(buf) == 0:
                ident = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            else:
                ok.zero_()
        ident = ident.to(self.device)
        dist.broadcast(ident, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            return                                   # no RCCL for the library on some rank: every rank keeps the torch.distributed path
        raw = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(ident.cpu().numpy().tobytes()))
        h = ctypes.c_void_p(None)
        with torch.cuda.device(self.device):
            rc = lib.ls_dist_create(raw, self.rank, self.P, self.device.index, ctypes.byref(h))
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            if rc == 0:
                lib.ls_dist_destroy(h)
            return
        self._comm = h
        # never trust an untested transport with the job: one solve through each path, same right-hand side on every rank; the
        # exchange sum is exact (one non-zero contributor per entry), so the two results must be IDENTICAL on this rank's rows.
        # Every rank runs the SAME sequence of collectives whatever happens locally: success is agreed on (all-reduce MIN) after each
        # step separately, and a step that contains a collective is only entered when every rank got through the step before it -- a
        # rank that raised in one solve must not leave its peers waiting inside the other solve's all-reduce.
        def agreed(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return int(t.item()) == 1

        def give_up():
            lib.ls_dist_destroy(h)
            self._comm = ctypes.c_void_p(None)

        gen = torch.Generator(device="cpu").manual_seed(1234)
        bt = torch.randn((self.V, 3), generator=gen, dtype=torch.float32).to(self.device)
        xa = xb = None
        self._comm = ctypes.c_void_p(None)
        try:                                         # step 1: the torch.distributed path (what the job falls back to)
            xb = self.solve(bt)
            torch.cuda.synchronize(self.device)
            fine = True
        except Exception:
            fine = False
        if not agreed(fine):
            return give_up()
        try:                                         # step 2: this rank's local half of a native solve (no collective inside)
            probe = torch.zeros_like(bt)
            region, n_region = ctypes.c_void_p(None), ctypes.c_int64(0)
            with torch.cuda.device(self.device):
                _native.check(lib.ls_direct_exchange_region(self.local._direct._h, 3, ctypes.byref(region), ctypes.byref(n_region)))
                _native.check(lib.ls_direct_solve_part(self.local._direct._h, _native.ptr(bt), _native.ptr(probe), 3, 0, region,
                                                       _native.stream_of(self.device)))
            torch.cuda.synchronize(self.device)
            fine = True
        except Exception:
            fine = False
        if not agreed(fine):
            return give_up()
        self._comm = h
        try:                                         # step 3: the native solve (part 0 -> RCCL all-reduce in place -> part 1)
            xa = self.solve(bt)
            torch.cuda.synchronize(self.device)
            fine = bool(torch.equal(xa[self.owned], xb[self.owned]))
        except Exception:
            fine = False
        if not agreed(fine):
            return give_up()
        self.last_info["collective"] = "ls_dist (RCCL, in place, on the solve's stream)"

    def close(self):
        """Destroy the library's communicator (collective-free: ncclCommDestroy of this rank's handle). Call it before the process group
        goes away; __del__ only does it while the interpreter is still alive."""
        c = getattr(self, "_comm", None)
        if c is not None and c.value:
            try:
                _native.lib().ls_dist_destroy(c)
            except Exception:
                pass
        self._comm = ctypes.c_void_p(None)

    def __del__(self):
        try:                                           # (at interpreter shutdown even `import sys` can raise: then there is nothing to do)
            import sys
            if sys is None or sys.is_finalizing():    # ncclCommDestroy may block on peers that are already gone
                return
            self.close()
        except Exception:
            pass

    def info(self):
        return self.local.info()

    def communicator(self):
        """(rank, world) as the library's RCCL communicator reports them (ncclCommUserRank / ncclCommCount), None without one"""
        if not self._comm.value:
            return None
        rk, wd = ctypes.c_int(-1), ctypes.c_int(-1)
        _native.check(_native.lib().ls_dist_info(self._comm, ctypes.byref(rk), ctypes.byref(wd)))
        return rk.value, wd.value

    def profile_parts(self, b, repeats=5):
        """The three pieces of a sharded solve on THIS rank, from HIP events on the solve's stream (mean of `repeats` solves, us):
        part 0 (own subtrees upwards), the all-reduce of the exchange region, part 1 (replicated levels + own subtrees downwards).
        The collective's figure includes the wait for the slowest rank's part 0. Same launches as solve(), issued piece by piece."""
        b = b.contiguous()
        k, dev, lib, h = b.shape[1], self.device, _native.lib(), self.local._direct._h
        x = torch.zeros_like(b)
        if self.P == 1:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            tot = 0.0
            for _ in range(repeats):
                ev[0].record(); self.local._direct.solve(b, x); ev[1].record()
                torch.cuda.synchronize(dev)
                tot += ev[0].elapsed_time(ev[1]) * 1e3 / repeats
            return dict(part0_us=tot, collective_us=0.0, part1_us=0.0)
        ex = self._exchange.get(k)
        if ex is None:
            ex = self._exchange[k] = torch.zeros(max(1, self.exchange_floats_per_column * k), dtype=torch.float32, device=dev)
        acc = [0.0, 0.0, 0.0]
        for _ in range(repeats):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.cuda.device(dev):
                ev[0].record()
                _native.check(lib.ls_direct_solve_part(h, _native.ptr(b), _native.ptr(x), k, 0, _native.ptr(ex), _native.stream_of(dev)))
                ev[1].record()
                if self._comm.value:
                    _native.check(lib.ls_dist_allreduce_sum(self._comm, _native.ptr(ex), ex.numel(), _native.stream_of(dev)))
                else:
                    _all_reduce_sum(ex, self.group)
                ev[2].record()
                _native.check(lib.ls_direct_solve_part(h, _native.ptr(b), _native.ptr(x), k, 1, _native.ptr(ex), _native.stream_of(dev)))
                ev[3].record()
            torch.cuda.synchronize(dev)
            for i in range(3):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) * 1e3 / repeats
        return dict(part0_us=acc[0], collective_us=acc[1], part1_us=acc[2])

    def solve(self, b, gather=False):
        _native.require_device(b, "b")
        if b.dim() != 2 or b.shape[0] != self.V or not 1 <= b.shape[1] <= 4 or b.dtype != torch.float32:
            raise ValueError(f"expected a float32 ({self.V}, 1..4) right-hand side, got {tuple(b.shape)} {b.dtype}")
        b = b.contiguous()
        k = b.shape[1]
        x = torch.zeros_like(b)
        lib, h, dev = _native.lib(), self.local._direct._h, self.device
        if self.P == 1:
            self.local._direct.solve(b, x)
            return x
        if self._comm.value and not gather:
            with torch.cuda.device(dev):
                _native.check(lib.ls_dist_direct_solve(self._comm, h, _native.ptr(b), _native.ptr(x), k, _native.stream_of(dev)))
            return x
        ex = self._exchange.get(k)
        if ex is None:
            ex = self._exchange[k] = torch.zeros(max(1, self.exchange_floats_per_column * k), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.ls_direct_solve_part(h, _native.ptr(b), _native.ptr(x), k, 0, _native.ptr(ex), _native.stream_of(dev)))
        _all_reduce_sum(ex, self.group)
        with torch.cuda.device(dev):
            _native.check(lib.ls_direct_solve_part(h, _native.ptr(b), _native.ptr(x), k, 1, _native.ptr(ex), _native.stream_of(dev)))
        if gather:
            x = x * self.owned[:, None]
            _all_reduce_sum(x, self.group)
        return x


class MultiDeviceDirect:
    """The subtree-sharded direct solver driven by ONE process over several devices (SURVEY.md section 8e "process model"): what
    `CholeskySolver` builds when LARGESTEPS_DEVICES names more than one device, so that the reference's call sites
    (`from_differential(M, u, 'Cholesky')`, scripts/main.py:173) use several GPUs unchanged -- no torchrun, no rank logic.

    Device r holds shard r of the factor (`ls_direct_factor(shard_rank = r, shard_count = N)`; the matrix is copied to it once).
    A solve:  b -> every device (peer copies);  part 0 on every device;  the exchange buffers summed -- RCCL
    (`torch.cuda.nccl.all_reduce`, one communicator over the N devices) when the devices are distinct, peer copies through the
    first device when a device is listed twice (the loopback form the 1-GPU tests run);  part 1;  every device's own rows of x
    back to the first device. All launches are asynchronous, ordered by torch's streams. The host issues ~10 operations per
    device and solve, so this mode pays from a few million vertices on (a 1M-vertex solve is 0.24 ms on ONE device); the
    one-process-per-GPU form (`ShardedDirect` under torchrun) has no such overhead."""

    def __init__(self, M, devices, leaf_size=None, arity=None):
        from .solvers import _NativeDirect
        csr = _native.csr_of(M)
        if not _native.is_symmetric(csr):
            raise ValueError("MultiDeviceDirect: the matrix is not symmetric")
        self.V, self.home = csr.V, csr.device
        self.devices = [torch.device("cuda", int(d)) for d in devices]
        n = len(self.devices)
        if n < 2:
            raise ValueError("MultiDeviceDirect needs at least two device entries")
        self.loopback = len({d.index for d in self.devices}) < n
        self._csr, self.parts, self.owned, self._ex = [], [], [], {}
        per_col = None
        for r, dev in enumerate(self.devices):
            if dev == self.home:
                c = csr
            else:
                c = _native.CsrMatrix(csr.V, csr.rowptr.to(dev), csr.col.to(dev), csr.val.to(dev), symmetric=csr.symmetric, a_min=csr.a_min,
                                      uniform=csr.uniform, positions=None if csr.positions is None else csr.positions.to(dev))
            self._csr.append(c)
            part = _NativeDirect(c, leaf_size, arity, -1, True, shard=(r, n))
            rk, cnt, cut, pc = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
            mask = np.zeros(self.V, dtype=np.uint8)
            _native.check(_native.lib().ls_direct_shard_info(part._h, ctypes.byref(rk), ctypes.byref(cnt), ctypes.byref(cut), ctypes.byref(pc),
                                                             mask.ctypes.data_as(ctypes.c_void_p)))
            idx = torch.from_numpy(np.flatnonzero(mask).astype(np.int64))
            self.parts.append(part)
            self.owned.append((idx.to(dev), idx.to(self.home)))
            per_col = pc.value
        self.exchange_floats_per_column = per_col
        self.method = "nested-dissection"
        self.last_info = dict(iterations=0, converged=True, method="nested-dissection", devices=[d.index for d in self.devices],
                              transport="peer copies (loopback)" if self.loopback else "RCCL (torch.cuda.nccl)")

    def solve(self, b, backward=False):
        _native.require_device(b, "b")
        if b.device != self.home:
            raise RuntimeError(f"matrix ({self.home}) and b ({b.device}) must be on the same device")
        if b.dtype != torch.float32 or b.dim() not in (1, 2) or b.shape[0] != self.V:
            raise ValueError(f"Invalid right-hand side {tuple(b.shape)} {b.dtype}: expected float32 ({self.V}, k)")
        squeeze = b.dim() == 1
        b2 = (b.detach().unsqueeze(1) if squeeze else b.detach()).contiguous()
        out = torch.empty_like(b2)
        for c0 in range(0, b2.shape[1], 4):
            c1 = min(b2.shape[1], c0 + 4)
            out[:, c0:c1] = self._solve4(b2 if (c0 == 0 and c1 == b2.shape[1]) else b2[:, c0:c1].contiguous())
        return out.squeeze(1) if squeeze else out

    def _solve4(self, b):
        lib, k = _native.lib(), b.shape[1]
        bs = [b if d == self.home else b.to(d, non_blocking=True) for d in self.devices]
        xs = [torch.empty_like(t) for t in bs]
        ex = self._ex.get(k)
        if ex is None:
            ex = self._ex[k] = [torch.zeros(max(1, self.exchange_floats_per_column * k), dtype=torch.float32, device=d) for d in self.devices]
        for part, d, br, xr, er in zip(self.parts, self.devices, bs, xs, ex):
            with torch.cuda.device(d):
                _native.check(lib.ls_direct_solve_part(part._h, _native.ptr(br), _native.ptr(xr), k, 0, _native.ptr(er), _native.stream_of(d)))
        if self.loopback:
            total = ex[0].clone()
            for er in ex[1:]:
                total += er.to(self.devices[0])
            for er in ex:
                er.copy_(total)
        else:
            torch.cuda.nccl.all_reduce(ex)
        for part, d, br, xr, er in zip(self.parts, self.devices, bs, xs, ex):
            with torch.cuda.device(d):
                _native.check(lib.ls_direct_solve_part(part._h, _native.ptr(br), _native.ptr(xr), k, 1, _native.ptr(er), _native.stream_of(d)))
        x = torch.empty_like(b)
        for (on_dev, on_home), xr in zip(self.owned, xs):
            x.index_copy_(0, on_home, xr.index_select(0, on_dev).to(self.home, non_blocking=True))
        return x


def _all_reduce_min(t, group):
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)


def _all_reduce_sum(t, group):
    """all-reduce of a device tensor; the gloo backend (loopback tests: several ranks on one GPU) is staged through the host"""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def pick_depth(rowptr, col, V, P, max_depth=64, max_overhead=1.0):
    """Largest halo depth whose redundantly computed ghost rows stay below `max_overhead` of the owned rows on every
    rank (banded orderings: a layer is one 'grid row'; badly ordered meshes fall back to depth 1). The default is
    generous on purpose: at strong-scaling sizes a shard's kernels take microseconds while every exchange costs a
    host-driven RCCL group launch, so recomputing ghost rows is far cheaper than talking more often."""
    if P == 1:
        return 1
    rowptr, col = np.ascontiguousarray(rowptr, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32)
    bounds = block_bounds(V, P)
    best = max_depth
    sizes = np.zeros(max_depth, dtype=np.int64)
    for q in range(P):
        _native.check(_native.lib().ls_shard_layer_sizes(int(V), rowptr.ctypes.data_as(ctypes.c_void_p), col.ctypes.data_as(ctypes.c_void_p),
                                                         int(bounds[q]), int(bounds[q + 1]), int(max_depth), sizes.ctypes.data_as(ctypes.c_void_p)))
        own = bounds[q + 1] - bounds[q]
        extra, d = 0, 1
        for j in range(max_depth - 1):                          # depth j+2 computes layers 1..j+1
            extra += int(sizes[j])
            if extra > max_overhead * own:
                break
            d = j + 2
        best = min(best, d)
    return max(1, best)


def _common_grid(n_max):
    block = 256
    T = -(-n_max // block)
    return block, (max(T, 1) if T < 8 else min((T + 7) & ~7, 1024))


def shard_from_matrix(M, group=None, device=None, method="auto", depth=None, **solver_kw):
    """Convenience: every rank holds the full matrix M (as compute_matrix returns it) on its GPU; build this
    rank's plan + HIP ops + driver. method: 'chebyshev' (needs the spectral enclosure compute_matrix attaches),
    'pcg', or 'auto'. Returns (plan, solver)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = dist.get_world_size(group) if dist.is_initialized() else 1
    csr = _native.csr_of(M)
    rowptr, col, val = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), csr.val.cpu().numpy()
    dev = device if device is not None else csr.device
    cheb = method == "chebyshev" or (method == "auto" and csr.a_min is not None)
    if cheb and csr.a_min is None:
        raise ValueError("the Chebyshev solver needs a matrix built by compute_matrix (spectral enclosure)")
    if cheb:
        d = depth if depth is not None else pick_depth(rowptr, col, csr.V, P)
        plan = ShardPlan.build(rowptr, col, val, csr.V, P, rank, depth=d)
        # every rank launches the same grid (sized on the largest extended block)
        n_max = torch.tensor([plan.n_rows], dtype=torch.int64, device=dev if (P > 1 and dist.get_backend(group) == "nccl") else "cpu")
        if P > 1:
            dist.all_reduce(n_max, op=dist.ReduceOp.MAX, group=group)
        block, grid = _common_grid(int(n_max.item()))
        ops = HipShardOps(plan, dev, grid=grid, block=block)
        solver = ShardedChebyshev(plan, ops, csr.a_min, group=group, **solver_kw)
        if method == "auto" and solver.schedule(solver.rtol)[0] > 400:      # loose enclosure: PCG wins (see solvers.py)
            cheb = False
        else:
            return plan, solver
    plan = ShardPlan.build(rowptr, col, val, csr.V, P, rank, depth=1)
    block, grid = _common_grid(int(np.diff(block_bounds(csr.V, P)).max()))
    ops = HipShardOps(plan, dev, grid=grid, block=block)
    return plan, ShardedPCG(plan, ops, group=group, **solver_kw)


def bench_sharded(workload, device, steps, warmup, shard="auto"):
    """bench.py's N > 1 leg: one from_differential solve of the whole mesh over the ranks of the default group.
    shard = 'vertex' / 'auto' (vertex blocks = subtrees of the direct solver's elimination tree, one small all-reduce per solve),
    'columns' (right-hand sides across ranks), 'halo' (contiguous vertex blocks of the Chebyshev / PCG iteration with halo
    exchange) or 'replicas' (every rank solves its own copy of the system -- independent meshes, weak scaling)."""
    import time
    from . import synthetic
    from .geometry import compute_matrix
    from .parameterize import to_differential
    from .solvers import CholeskySolver

    rank, world = dist.get_rank(), dist.get_world_size()
    v, f, cfg = synthetic.config_mesh(workload)
    tv, tf = torch.from_numpy(v).to(device), torch.from_numpy(f).to(device)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u_full = to_differential(M, tv)
    k = u_full.shape[1]
    if shard == "auto":
        shard = "vertex"
    if shard == "vertex":
        # the north star's partition: vertex blocks = subtrees of the direct solver's elimination tree, one summed exchange per solve
        try:
            sd = ShardedDirect(M)
        except (ValueError, RuntimeError):
            sd = None            # no direct factorisation for this matrix: vertex blocks of the iteration instead
        # every rank must take the same branch (a rank alone in the fallback's collectives would hang the job): agree on the minimum
        ok = torch.tensor([1 if sd is not None else 0], dtype=torch.int32, device=device)
        _all_reduce_min(ok, None)
        if int(ok.item()) == 0:
            sd = None
        if sd is not None:
            x = None
            for _ in range(warmup):
                x = sd.solve(u_full)
            dist.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                x = sd.solve(u_full)
            torch.cuda.synchronize(device)
            dist.barrier()
            elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
            err = (x - tv)[sd.owned].abs().max().reshape(1).double()
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
            inf = sd.info()
            words = torch.tensor([inf["factor_entries"]], dtype=torch.int64, device=device)
            dist.all_reduce(words)              # every rank reads its share of the bottom levels and all of the replicated top
            rows = torch.tensor([int(sd.owned.sum())], dtype=torch.int64, device=device)
            dist.all_reduce(rows, op=dist.ReduceOp.MAX)
            # after the timed region: every rank's three pieces of a solve by HIP events, its device and what its RCCL communicator says
            parts = sd.profile_parts(u_full, repeats=max(1, min(steps, 5)))
            comm = sd.communicator()
            mine = dict(rank=rank, device=torch.cuda.get_device_name(device), device_index=device.index, own_rows=int(sd.owned.sum()),
                        kernel_us=parts["part0_us"] + parts["part1_us"], communicator_rank_world=list(comm) if comm else None, **parts)
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            # Self-check AFTER the timed region, so that a first run on real multi-GPU hardware cannot report a wrong answer as a throughput:
            # (1) every row of x has exactly one designated owner; (2) the sharded x, assembled from the owners' rows, against ONE unsharded
            # ls_direct_solve of the same system (every rank holds the whole matrix): max-abs <= 2e-5 of max |x| -- two fp32 evaluations of
            # the same factorisation in a different summation order; (3) the library's RCCL communicator reports the job's world size.
            from .solvers import NestedDissectionSolver
            cover = sd.owned.to(torch.int32).clone()
            dist.all_reduce(cover)
            x_all = torch.where(sd.owned.unsqueeze(1), x, torch.zeros_like(x))
            dist.all_reduce(x_all)
            single = NestedDissectionSolver(M)
            x_one = single.solve(u_full)
            single.close()
            scale = float(x_one.abs().max())
            diff = (x_all - x_one).abs().max().reshape(1).double()
            dist.all_reduce(diff, op=dist.ReduceOp.MAX)
            comm_ok = all(e["communicator_rank_world"] is None or e["communicator_rank_world"][1] == world for e in everyone)
            check = dict(max_abs_diff_vs_unsharded=float(diff.item()), max_abs_x=scale, tolerance_rel=2e-5,
                         every_row_has_one_owner=bool((cover == 1).all().item()), communicator_world_matches=bool(comm_ok))
            check["ok"] = bool(check["max_abs_diff_vs_unsharded"] <= 2e-5 * scale and check["every_row_has_one_owner"] and comm_ok)
            return dict(per_rank=everyone, devices=[e["device_index"] for e in everyone], shard_check=check,
                        communicator=(dict(kind="ls_dist (library's own RCCL communicator, all-reduce in place on the solve's stream)", ranks=comm[1]) if comm else
                                      dict(kind=f"torch.distributed all_reduce ({dist.get_backend()})", ranks=world)),
                        V=v.shape[0], nnz=int(M._nnz()), ms_per_step=float(elapsed.item()) / steps * 1e3, err=float(err.item()), shard="vertex",
                        iterations=0, converged=True, halo=int(sd.exchange_floats_per_column * k), method="nested-dissection", depth=sd.cut_level,
                        rows_per_rank=int(rows.item()), solve_bytes=int(4 * int(words.item()) + 4 * k * 4 * v.shape[0]),
                        solver=(f"HIP nested-dissection direct solver sharded by subtrees of its elimination tree: cut at tree level {sd.cut_level}, "
                                f"every rank runs its subtrees + the replicated levels above, ONE all-reduce of {sd.exchange_floats_per_column * k * 4 / 1024:.0f} "
                                f"KiB per solve (RCCL), x stays sharded by vertex block; {inf['launches']} launches per rank and solve"))
        shard = "halo"
    if shard == "replicas":
        local = CholeskySolver(M)
        solver = local
        u, ref, plan = u_full, tv, None
        pick = lambda x: x                                          # noqa: E731
    elif shard == "columns":
        # the single-GPU default path on this rank's columns; fewer columns per rank -> deeper patch plan
        local = CholeskySolver(M, patch_columns=max(1, min(3, -(-k // min(world, k)))))
        solver = ColumnSharded(lambda bc: local.solve(bc), k)
        u, ref, plan = u_full, tv, None
        pick = lambda x: x                                          # noqa: E731
    else:
        plan, solver = shard_from_matrix(M, device=device, rtol=1e-6)
        u, ref = u_full[plan.lo:plan.hi].contiguous(), tv[plan.lo:plan.hi]
        pick = lambda x: x                                          # noqa: E731
    x = None
    for _ in range(warmup):
        x = solver.solve(u)
    dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        x = solver.solve(u)
    torch.cuda.synchronize(device)
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    err = (pick(x) - ref).abs().max().reshape(1).double()
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    out = dict(V=v.shape[0], nnz=int(M._nnz()), ms_per_step=float(elapsed.item()) / steps * 1e3, err=float(err.item()), shard=shard)
    if shard == "replicas":
        out.update(iterations=0, converged=True, halo=0, method=local.method, depth=0, rows_per_rank=v.shape[0], replicas=world,
                   solve_bytes=(int(world * 4 * local.info()["factor_entries"]) if local.method == "nested-dissection" else None),
                   solver=f"{world} independent replicas of the single-GPU solver ({local.method}), one full system per rank, no communication")
        return out
    if shard == "columns":
        info = local.last_info if local.last_info is not None else dict(iterations=0, converged=True, method="idle")
        its = torch.tensor([info["iterations"], int(info["converged"])], dtype=torch.int64, device=device)
        dist.all_reduce(its, op=dist.ReduceOp.MAX)
        active = min(world, k)
        if local.method == "nested-dissection":
            what = (f"HIP nested-dissection direct solver (factor once per rank, {local.info()['launches']} launches per re-solve)")
            # every active rank reads the whole factor for its column(s)
            out["solve_bytes"] = int(active * 4 * local.info()["factor_entries"] + 4 * k * 4 * v.shape[0])
        else:
            what = "HIP Chebyshev-Jacobi / Jacobi-PCG iteration"
            out["solve_bytes"] = None
        out.update(iterations=int(its[0]), converged=bool(its[1]), halo=0, method=local.method, depth=0, rows_per_rank=v.shape[0],
                   solver=(f"{what}; the {k} right-hand-side columns solved on {active} of {world} ranks, one all-gather of the "
                           f"solution per solve (RCCL), no per-iteration communication"))
        return out
    halo = torch.tensor([plan.n_halo], dtype=torch.int64, device=device)
    dist.all_reduce(halo, op=dist.ReduceOp.MAX)
    info = solver.last_info
    cheb = isinstance(solver, ShardedChebyshev)
    desc = (f"HIP Chebyshev-Jacobi sharded over {world} vertex blocks, halo depth {plan.depth} "
            f"({info.get('exchanges', 0)} neighbour exchanges per solve, no all-reduce in the iteration, RCCL)") if cheb else \
           f"HIP Jacobi-PCG sharded over {world} vertex blocks (halo isend/irecv + 2 all-reduces per iteration, RCCL)"
    out.update(iterations=info["iterations"], converged=info["converged"], halo=int(halo.item()), solver=desc,
               method="chebyshev" if cheb else "pcg", depth=plan.depth, rows_per_rank=plan.n_rows)
    return out
