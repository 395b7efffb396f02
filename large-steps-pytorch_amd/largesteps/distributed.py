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
    ShardPlan      pure numpy: row block, local column ids, halo / send lists.      (CPU tests)
    ShardedPCG     the iteration driver: collectives + a LocalOps object.           (CPU tests with gloo)
    HipShardOps    LocalOps on the HIP kernels -- the only implementation shipped.  (GPU tests)
"""
import ctypes

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
    """Everything rank `rank` of `P` needs to know about its block of a V x V CSR matrix (host side, numpy).

    rowptr/col/val : local CSR; columns are local ids -- [0, n_own) owned, [n_own, n_own + n_halo) halo
    halo_global    : global vertex id of every halo column (sorted => grouped by owner rank)
    recv           : [(src_rank, offset_in_halo, count)]
    send           : [(dst_rank, local_row_ids int32)]  rows of p this rank must ship each iteration
    """

    def __init__(self, rank, P, lo, hi, rowptr, col, val, halo_global, recv, send):
        self.rank, self.P, self.lo, self.hi = rank, P, int(lo), int(hi)
        self.n_own = int(hi - lo)
        self.n_halo = int(halo_global.shape[0])
        self.n_cols = self.n_own + self.n_halo
        self.rowptr, self.col, self.val = rowptr, col, val
        self.halo_global, self.recv, self.send = halo_global, recv, send

    @staticmethod
    def _halo_of(rowptr, col, lo, hi):
        c = col[rowptr[lo]:rowptr[hi]]
        return np.unique(c[(c < lo) | (c >= hi)])

    @staticmethod
    def build(rowptr, col, val, V, P, rank):
        rowptr = np.asarray(rowptr).astype(np.int64)
        col = np.asarray(col).astype(np.int64)
        val = np.asarray(val, dtype=np.float32)
        if P < 1 or not (0 <= rank < P):
            raise ValueError(f"invalid rank {rank} of {P}")
        if P > max(V, 1):
            raise ValueError(f"cannot cut {V} vertices into {P} non-empty blocks")
        bounds = block_bounds(V, P)
        lo, hi = bounds[rank], bounds[rank + 1]
        s, e = rowptr[lo], rowptr[hi]
        c = col[s:e]
        own = (c >= lo) & (c < hi)
        halo = np.unique(c[~own])
        local = np.where(own, c - lo, (hi - lo) + np.searchsorted(halo, c))
        owner = np.searchsorted(bounds, halo, side="right") - 1
        recv = []
        for q in np.unique(owner):
            idx = np.nonzero(owner == q)[0]
            assert idx[-1] - idx[0] + 1 == idx.shape[0]          # contiguous: halo sorted, blocks contiguous
            recv.append((int(q), int(idx[0]), int(idx.shape[0])))
        send = []
        for q in range(P):
            if q == rank:
                continue
            hq = ShardPlan._halo_of(rowptr, col, bounds[q], bounds[q + 1])
            mine = hq[(hq >= lo) & (hq < hi)]
            if mine.shape[0]:
                send.append((q, (mine - lo).astype(np.int32)))
        return ShardPlan(rank, P, lo, hi, (rowptr[lo:hi + 1] - s).astype(np.int32), local.astype(np.int32),
                         val[s:e].copy(), halo, recv, send)


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
                                                   plan.n_own, plan.n_cols, plan.col.shape[0], _KMAX, dev.index,
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

    def pack(self, idx, k, out):
        _native.check(_native.lib().ls_gather_rows(_native.ptr(self._p_flat), _native.ptr(idx), idx.shape[0], k,
                                                   _native.ptr(out), self.device.index, _native.stream_of(self.device)))

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
        plan, ops = self.plan, self.ops
        if plan.P == 1 or (not plan.recv and not plan.send):
            return
        p = ops.p_ext(k)
        # Loopback transport for single-GPU test boxes: gloo cannot isend/irecv device tensors,
        # so with backend gloo + device tensors the halo rows are staged through host memory. RCCL ("nccl")
        # moves the device buffers directly.
        stage = p.is_cuda and dist.get_backend(self.group) == "gloo"
        reqs, landed = [], []
        for q, idx in ops.send_idx:
            buf = self._sendbuf.get((q, k))
            if buf is None:
                buf = self._sendbuf[(q, k)] = torch.empty((idx.shape[0], k), dtype=torch.float32, device=p.device)
            ops.pack(idx, k, buf)
            reqs.append(dist.P2POp(dist.isend, buf.cpu() if stage else buf, self._peer(q), group=self.group))
        for q, off, cnt in plan.recv:
            dst = p[plan.n_own + off: plan.n_own + off + cnt]
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


def shard_from_matrix(M, group=None, device=None, **solver_kw):
    """Convenience: every rank holds the full matrix M (as compute_matrix returns it) on its GPU; build this
    rank's plan + HIP ops + driver. Returns (plan, ShardedPCG)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = dist.get_world_size(group) if dist.is_initialized() else 1
    csr = _native.csr_of(M)
    plan = ShardPlan.build(csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), csr.val.cpu().numpy(), csr.V, P, rank)
    dev = device if device is not None else csr.device
    # every rank must launch the same grid so that the partial arrays line up: size it on the largest block
    n_max = int(np.diff(block_bounds(csr.V, P)).max())
    block = 256
    T = -(-n_max // block)
    grid = max(T, 1) if T < 8 else min(T & ~7, 1024)
    ops = HipShardOps(plan, dev, grid=grid, block=block)
    return plan, ShardedPCG(plan, ops, group=group, **solver_kw)


def bench_sharded(workload, device, steps, warmup):
    """bench.py's N > 1 leg: strong scaling of one from_differential solve over the ranks of the default group."""
    import time
    from . import synthetic
    from .geometry import compute_matrix
    from .parameterize import to_differential

    rank, world = dist.get_rank(), dist.get_world_size()
    v, f, cfg = synthetic.config_mesh(workload)
    tv, tf = torch.from_numpy(v).to(device), torch.from_numpy(f).to(device)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    plan, solver = shard_from_matrix(M, device=device, rtol=1e-6)
    u = to_differential(M, tv)[plan.lo:plan.hi].contiguous()
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
    err = (x - tv[plan.lo:plan.hi]).abs().max().reshape(1).double()
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    halo = torch.tensor([plan.n_halo], dtype=torch.int64, device=device)
    dist.all_reduce(halo, op=dist.ReduceOp.MAX)
    info = solver.last_info
    return dict(V=v.shape[0], nnz=int(M._nnz()), ms_per_step=float(elapsed.item()) / steps * 1e3, iterations=info["iterations"],
                converged=info["converged"], err=float(err.item()), halo=int(halo.item()),
                solver=f"HIP Jacobi-PCG sharded over {world} vertex blocks (halo isend/irecv + 2 all-reduces per iteration, RCCL)",
                kernel="k_spmv_dot<3> (K1 on the shard's SELL-64 block)", k1_gbs=0.0)
