"""
One optimisation step as a captured graph, without restructuring the caller's loop.

The loop body of the reference (scripts/main.py:172-208: from_differential -> normals -> render / loss -> backward -> optimizer step)
issues ~45 kernel launches through Python and autograd; at the reference's own mesh sizes (70k-250k vertices) the GPU work of a step
is shorter than the host work that launches it (tools/bench_step.py: 70k vertices 0.46 ms eager, 0.23 ms replayed). Every launch of this
package is capture-safe (the solver, the SpMV, the normals, `AdamUniform(capturable=True)`), so the whole body can be recorded once with
`torch.cuda.graph` and replayed. `CapturedStep` packages the recipe -- warm-up on a side stream, capture, replay -- around a plain
function:

    opt = AdamUniform([u], 3e-2, capturable=True)
    def body():
        v = from_differential(M, u, 'Cholesky')
        n = compute_vertex_normals(v, f, compute_face_normals(v, f))
        loss = (render(v, n) - target).abs().mean()
        opt.zero_grad(set_to_none=True)        # (set_to_none=True: the gradient tensors are then graph-private and reused)
        loss.backward()
        opt.step()
        return loss, v
    step = CapturedStep(body)
    for it in range(steps):
        loss, v = step()                        # tensors owned by the graph: overwritten by the next call (clone what must survive)

Rules (torch.cuda.graph's, restated): shapes, dtypes and the SEQUENCE of operations must not change between calls; tensors the body
reads from outside (u, M, targets) are read at their current addresses -- update them IN PLACE (u is: the optimizer does); nothing in
the body may synchronise with the host (`.item()`, `.cpu()`, printing a tensor); a remesh (new shapes) needs a new CapturedStep.
"""
import torch

__all__ = ["CapturedStep"]


class CapturedStep:
    """Record `body()` once as a device graph and replay it on every call.

    body : callable without arguments; returns a tensor, a tuple / list of tensors, or None
    warmup : eager executions on a side stream before the capture (allocator, lazy initialisations -- e.g. the solver of
             `from_differential` is constructed on the first call --, optimizer state); they are REAL steps
    Call the object to run one step; it returns what `body` returned (the graph's own output tensors).
    """

    def __init__(self, body, warmup=3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep needs a HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.body = body
        self.steps_run = 0
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):
                    self.out = body()
                    self.steps_run += 1
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = body()          # recorded, NOT executed: the capture itself is not a step

    def __call__(self):
        self.graph.replay()
        self.steps_run += 1
        return self.out
