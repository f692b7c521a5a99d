#!/usr/bin/env python
"""Which formulation of the critic's tall-skinny first-layer products does the BLAS back end run fastest?  (learner.py::_value_central:
base = obs_all [b, n*o] @ W_obs^T [n*o, h]; backward: dW = dy^T [h, b] @ obs_all [b, n*o].)  b = 262144, n*o = 2204, h = 64, f32."""
import time
import torch
b, k, h = 262144, 2204, 64
dev = "cuda:0"
A = torch.randn(b, k, device=dev); W = torch.randn(h, k, device=dev); dy = torch.randn(b, h, device=dev)
At = A.t().contiguous()


def bench(name, fn, flops, bytes_):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:58s} {dt * 1e3:7.2f} ms  {flops / dt / 1e12:6.1f} TFLOP/s  {bytes_ / dt / 1e12:5.2f} TB/s", flush=True)
    return out


fl = 2.0 * b * k * h; by = 4.0 * (b * k + b * h)
ref = bench("F.linear(A, W)                      [fwd, as shipped]", lambda: torch.nn.functional.linear(A, W), fl, by)
o = bench("(W @ A.t()).t()", lambda: (W @ A.t()).t(), fl, by); print("   max diff", (o - ref).abs().max().item())
o = bench("A @ W.t().contiguous()", lambda: A @ W.t().contiguous(), fl, by)
o = bench("(W @ At)  [A stored transposed]", lambda: W @ At, fl, by)
Wc = W.t().contiguous()
o = bench("torch.mm(A, Wc)", lambda: torch.mm(A, Wc), fl, by)
for parts in (2, 4, 8):
    ks = k // parts
    o = bench(f"sum of {parts} K-slices (addmm chain)", lambda: sum(A[:, i * ks:(i + 1) * ks if i < parts - 1 else k] @ Wc[i * ks:(i + 1) * ks if i < parts - 1 else k] for i in range(parts)), fl, by)
A3 = A.view(b, 38, 58)
W3 = W.view(h, 38, 58)
o = bench("einsum bno,hno->bh", lambda: torch.einsum("bno,hno->bh", A3, W3), fl, by)
print("backward dW = dy^T @ A")
ref = bench("dy.t() @ A                          [as autograd does]", lambda: dy.t() @ A, fl, by)
o = bench("(A.t() @ dy).t()", lambda: (A.t() @ dy).t(), fl, by); print("   max diff", (o - ref).abs().max().item())
dyt = dy.t().contiguous()
o = bench("dyt @ A  [dy stored transposed]", lambda: dyt @ A, fl, by)
for parts in (4, 16, 64):
    bs = b // parts
    o = bench(f"sum over {parts} row blocks (baddbmm)", lambda: torch.bmm(dy.view(parts, bs, h).transpose(1, 2), A.view(parts, bs, k)).sum(0), fl, by)
print("as the learner calls it: weight = a column slice of fc1.weight [h, n*o + n + n*a], with bias; then the same with autograd")
Wfull = torch.randn(h, k + 38 + 38, device=dev, requires_grad=True); bias = torch.randn(h, device=dev, requires_grad=True)
o = bench("F.linear(A, Wfull[:, :k], bias)   [no grad]", lambda: torch.nn.functional.linear(A, Wfull.detach()[:, :k], bias.detach()), fl, by)
o = bench("F.linear(A, Wfull[:, :k].contiguous(), bias)", lambda: torch.nn.functional.linear(A, Wfull.detach()[:, :k].contiguous(), bias.detach()), fl, by)
act = torch.randn(b, 38, device=dev)
o = bench("F.linear(act_all [b, 38], Wfull[:, k+38:])", lambda: torch.nn.functional.linear(act, Wfull.detach()[:, k + 38:]), 2.0 * b * 38 * h, 4.0 * b * (38 + h))


def fwd_bwd():
    y = torch.nn.functional.linear(A, Wfull[:, :k], bias)
    y.sum().backward()
    return y


o = bench("fwd + bwd through the sliced weight (dW into Wfull.grad)", fwd_bwd, 2 * fl, 2 * by)
Ag = A.clone().requires_grad_(True)


def fwd_bwd_A():
    y = torch.nn.functional.linear(Ag, Wfull[:, :k], bias)
    y.sum().backward()
    return y


o = bench("... also with a gradient wrt A (the policy update's state path?)", fwd_bwd_A, 3 * fl, 3 * by)
x = torch.randn(b * 38, h, device=dev); W2 = torch.randn(h, h, device=dev)
o = bench("trunk: F.linear([b*n, 64], [64, 64])", lambda: torch.nn.functional.linear(x, W2), 2.0 * b * 38 * h * h, 8.0 * b * 38 * h)
