#!/usr/bin/env python
"""Micro-benchmark: nn.LayerNorm on [rows, 64] (the policy trunk of the rollout) vs a var_mean formulation."""
import torch, time
dev = "cuda:0"
for rows in (90112, 311296):
    x = torch.randn(rows, 64, device=dev); ln = torch.nn.LayerNorm(64).to(dev)
    def manual(x):
        var, mean = torch.var_mean(x, dim=-1, unbiased=False, keepdim=True)
        return (x - mean) * torch.rsqrt(var + ln.eps) * ln.weight + ln.bias
    def native(x):
        return torch.nn.functional.layer_norm(x, (64,), ln.weight, ln.bias, ln.eps)
    gru = torch.nn.GRUCell(64, 64).to(dev); h = torch.randn(rows, 64, device=dev)
    fns = {"nn.LayerNorm": native, "var_mean LN": manual, "GRUCell": lambda x: gru(x, h), "linear 64->192": lambda x: torch.nn.functional.linear(x, gru.weight_ih)}
    with torch.no_grad():
        for name, f in fns.items():
            for _ in range(5): f(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): f(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
            print(f"rows={rows} {name:16s} {dt*1e6:8.1f} us", flush=True)
        print("max diff", (native(x) - manual(x)).abs().max().item())
