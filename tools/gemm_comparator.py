"""Measurement only (never part of the product): the vendor library's bf16 GEMM of the output-layer shape on this box, through
torch.mm (hipBLASLt / rocBLAS), next to which bench.py's own gemm_bf16_pipe_kernel number can be read."""
import json
import time

import torch

T, K, N = 32768, 2048, 10000
a = (torch.randn((T, K), device="cuda") * 0.5).clamp_(min=0).to(torch.bfloat16)      # post-ReLU activations
w = (torch.randn((N, K), device="cuda") / 45.0).to(torch.bfloat16)
for _ in range(3):
    c = a @ w.t()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    c = a @ w.t()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps(dict(op="torch.mm bf16 [32768 x 2048] x [2048 x 10000]^T -> bf16", ms=round(dt * 1e3, 4),
                      tflops=round(2.0 * T * K * N / dt / 1e12, 1), note="vendor library comparator, measurement only; output bf16 (the scorer writes f32 scores and fuses bias / prior / arg-min)")))
