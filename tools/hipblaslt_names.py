"""Calibration helper: run the vendor GEMM on the step's shapes so that rocprofv3 --kernel-trace shows which
Tensile macro-tile / depth / workgroup hipBLASLt picks (its kernel names encode them)."""
import torch
M = 66 * 149
for m, n, k in [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (66 * 4799, 512, 1536), (66 * 599, 512, 1536)]:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    print(m, n, k, float(c.float().abs().mean()))
