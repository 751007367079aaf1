"""Debug helper: outputs of the 256x128 ring GEMM with deferred vs immediate stores must be bitwise equal
(run twice: with and without W2V2_NO_DEFER=1, compare the printed checksums)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from w2v2_speaker_amd import ops
dev = "cuda"
torch.manual_seed(0)
M = int(os.environ.get("DBG_M", 66 * 149))
out = []
for (N, K, epi) in [(2304, 768, "bias"), (768, 2304, "add"), (3072, 768, "gelu_bwd"), (768, 768, "none"), (1536, 768, "add")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.randn(M, N, device=dev).to(torch.bfloat16)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi == "bias":
        kw = dict(epilogue=ops.EPI_BIAS, bias=bias)
    elif epi == "add":
        kw = dict(epilogue=ops.EPI_ADD, aux=C, ldaux=N)            # in place, as the engine's dX products
    elif epi == "gelu_bwd":
        kw = dict(epilogue=ops.EPI_GELU_BWD, aux=aux, ldaux=N)
    g = ops.Gemm(M, N, K, A, B, C, lda=K, ldb=K, ldc=N, **kw)
    g()
    torch.cuda.synchronize()
    out.append(hashlib.md5(C.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10] + ":" + g.kernel_name[-12:])
print(" ".join(out))
