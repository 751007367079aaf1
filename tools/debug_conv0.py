import sys; sys.path.insert(0, "/root/repo")
import torch
from w2v2_speaker_amd import ops as o
torch.manual_seed(0)
B, N, C, k, s = 1, 4000, 128, 10, 5
wav = torch.randn(B, N); w = 0.4 * torch.randn(C, 1, k)
gamma, beta = torch.ones(C), torch.zeros(C)
u = torch.nn.functional.conv1d(wav[:, None].double(), w.double(), stride=s)
L = u.shape[2]
for dt in (torch.float32, torch.bfloat16):
    out = torch.zeros(B, L, C, dtype=dt, device="cuda")
    work = o.conv0_workspace(B, N, C, k, s, "cuda")
    o.conv0_groupnorm_gelu(wav.cuda(), w.cuda(), gamma.cuda(), beta.cuda(), out, work, k, s)
    torch.cuda.synchronize()
    nchunk = (L + 127) // 128
    part = work[:B * nchunk * C * 2].view(B, nchunk, C, 2).cpu().double()
    ref1 = torch.stack([u[0, :, j * 128:(j + 1) * 128].sum(dim=1) for j in range(nchunk)])     # [nchunk, C]
    d = (part[0, :, :, 0] - ref1).abs()
    print(dt, "partial sum err max", d.max().item(), "per chunk", d.max(dim=1).values.tolist())
    print("   worst channel per chunk", d.argmax(dim=1).tolist())
