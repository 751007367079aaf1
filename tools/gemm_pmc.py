"""One GEMM shape of the training step launched a few times (for rocprofv3 --pmc passes) and a summariser.
    rocprofv3 --kernel-trace --pmc <counters> -d DIR -- python3 tools/gemm_pmc.py run qkv
    python3 tools/gemm_pmc.py sum DIR
"""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"qkv": (9834, 2304, 768), "out": (9834, 768, 768), "ffn1": (9834, 3072, 768), "ffn2": (9834, 768, 3072),
          "conv1": (66 * 4799, 512, 1536)}


def run(name):
    import torch
    from w2v2_speaker_amd import ops
    m, n, k = SHAPES[name]
    dt = torch.float16
    A = torch.randn(m, k, device="cuda").to(dt)
    B = (torch.randn(n, k, device="cuda") / k ** 0.5).to(dt)
    C = torch.zeros(m, n, dtype=dt, device="cuda")
    bias = torch.zeros(n, device="cuda")
    g = ops.Gemm(m, n, k, A, B, C, lda=k, ldb=k, ldc=n, epilogue=ops.EPI_BIAS, bias=bias)
    for _ in range(8):
        g()
    torch.cuda.synchronize()


def summarise(d):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    q = ("select k.name, c.counter_name, count(*), avg(c.value), avg(k.end - k.start) from counters_collection c join "
         "kernels k on k.dispatch_id = c.dispatch_id where k.name like '%gemm%' group by k.name, c.counter_name")
    for name, cn, n, v, dur in cur.execute(q):
        print(f"{name[:40]:40s} {cn:32s} n={n} avg={v:14.1f} dur_us={dur / 1e3:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarise(sys.argv[2])
