#!/usr/bin/env python3
"""Yardstick only (never on the product path): what the vendor BLAS (torch.matmul -> hipBLASLt / rocBLAS) reaches on the four
DiT block GEMM shapes at C2 (M = 938) and C3 (M = 60 032), plain bf16 GEMM without the fused epilogues, 22 rotating weight
sets (cold weights, as in the block chain).  Prints us per launch and TFLOP/s; tells how much headroom a hand-written
main loop has at these shapes on this box."""
import sys
import time

import torch


def main():
    dev = "cuda"
    shapes = [("QKV", 3072, 1024), ("OUT", 1024, 1024), ("FF1", 2048, 1024), ("FF2", 1024, 2048)]
    Ms = [int(x) for x in sys.argv[1:]] or [938, 60032]   # optional: row counts on the command line
    for M in Ms:
        for name, N, K in shapes:
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            Ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(22)]
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for w in Ws[:3]:
                torch.matmul(A, w.t(), out=out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for w in Ws:
                    torch.matmul(A, w.t(), out=out)
            g.replay()
            torch.cuda.synchronize()
            reps = 20 if M < 2000 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (reps * 22) * 1e6
            print(f"M={M:6d} {name} N={N} K={K}: {us:8.2f} us per launch in a 22-launch graph, {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s "
                  f"({2.0 * M * N * K / us / 1e6 / 2500:.3f} of 2.5 PF)", flush=True)


if __name__ == "__main__":
    sys.exit(main())
