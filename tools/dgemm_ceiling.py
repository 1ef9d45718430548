#!/usr/bin/env python3
"""dgemm_ceiling.py — what the vendor FP64 GEMM (rocBLAS / hipBLASLt through torch.mm) sustains on this box.

Calibration for DESIGN.md §4.5 / the `roofline` object of bench.py: the 78.6 TFLOP/s FP64 matrix peak assumes the
boost clock; this prints the practical ceiling a tuned library kernel reaches for shapes like the trailing update
(C[n,n] -= A[n,256] A[n,256]^T) and for a large square product. Dev tool, not part of the product path.
"""
import time
import torch

def bench(fn, flops, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return flops / dt / 1e12, dt * 1e3

def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    for n in (4096, 8192):
        a = torch.randn(n, n, dtype=torch.float64, device=dev)
        b = torch.randn(n, n, dtype=torch.float64, device=dev)
        tf, ms = bench(lambda: torch.mm(a, b), 2.0 * n ** 3)
        print(f"torch.mm f64 {n}x{n}x{n}: {tf:.1f} TFLOP/s ({ms:.2f} ms)")
    for n in (6400, 12800):
        a = torch.randn(n, 256, dtype=torch.float64, device=dev)
        c = torch.randn(n, n, dtype=torch.float64, device=dev)
        tf, ms = bench(lambda: torch.addmm(c, a, a.t(), alpha=-1.0, out=c), 2.0 * n * n * 256)
        print(f"torch.addmm f64 rank-256 update of {n}x{n} (full square, 2x the triangle's work): {tf:.1f} TFLOP/s ({ms:.2f} ms)")

if __name__ == "__main__":
    main()
