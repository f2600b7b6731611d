"""Probe: latency of the dense fp64 block operations a block-tridiagonal Kronecker solver needs (B = k*m2)."""
import time, torch
dev = "cuda"
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for B in (128, 384, 512):
    for nb in (1, 8, 22):
        A = torch.randn(nb, B, B, dtype=torch.float64, device=dev); A = A @ A.transpose(1, 2) + B * torch.eye(B, dtype=torch.float64, device=dev)
        E = torch.randn(nb, B, B, dtype=torch.float64, device=dev)
        L = torch.linalg.cholesky(A)
        print("B %4d batch %3d: cholesky %8.1f us | solve_triangular %8.1f us | bmm %8.1f us | cholesky_ex %8.1f | inv_tri %8.1f" % (
            B, nb, t(lambda: torch.linalg.cholesky(A)), t(lambda: torch.linalg.solve_triangular(L, E, upper=False)),
            t(lambda: torch.bmm(E, E.transpose(1, 2))), t(lambda: torch.linalg.cholesky_ex(A, check_errors=False)),
            t(lambda: torch.linalg.solve_triangular(L, torch.eye(B, dtype=torch.float64, device=dev).expand(nb, B, B), upper=False))), flush=True)
