import torch, time
torch.manual_seed(0)
for n in (96, 160):
    for B in (64, 256, 512, 1024, 2048):
        A = torch.randn(B, n, n, dtype=torch.float64, device="cuda") + n * torch.eye(n, dtype=torch.float64, device="cuda")
        b = torch.randn(B, n, dtype=torch.float64, device="cuda")
        try:
            torch.cuda.synchronize(); t0 = time.time()
            x, info = torch.linalg.solve_ex(A, b, check_errors=False)
            torch.cuda.synchronize(); t = time.time() - t0
            print(n, B, "solve_ex ok", f"{t*1e3:.1f} ms", int(info.abs().max()), flush=True)
        except Exception as e:
            print(n, B, "solve_ex FAILED", str(e)[:90], flush=True)
        try:
            torch.cuda.synchronize(); t0 = time.time()
            x = torch.linalg.solve(A, b)
            torch.cuda.synchronize(); t = time.time() - t0
            print(n, B, "solve ok", f"{t*1e3:.1f} ms", flush=True)
        except Exception as e:
            print(n, B, "solve FAILED", str(e)[:90], flush=True)
