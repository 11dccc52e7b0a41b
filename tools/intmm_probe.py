import torch, time
for nn in (16, 32, 64, 1024):
    for mk in (1024, 4096):
        A = torch.randint(-128, 128, (mk, mk), dtype=torch.int8).cuda()
        B = torch.randint(-128, 128, (mk, nn), dtype=torch.int8).cuda()
        try:
            for _ in range(5): C = torch._int_mm(A, B)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): C = torch._int_mm(A, B)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            ref = (A[:64].cpu().long() @ B.cpu().long())
            ok = bool((C[:64].cpu().long() == ref).all())
            print(f"torch._int_mm {mk}x{mk}x{nn}: {us:.2f} us {2.0*mk*mk*nn/us/1e6:.1f} TOPS exact={ok}")
        except Exception as e:
            print("int_mm failed", mk, nn, repr(e)[:200])
