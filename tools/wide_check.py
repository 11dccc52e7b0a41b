"""Wide-operand FP4 kernel (bitmm_fp4_wide.hip.h): popcount engine vs mfma engine on ragged shapes in all three output
forms (raw random words, padding bits included where the layout allows), then timings at N = 512 / 1024 / 2048.
The library reads its diagnostic switches once per process, so the two timing columns come from two child processes
(`wide_check.py time-one` under QGTC_NO_WIDE=1 / unset): the parent only lays their lines side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC

torch.manual_seed(1)
bad = 0
shapes = [(128, 1024, 512), (129, 1024, 257), (1000, 1152, 300), (77, 128, 1000), (513, 2176, 1030), (2048, 4096, 1024), (300, 896, 513), (8, 3968, 264)]
if len(sys.argv) > 1 and sys.argv[1] == "time":
    shapes = []
for (M, K, N) in shapes:
  for (a, w) in ((1, 1), (1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (2, 4), (4, 2), (1, 8), (2, 8), (8, 1), (8, 2)):
    A = torch.randint(0, 2 ** a, (M, K)).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    for ob in (1, 3, 10):
        outs = {}
        for eng in ("popcount", "mfma"):
            QGTC.set_engine(eng)
            outs[eng] = (QGTC.bitMM2Bit(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Bit_col(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Int(bA, bX, M, K, N, a, w))
        for i, name in enumerate(("rows", "cols", "int")):
            ok = torch.equal(outs["popcount"][i], outs["mfma"][i])
            if not ok:
                bad += 1
                d = (outs["popcount"][i] != outs["mfma"][i])
                print(f"MISMATCH {M}x{K}x{N} a={a} w={w} ob={ob} {name}: {int(d.sum())} of {d.numel()} differ; first at {d.flatten().nonzero()[:4].flatten().tolist()}")
    print(f"{M}x{K}x{N} a={a} w={w} checked", flush=True)
print("mismatches:", bad)
TIMED = ((4096, 4096, 512), (4096, 4096, 1024), (8192, 4096, 1024), (8192, 8192, 2048), (4096, 4096, 2048))
PLANES = ((1, 1), (1, 2), (2, 2), (1, 4), (2, 4), (4, 1), (1, 8), (2, 8))
if len(sys.argv) > 1 and sys.argv[1] == "time-one":   # one column: whatever QGTC_NO_WIDE this process started with
    for (M, K, N) in TIMED:
        for (a, w) in PLANES:
            A = torch.randint(0, 2 ** a, (M, K)).float().cuda()
            X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
            bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
            QGTC.set_engine("mfma")
            QGTC.profile(bA, bX, M, K, N, a, w, w, 5)
            us = min(QGTC.profile(bA, bX, M, K, N, a, w, w, 50) for _ in range(3)) * 1e3 / 50
            print(f"T {M} {K} {N} {a} {w} {us:.3f}", flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "time":   # (this process has not touched the GPU: `time` skips the checks above)
    import subprocess
    cols = {}
    for name, env_val in (("128-tile", "1"), ("wide", None)):
        env = dict(os.environ)
        env.pop("QGTC_NO_WIDE", None)
        if env_val:
            env["QGTC_NO_WIDE"] = env_val
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "time-one"], env=env, capture_output=True, text=True, check=True).stdout
        for l in out.splitlines():
            if l.startswith("T "):
                _, M, K, N, a, w, us = l.split()
                cols.setdefault((int(M), int(K), int(N), int(a), int(w)), {})[name] = float(us)
    for (M, K, N, a, w), c in cols.items():
        print(f"{M}x{K}x{N} a={a} w={w}:" + "".join(f"  {n} {u:7.2f} us ({2.0 * M * K * N / u / 1e6:7.0f} TOPS)" for n, u in c.items()), flush=True)
else:
    print("timings: run `python tools/wide_check.py time` (each switch setting in its own process)")
