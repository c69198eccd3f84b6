import sys, torch
sys.path.insert(0, "stable-fast_amd")
from sfast.hip import functional as F
dev = "cuda"
for (B, S, H, D, Skv) in [(2, 256, 8, 160, 256), (2, 64, 8, 160, 64), (2, 256, 8, 160, 77), (2, 64, 8, 160, 77), (2, 1024, 8, 80, 1024), (2, 1024, 8, 80, 77)]:
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.float16)
    k = torch.randn(B, Skv, H, D, device=dev, dtype=torch.float16)
    v = torch.randn_like(k)
    res = []
    for var in (2, 4):
        for _ in range(5):
            F.attention(q, k, v, variant=var)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            F.attention(q, k, v, variant=var)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 10)
    print(f"D={D} Sq={S} Skv={Skv}: nw2 {res[0]:.1f} us  nw4 {res[1]:.1f} us")
