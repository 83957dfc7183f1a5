import ctypes, os
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "smfmac_hazard2.so"))
torch.manual_seed(0)
dev = "cuda"
a = (torch.randint(0, 2, (64, 8)) * 2.0).half()
b = torch.randn(4, 64, 16).half()
idx = torch.randint(0, 2 ** 31 - 1, (64,), dtype=torch.int32)
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smfmac_hazard.py")).read().split("def ref")[1].join(["def ref", ""]).split("ad, bd, idxd")[0])
ad = a.view(torch.int32).to(dev).contiguous(); bd = b.to(dev).contiguous(); idxd = idx.to(dev)
d = torch.zeros(4, 64, 4, dtype=torch.float32, device=dev)
for nops, gap in ((0, 0), (1, 0), (4, 0), (8, 0), (0, 1), (0, 2), (0, 4), (0, 8), (8, 8)):
    d.zero_()
    assert lib.smfmac_hazard2(ctypes.c_void_p(ad.data_ptr()), ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(idxd.data_ptr()),
                              ctypes.c_void_p(d.data_ptr()), nops, gap) == 0
    got = d.cpu().numpy()
    print(f"nops {nops} gap {gap}: max err per instruction", [float(np.abs(got[k] - ref(a, b[k], idx, 0)).max()) for k in range(4)])
