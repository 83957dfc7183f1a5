import ctypes, os
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "smfmac_hazard.so"))
torch.manual_seed(0)
dev = "cuda"
# random structured operands; reference computed on the host from the decoded semantics
a = (torch.randint(0, 2, (2, 64, 8)) * 2.0).half()
b = torch.randn(2, 64, 16).half()
idx = torch.randint(0, 2 ** 31 - 1, (64,), dtype=torch.int32)


def ref(av, bv, idxv, abid):
    av, bv = av.float().numpy(), bv.float().numpy()
    out = np.zeros((64, 4), np.float32)
    bmat = np.zeros((64, 16), np.float32)   # [rho][n]: rho = 32 m + 8 bg + e  <->  B lane (n, bg) element 8 m + e
    for lane in range(64):
        n, bg = lane & 15, lane >> 4
        for j in range(16):
            bmat[32 * (j >> 3) + 8 * bg + (j & 7), n] = bv[lane, j]
    for lane in range(64):
        r, g = lane & 15, lane >> 4
        code = (int(idxv[lane]) >> (16 * abid)) & 0xFFFF
        for s in range(8):
            pos = (code >> (2 * s)) & 3
            rho = 16 * g + 4 * (s >> 1) + pos
            for n in range(16):
                out[16 * (r >> 2) + n, r & 3] += av[lane, s] * bmat[rho, n]
    return out


ad, bd, idxd = a.to(dev).contiguous(), b.to(dev).contiguous(), idx.to(dev)
d = torch.zeros(2, 64, 4, dtype=torch.float32, device=dev)
for mode in (0, 1):
    for nops in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        d.zero_()
        assert lib.smfmac_hazard(ctypes.c_void_p(ad.data_ptr()), ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(idxd.data_ptr()),
                                 ctypes.c_void_p(d.data_ptr()), mode, nops) == 0
        got = d.cpu().numpy()
        if mode == 0:
            w0, w1 = ref(a[0], b[0], idx, 0), ref(a[0], b[1], idx, 0)
        else:
            w0, w1 = ref(a[0], b[0], idx, 0), ref(a[1], b[0], idx, 1)
        e0, e1 = np.abs(got[0] - w0).max(), np.abs(got[1] - w1).max()
        print(f"mode {mode} nops {nops:2d}: first max err {e0:.3g}, second max err {e1:.3g}")
