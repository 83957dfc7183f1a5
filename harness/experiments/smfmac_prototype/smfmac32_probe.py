"""Decode v_smfmac_f32_16x16x32_f16's operand pairing: which B element multiplies A's kept slot s of lane group g under code c."""
import ctypes, os
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(os.path.dirname(HERE), "build", "smfmac32_probe.so"))
dev = "cuda"
b_ids = (torch.arange(64)[:, None] * 8 + torch.arange(8)[None, :]).to(torch.float16).to(dev).contiguous()
d = torch.zeros(64, 4, dtype=torch.float32, device=dev)
for abid in range(4):
    print("abid", abid)
    for g in range(4):
        line = f"  g{g}:"
        for s in range(4):
            cells = []
            for c in range(4):
                a = torch.zeros(64, 4, dtype=torch.float16, device=dev)
                a[16 * g:16 * g + 16, s] = 1.0
                idx = torch.full((64,), (c << (2 * s)) << (8 * abid), dtype=torch.int64, device=dev).to(torch.int32)
                assert lib.smfmac32_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b_ids.data_ptr()), ctypes.c_void_p(idx.data_ptr()),
                                          ctypes.c_void_p(d.data_ptr()), abid) == 0
                ids = d.cpu().numpy()[0:16, 0].astype(int)
                lanes, js = ids // 8, ids % 8
                ok = all(int(lanes[n]) & 15 == n for n in range(16)) and len(set(js)) == 1 and len(set(l >> 4 for l in lanes)) == 1
                cells.append(f"(bg{int(lanes[0]) >> 4},j{int(js[0])}{'' if ok else '!'})")
            line += f" slot{s}:" + "".join(cells)
        print(line)
