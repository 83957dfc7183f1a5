"""Decode v_smfmac_f32_16x16x64_f16's operand pairing empirically: which B register element multiplies A's kept slot s of
lane group g under index code c."""
import ctypes, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(os.path.dirname(HERE), "build", "smfmac_probe.so"))
dev = "cuda"
b_ids = (torch.arange(64)[:, None] * 16 + torch.arange(16)[None, :]).to(torch.float16).to(dev).contiguous()
d = torch.zeros(64, 4, dtype=torch.float32, device=dev)


def run(a, idx, abid=0):
    rc = lib.smfmac_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b_ids.data_ptr()), ctypes.c_void_p(idx.data_ptr()),
                          ctypes.c_void_p(d.data_ptr()), abid)
    assert rc == 0
    return d.cpu().numpy().copy()


# 1. D layout + A row mapping: only lane L has a kept value -> which D rows light up
print("== which output rows does A lane L feed (slot 0, code 0)")
for L in (0, 1, 15, 16, 17, 32, 48, 63):
    a = torch.zeros(64, 8, dtype=torch.float16, device=dev)
    a[L, 0] = 1.0
    out = run(a, torch.zeros(64, dtype=torch.int32, device=dev))
    nz = np.argwhere(out != 0)
    rows = sorted({(int(l) >> 4) * 4 + int(i) for l, i in nz})
    cols = sorted({int(l) & 15 for l, i in nz})
    print(f"  lane {L}: D rows (assuming row=4(lane>>4)+i) {rows} cols {cols[:3]}..{cols[-3:] if cols else ''} sample value {out[nz[0][0], nz[0][1]] if len(nz) else None}")

print("== pairing: (g, slot, code) -> B element (lane group, j) per output column 0; consistency over columns")
table = {}
for abid in (0, 1):
    for g in range(4):
        for s in range(8):
            for c in range(4):
                a = torch.zeros(64, 8, dtype=torch.float16, device=dev)
                a[16 * g:16 * g + 16, s] = 1.0
                idx = torch.full((64,), (c << (2 * s)) << (16 * abid), dtype=torch.int64, device=dev).to(torch.int32)
                out = run(a, idx, abid)
                # D[row R][col n] at lane 16*(R//4) + n ... read row 0: lanes 0..15, i=0
                ids = out[0:16, 0].astype(int)
                lanes, js = ids // 16, ids % 16
                ok = all(int(lanes[n]) & 15 == n for n in range(16)) and len(set(js)) == 1 and len(set(l >> 4 for l in lanes)) == 1
                table[(abid, g, s, c)] = (int(lanes[0]) >> 4, int(js[0]), ok)
    print(f"abid {abid}")
    for g in range(4):
        for s in range(8):
            print(f"  g{g} slot{s}: " + "  ".join(f"c{c}->(bg{table[(abid, g, s, c)][0]},j{table[(abid, g, s, c)][1]:2d}{'' if table[(abid, g, s, c)][2] else '!'})" for c in range(4)))
