"""Store-shape microbenchmark (store_shape.hip): TB/s of 16 x 128 fp32 tile stores by lane -> address mapping and footprint.
    build here: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o store_shape.so store_shape.hip ; run on the GPU box."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "store_shape.so"))
names = ["16 rows x 64 B", "8 rows x 128 B", "4 rows x 256 B", "2 rows x 512 B", "dword x4 (4 rows x 64 B)"]
for mb in (64, 171, 512, 1600):
    tiles = mb * (1 << 20) // 8192
    out = torch.empty(tiles * 16 * 128, dtype=torch.float32, device="cuda")
    line = f"{mb:5d} MB:"
    for waves_per_cu in (8, 16):
        for mode in range(5):
            grid = 256 * waves_per_cu
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                assert lib.launch_store(mode, ctypes.c_void_p(out.data_ptr()), tiles, 128, grid, ctypes.c_void_p(st)) == 0
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                lib.launch_store(mode, ctypes.c_void_p(out.data_ptr()), tiles, 128, grid, ctypes.c_void_p(st))
            e.record()
            e.synchronize()
            ms = s.elapsed_time(e) / 10
            line += f"  [{waves_per_cu}w m{mode}] {tiles * 8192 / ms / 1e9:5.2f}"
    print(line + "  TB/s", flush=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        out.zero_()
    e.record()
    e.synchronize()
    print(f"        torch zero_: {out.numel() * 4 / (s.elapsed_time(e) / 10) / 1e9:.2f} TB/s")
print("modes:", names)
