import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "smfmac_rate.so"))
lib.rate_ms.restype = ctypes.c_float
out = torch.zeros(512 * 256, device="cuda")
iters = 20000
for sparse in (0, 1, 2, 0, 1, 2):
    ms = lib.rate_ms(sparse, iters, ctypes.c_void_p(out.data_ptr()))
    n = iters * 8 * 2  # MFMAs per SIMD: 2 workgroups per CU x 4 waves -> 2 waves per SIMD
    print(f"{('mfma 16x16x32', 'smfmac 16x16x64', 'smfmac 16x16x32')[sparse]}: {ms:.3f} ms, {ms * 1e6 / n:.2f} ns per instruction per SIMD")
