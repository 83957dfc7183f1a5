import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/voltrix-spmm_amd"]
os.environ.setdefault("VOLTRIX_CACHE_DIR", "/root/repo/voltrix-spmm_amd/.jit_cache")
import torch, synth_graphs, voltrix
from voltrix.jit_kernels.spmm import graph_bucket_keys
from voltrix.jit_kernels.tuner import jit_tuner
dev = torch.device("cuda")
ip, ix, _ = synth_graphs.generate("products_like", device=dev, scale=1.0)
n = ip.numel() - 1
os.environ["VOLTRIX_HYBRID"] = "0"
h = voltrix.csr_preprocess_device(ip, ix, n)
keys = {"feature_hash": "x", "embedding_dim": 512, "dtype": "torch.float16", "device": torch.cuda.get_device_name(dev), "two_level": False, "weighted": False}
bk = graph_bucket_keys(h[0], n, keys)
sig = jit_tuner._signature("spmm_kernel@bucket", bk)
st = jit_tuner._load_store()
k = f"{sig[0]}|{sig[1]}"
print(k)
print(k in st, [x for x in st if "512" in x and "'log2_rows': 21" in x])
