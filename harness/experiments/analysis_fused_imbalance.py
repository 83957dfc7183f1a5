import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/voltrix-spmm_amd']
import torch, numpy as np
import synth_graphs
from voltrix import hybrid
torch.set_num_threads(8)
t=time.time()
ip, ix, _ = synth_graphs.generate("reddit_like")
n = ip.numel()-1
print("gen", time.time()-t, n, ix.numel(), flush=True)
t=time.time()
ri, rx, pc, rp, inv = hybrid.split_shared_columns(ip, ix, n, n, 512, 3)
print("split", time.time()-t, flush=True)
npanels = (n+511)//512
cnt = torch.bincount(pc // n, minlength=npanels)
ks = (cnt + 31)//32
# residual TC blocks per window: distinct columns per window / 8
deg = (ri[1:]-ri[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n), deg)
key = torch.unique((rows//16)*n + rx.long())
win = key // n
W = (n+15)//16
u = torch.bincount(win, minlength=W)
blocks = torch.clamp((u+7)//8, min=1)
stages = torch.where(u>0, (blocks+3)//4, torch.zeros_like(blocks))
pad = (-W) % 32
st = torch.cat([stages, torch.zeros(pad, dtype=stages.dtype)]).view(-1, 4, 8).sum(2)   # [panel, wave(4)]
recs = st.double()
print("k-steps", int(ks.sum()), "records", int(recs.sum()))
wave_max = recs.max(1).values; wave_mean = recs.mean(1)
print("intra-WG wave imbalance: sum(max)/sum(mean) =", float(wave_max.sum()/wave_mean.sum()))
# WG time model: residual step 1 unit, k-step c units (panel k-step 64 mfma ~ 0.45us vs resid step 0.6us) -> c=0.75
for c in (0.0, 0.75, 1.5):
    wg = wave_max + c*ks.double()
    ideal = (wave_mean + c*ks.double()).sum()/256
    # LPT per XCD: 8 ranges of consecutive panels, 32 CUs each, longest first list scheduling
    per = (npanels+7)//8
    mk = 0
    for x in range(8):
        w = wg[x*per:(x+1)*per].sort(descending=True).values.tolist()
        cus = [0.0]*32
        for v in w:
            i = cus.index(min(cus)); cus[i]+=v
        mk = max(mk, max(cus))
    print(f"c={c}: makespan/ideal = {mk/ideal:.3f}  (ideal per-CU {ideal:.0f}, makespan {mk:.0f}; mean WG {float(wg.mean()):.0f}, max WG {float(wg.max()):.0f})")
