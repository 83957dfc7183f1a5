import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'voltrix-spmm_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch, synth_graphs
from oracle import oracle_c
from voltrix.schedule import build_stage_list
name = sys.argv[1] if len(sys.argv) > 1 else 'reddit_uniform'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
indptr, indices, cfg = synth_graphs.generate(name, scale=scale)
N = indptr.numel()-1
p1, packed, hind = oracle_c.csr_preprocess(indptr.numpy(), indices.numpy(), N)
first_col = hind.reshape(-1,8)[:,0]
npanel = 29
panel_rows = N // npanel + 1
for groups, nw, bal in ((4, 256, False), (4, 256, True), (8, 128, True), (2, 384, True)):
    sl = build_stage_list(torch.from_numpy(p1), torch.from_numpy(packed.view(np.int32)).view(torch.uint32), torch.from_numpy(hind), N, num_waves=nw, groups=groups, depth=3, mode='sweep', panel_rows=panel_rows, near_rows=0, balance=bal)
    ent, wp = sl.entries.numpy(), sl.wave_ptr.numpy()
    L = (wp[1:] - wp[:-1] - 7)
    maxL = L.max()
    # position matrix: wave x time -> fractional sweep position (round + panel/npanel), nan when finished
    posm = np.full((nw, maxL), np.nan)
    for v in range(nw):
        lst = ent[wp[v]:wp[v]+L[v]]
        pan = first_col[lst[:,0]] // panel_rows
        # round index = number of times panel index wrapped
        wrap = np.concatenate([[0], np.cumsum(np.diff(pan) < -npanel//2)])
        posm[v,:L[v]] = wrap * npanel + pan
    # only XCD 0 waves (v % 8 == 0)
    pm = posm[0::8]
    spreads = []
    for k in range(0, maxL, 50):
        col = pm[:,k]; col = col[~np.isnan(col)]
        if len(col) > 4: spreads.append(np.percentile(col, 95) - np.percentile(col, 5))
    print(f"G={groups} waves={nw} bal={bal}: list len mean={L.mean():.0f} std={L.std():.0f} max={maxL}; XCD0 5-95% position spread: mean={np.mean(spreads):.2f} panels, max={np.max(spreads):.2f} (panel = {panel_rows} rows = {panel_rows*256/1e6:.2f} MB)")
