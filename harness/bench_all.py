#!/usr/bin/env python3
"""Sweep datasets x feature widths x methods and collect ``results.csv`` -- the role of the reference's
bench/bench_all.py:62-172: for every ``<name>.npz`` of a folder (files with "reorder" in the name are the reordered twins,
bench_all.py:64-66) and every feature width, dump the graph with graph_gen.py, time every method in its own process,
scrape the method's time line (bench_all.py:21-29) and append ``Method,Dataset,FeatDim,Reorder,Time (ms)`` rows
(bench_all.py:75; Reorder is ``N`` / ``Y`` as in bench_all.py:131,158).  Methods that do not reorder run on ``<name>.npz``
(hipSPARSE = torch.sparse.mm and, round 6, rocSPARSE-best here; the reference's TC-GNN / GE-SPMM / RoDe / Sputnik / DTC runners are competitor kernels, out of scope),
Voltrix runs on ``<name>.reorder.npz`` -- written by ``graph_gen.py --write_reorder METHOD`` when it is not there yet, where
the reference expects an externally reordered file -- and, beyond the reference, also on the un-reordered file, so the CSV
shows what the reorder buys.  ``--synthetic reddit_like:0.05,...`` sweeps the stand-in generators instead of a folder.

    python harness/bench_all.py --datasets_folder DIR [--feature_dims 256,512,1024] [--output_file results.csv] [--append]
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

TIME_PATTERN = {"Voltrix": "[Voltrix] time: ", "Voltrix-fp16": "[Voltrix] time: ", "hipSPARSE": "[hipSPARSE] Elapsed time: ",
                # round 6: rocSPARSE's generic SpMM, the best of its four CSR algorithms (harness/bm_rocsparse.py; -fp16: its fp16-in path)
                "rocSPARSE-best": "[rocSPARSE-best] Elapsed time: ", "rocSPARSE-best-fp16": "[rocSPARSE-best] Elapsed time: "}
FEATURE_DIMS = [256, 512, 1024]        # bench_all.py:18


def run(cmd, env):
    return subprocess.run([sys.executable, *cmd], capture_output=True, text=True, env=env, stdin=subprocess.DEVNULL)


def scrape(method, stdout):
    pat = TIME_PATTERN[method]
    return stdout.split(pat)[1].split(" ms")[0] if pat in stdout else "NAN"


def npz_names(folder):
    return sorted(os.path.splitext(f)[0] for f in os.listdir(folder)
                  if f.endswith(".npz") and "reorder" not in f and os.path.isfile(os.path.join(folder, f)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets_folder", default=os.getenv("DATASET_PATH"))
    ap.add_argument("--synthetic", default=None, help="comma list of NAME[:scale] (synth_graphs) instead of a folder")
    ap.add_argument("--feature_dims", default=",".join(map(str, FEATURE_DIMS)))
    ap.add_argument("--methods", default="hipSPARSE,Voltrix")
    ap.add_argument("--reorder_method", default="auto", choices=["auto", "spectral", "bfs", "rcm"],
                    help="how a missing <name>.reorder.npz is produced")
    ap.add_argument("--output_file", default="results.csv")
    ap.add_argument("--append", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--direct", action="store_true",
                    help="the methods read <name>.npz themselves (bm_voltrix.py / bm_sparse.py --npz) instead of the text dump of "
                         "graph_gen.py: same graph, same protocol, without minutes of np.savetxt / np.loadtxt on a 100 M-edge "
                         "graph.  Method 'Voltrix-fp16' = the same operator handed fp16 features (the reference has fp32 only)")
    ap.add_argument("--skip_reorder", action="store_true", help="no Reorder=Y rows (no <name>.reorder.npz is produced)")
    args = ap.parse_args(argv)
    methods = args.methods.split(",")
    assert all(m in TIME_PATTERN for m in methods), f"methods: {sorted(TIME_PATTERN)}"
    dims = [int(d) for d in args.feature_dims.split(",")]
    env = dict(os.environ)
    work = tempfile.mkdtemp(prefix="bench_all_")
    folder = args.datasets_folder
    if args.synthetic:
        sys.path.insert(0, REPO)
        import synth_graphs

        folder = os.path.join(work, "datasets")
        os.makedirs(folder)
        for item in args.synthetic.split(","):
            name, _, scale = item.partition(":")
            import torch

            ip, ix, _ = synth_graphs.generate(name, scale=float(scale or 1.0),
                                              device="cuda" if torch.cuda.is_available() else "cpu")
            ip, ix = ip.cpu(), ix.cpu()
            torch.cuda.empty_cache() if torch.cuda.is_available() else None
            import scipy.sparse as sp

            n_ = ip.numel() - 1      # a scipy CSR archive (uncompressed): loads without the edge-list -> CSR conversion
            sp.save_npz(os.path.join(folder, f"{name}.npz"),
                        sp.csr_matrix((np.ones(ix.numel(), np.int8), ix.numpy(), ip.numpy()), shape=(n_, n_)), compressed=False)
    assert folder and os.path.isdir(folder), "give --datasets_folder (or DATASET_PATH) or --synthetic"
    if not args.append and os.path.exists(args.output_file):
        os.remove(args.output_file)
    if not os.path.exists(args.output_file):
        with open(args.output_file, "w") as f:
            f.write("Method,Dataset,FeatDim,Reorder,Time (ms)\n")

    def record(method, name, dim, mark, time):
        with open(args.output_file, "a") as f:
            f.write(f"{method},{name},{dim},{mark},{time}\n")
        print(f"{method} {name} F={dim} reorder={mark}: {time} ms", flush=True)

    for name in npz_names(folder) if args.direct else []:
        path = os.path.join(folder, name + ".npz")
        rpath = path[:-4] + ".reorder.npz"
        want_reorder = not args.skip_reorder and any(m.startswith("Voltrix") for m in methods)
        if want_reorder and not os.path.exists(rpath):
            gen = run([os.path.join(HERE, "graph_gen.py"), "--npz", path, "--write_reorder", args.reorder_method, "--no_dump"], env)
            assert gen.returncode == 0, gen.stderr[-2000:]
        for dim in dims:
            if "hipSPARSE" in methods:
                r = run([os.path.join(HERE, "bm_sparse.py"), "--npz", path, "--num_feats", str(dim), "--iters",
                         str(max(args.iters, 10))], env)
                record("hipSPARSE", name, dim, "N", scrape("hipSPARSE", r.stdout))
            for method in [m for m in methods if m.startswith("rocSPARSE-best")]:
                r = run([os.path.join(HERE, "bm_rocsparse.py"), "--npz", path, "--num_feats", str(dim), "--iters",
                         str(max(args.iters, 10)), *(["--half"] if method.endswith("fp16") else [])], env)
                record(method, name, dim, "N", scrape(method, r.stdout))
            for method in [m for m in methods if m.startswith("Voltrix")]:
                fp16 = ["--fp16"] if method == "Voltrix-fp16" else []
                for mark, flag in (("N", []), ("Y", ["--reorder"])) if want_reorder else (("N", []),):
                    r = run([os.path.join(HERE, "bm_voltrix.py"), "--npz", path, "--dataset", name, "--num_feats", str(dim),
                             "--iters", str(args.iters), *fp16, *flag], env)
                    if r.returncode != 0:
                        print(r.stderr[-1500:], file=sys.stderr)
                    record(method, name, dim, mark, scrape(method, r.stdout))
    for name in [] if args.direct else npz_names(folder):
        path = os.path.join(folder, name + ".npz")
        for dim in dims:
            dump = os.path.join(work, "dump")
            gen = run([os.path.join(HERE, "graph_gen.py"), "--npz", path, "--num_feats", str(dim), "--out_dir", dump], env)
            assert gen.returncode == 0, gen.stderr[-2000:]
            if "hipSPARSE" in methods:
                r = run([os.path.join(HERE, "bm_sparse.py"), "--dir", dump], env)
                record("hipSPARSE", name, dim, "N", scrape("hipSPARSE", r.stdout))
            if "rocSPARSE-best" in methods:
                r = run([os.path.join(HERE, "bm_rocsparse.py"), "--dir", dump, "--iters", str(max(args.iters, 10))], env)
                record("rocSPARSE-best", name, dim, "N", scrape("rocSPARSE-best", r.stdout))
            if "Voltrix" in methods:
                r = run([os.path.join(HERE, "bm_voltrix.py"), "--dir", dump, "--dataset", name, "--iters", str(args.iters)], env)
                record("Voltrix", name, dim, "N", scrape("Voltrix", r.stdout))
                flags = [] if os.path.exists(path[:-4] + ".reorder.npz") else ["--write_reorder", args.reorder_method]
                gen = run([os.path.join(HERE, "graph_gen.py"), "--npz", path, *flags, "--reorder", "--num_feats", str(dim),
                           "--out_dir", dump], env)
                assert gen.returncode == 0, gen.stderr[-2000:]
                r = run([os.path.join(HERE, "bm_voltrix.py"), "--dir", dump, "--dataset", name + ".reorder", "--iters",
                         str(args.iters)], env)
                record("Voltrix", name, dim, "Y", scrape("Voltrix", r.stdout))
    print(f"results -> {os.path.abspath(args.output_file)}")


if __name__ == "__main__":
    main()
