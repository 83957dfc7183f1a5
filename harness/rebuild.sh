#!/bin/bash
# Rebuild everything that travels to the GPU box after a header change: the ahead-of-time library and the prebuilt JIT
# kernels (their cache key hashes the include tree); cache entries of older header versions are dropped.
#   usage: harness/rebuild.sh        (from anywhere)
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
make -s -C "$REPO/voltrix-spmm_amd/csrc" -j6 2>&1 | grep -E "error|Error" || true
(cd "$REPO" && VOLTRIX_PREBUILD_PRUNE=1 python -c "import __graft_entry__ as g; g.build()") | tail -1
echo "jit cache entries: $(ls "$REPO/voltrix-spmm_amd/.jit_cache/cache" | wc -l)"
