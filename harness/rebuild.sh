#!/bin/bash
# Rebuild everything that travels to the GPU box after a header change: the ahead-of-time library and the prebuilt JIT
# kernels (their cache key hashes the include tree); cache entries of older header versions are dropped.
#   usage: harness/rebuild.sh        (from anywhere)
set -e -o pipefail
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
# a failed library build must stop here (the prebuild below would otherwise prune the JIT cache against a stale library)
if ! make -s -C "$REPO/voltrix-spmm_amd/csrc" -j6 > /tmp/voltrix_rebuild.log 2>&1; then
  grep -E "error|Error" /tmp/voltrix_rebuild.log || tail -20 /tmp/voltrix_rebuild.log
  echo "library build FAILED" >&2
  exit 1
fi
(cd "$REPO" && VOLTRIX_PREBUILD_PRUNE=1 python -c "import __graft_entry__ as g; g.build()") | tail -1
echo "jit cache entries: $(ls "$REPO/voltrix-spmm_amd/.jit_cache/cache" | wc -l)"
