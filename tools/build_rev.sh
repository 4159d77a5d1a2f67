#!/bin/bash
# build libge2e_hip.so of another git revision into speaker_embedding_ge2e_loss_amd/<name> (same-box A/B runs:
# tools/ab_bench.sh picks it up through GE2E_HIP_LIB); usage: tools/build_rev.sh <rev> <libname.so>
set -e
rev=$1; out=$2
tmp=$(mktemp -d)
git worktree add -f "$tmp" "$rev" > /dev/null 2>&1
(cd "$tmp" && python -c "
from speaker_embedding_ge2e_loss_amd import build
build.build(force=True, verbose=False)" > /dev/null 2>&1)
cp "$tmp/speaker_embedding_ge2e_loss_amd/libge2e_hip.so" "speaker_embedding_ge2e_loss_amd/$out"
git worktree remove --force "$tmp"; git worktree prune
ls -la "speaker_embedding_ge2e_loss_amd/$out"
