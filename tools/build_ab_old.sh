#!/bin/bash
# Build libw2v2hip.so of an EARLIER commit into tools/ab/lib_ab_old.so (git-ignored, travels with gpurun) for the
# same-box A/B of tools/ab_bench.sh:   bash tools/build_ab_old.sh <commit>    (container only: needs .git)
set -e
C=${1:-HEAD~1}
rm -rf /tmp/abtree; git worktree prune; git worktree add -f /tmp/abtree $C > /dev/null 2>&1
(cd /tmp/abtree && python3 -c "from w2v2_speaker_amd import _build; print(_build.build(force=True))" | tail -1)
cp /tmp/abtree/w2v2_speaker_amd/libw2v2hip.so tools/ab/lib_ab_old.so
git worktree remove --force /tmp/abtree
echo "lib_ab_old.so = $(git rev-parse --short $C)"
