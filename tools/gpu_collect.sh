#!/bin/bash
# LOCAL wrapper of the end-of-round evidence collection: refuses a dirty tree, stamps `git rev-parse HEAD` into
# every profile (the GPU box has no .git), runs tools/collect_round.sh on an MI355X through gpurun, then copies the summaries into
# profiles/<round>_*.   usage: RD_ROUND=r06 bash tools/gpu_collect.sh [extra gpurun --timeout seconds]
set -e
cd "$(dirname "$0")/.."
RD_ROUND=${RD_ROUND:-r06}
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then
    echo "tools/gpu_collect.sh: the working tree has uncommitted changes -- commit first: profiles must describe a commit" >&2
    git status --short --untracked-files=no >&2
    exit 2
fi
HEAD=$(git rev-parse --short=12 HEAD)
echo "$HEAD" > .collect_head          # travels with the snapshot (git-ignored)
/usr/local/graft/bin/gpurun --timeout ${1:-3300} -- "RD_ROUND=$RD_ROUND RD_HEAD=$HEAD bash tools/collect_round.sh" 2>&1 | tail -60
O=gpurun_out/$RD_ROUND
for f in $O/*.txt $O/*.json; do
    [ -f "$f" ] || continue
    case "$(basename $f)" in pytest.txt|collect_profiles.log) continue;; esac
    cp "$f" profiles/${RD_ROUND}_$(basename $f)
done
grep -E "passed|failed|error" $O/pytest.txt | tail -2
echo "collected at $HEAD -> profiles/${RD_ROUND}_*"
