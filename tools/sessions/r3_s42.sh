#!/bin/bash
# session 42: whole 1080p frames of scenes without lights on the per-wave resident kernel — the GPU suite, the profile set of C and D again
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest42.log 2>&1; grep -E "^FAILED" $OUT/pytest42.log | cut -c 1-160 | head; tail -1 $OUT/pytest42.log
bash tools/round_profiles.sh C D || exit 1
