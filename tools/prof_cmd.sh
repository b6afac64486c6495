#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, summary -> gpurun_out/<tag>/<name>_kernel_stats.md (+ the rocpd database kept
# for timeline scripts when KEEP_DB=1).   usage (repo root, via gpurun): bash tools/prof_cmd.sh <tag> <name> <cmd...>
TAG=$1; NAME=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$NAME -o $NAME -- "$@" > $OUT/rocprof_$NAME.log 2>&1); echo "rocprof $NAME exit $?"
db=$(find $OUT/prof_$NAME -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $OUT/${NAME}_kernel_stats.md "$*" > /dev/null
if [ "$KEEP_DB" == "1" ]; then cp $db $OUT/$NAME.db; fi
rm -rf $OUT/prof_$NAME
tail -4 $OUT/rocprof_$NAME.log
