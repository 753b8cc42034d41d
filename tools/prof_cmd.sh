#!/bin/bash
# Profiles an arbitrary command on the GPU box: kernel-trace pass + optional --pmc passes.
# usage: tools/prof_cmd.sh <tag> "<counter set 1>;<counter set 2>;..." <command...>
set -u
TAG=$1; shift
SETS=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/trace_cmd.log 2>&1
i=0
IFS=';' read -ra ARR <<< "$SETS"
for CTRS in "${ARR[@]}"; do
  [ -z "$CTRS" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc$i -o pmc$i -- "$@" > $OUT/pmc${i}_cmd.log 2>&1
done
python tools/prof_summary.py $(find $OUT -name '*.db' | sort) > $OUT/summary.txt 2>&1
grep -h '^{' $OUT/trace_cmd.log >> $OUT/summary.txt
[ -n "${PROF_DB_HOOK:-}" ] && python $PROF_DB_HOOK $(find $OUT/trace -name '*.db' | head -n 1) > $OUT/hook.txt 2>&1
find $OUT -name '*.db' -delete
cat $OUT/summary.txt
