#!/bin/bash
# usage: profiles/gpu_step.sh <tag> <timeout-s> <command...>   -- one GPU step under its own timeout; output to gpurun_out/<tag>.log
# exit code: the command's; after a timeout / kill (124, 137) a marker file stops the later steps of the same call
TAG=$1; T=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
if [ -e $ROOT/gpurun_out/.stop ]; then echo "[$TAG] skipped: an earlier step timed out"; exit 99; fi
timeout -k 10 $T "$@" > $ROOT/gpurun_out/$TAG.log 2>&1
RC=$?
echo "[$TAG] rc=$RC"; tail -n 6 $ROOT/gpurun_out/$TAG.log
if [ $RC -eq 124 ] || [ $RC -eq 137 ]; then touch $ROOT/gpurun_out/.stop; fi
exit $RC
