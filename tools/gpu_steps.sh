#!/bin/bash
# usage: bash tools/gpu_steps.sh <tag> "<seconds> <command>" ...    (on the GPU box, through gpurun)
# Runs the steps one after the other, each under its own `timeout -k 10`, logging to gpurun_out/<tag>_<i>.log.  A step that FAILS
# (assertion, non-zero exit) does not stop the following ones; a step that is KILLED at its limit (124 / 137) does: after a hang
# nothing else is started on the box.
TAG=$1; shift
mkdir -p gpurun_out
i=0
rc_all=0
for spec in "$@"; do
    i=$((i + 1))
    secs=${spec%% *}; cmd=${spec#* }
    log=gpurun_out/${TAG}_${i}.log
    echo "[$TAG step $i] timeout ${secs}s: $cmd" | tee $log
    timeout -k 10 $secs bash -c "$cmd" >> $log 2>&1
    rc=$?
    echo "[$TAG step $i] exit $rc" | tee -a $log
    tail -n 6 $log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$TAG] step $i was killed at its limit: stopping"; exit $rc; fi
    [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
