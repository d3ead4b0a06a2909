#!/bin/bash
# Runs a list of GPU steps one after the other on the gpurun box, each with its own timeout and log; stops at the
# first step that was KILLED (timeout / signal) -- a killed GPU step must not be followed by another one -- but
# carries on after an ordinary non-zero exit (a failing test).   usage: gpu_session.sh <outdir> "<name>|<secs>|<cmd>" ...
out=$1; shift
mkdir -p "$out"
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${secs}s): $cmd" | tee -a "$out/session.log"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - t0 ))s" | tee -a "$out/session.log"
  if [ $rc -ge 124 ]; then echo "=== step killed: stopping the session" | tee -a "$out/session.log"; exit $rc; fi
done
exit 0
