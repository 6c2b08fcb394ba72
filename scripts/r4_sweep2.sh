#!/bin/bash
# usage: scripts/r4_sweep2.sh <dims> "<VAR=a VAR2=b>" ... : stage A times under sets of environment knobs
dims=$1; shift
for v in "$@"; do
  echo "== $v"
  env $v python3 scripts/r4_stage_a.py $dims mixed 100 p 2>&1 | grep "stage_a=1" | sed 's/moffat.*mf_prep [0-9.]* //'
done
