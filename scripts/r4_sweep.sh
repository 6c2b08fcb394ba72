#!/bin/bash
# usage: scripts/r4_sweep.sh <dims> <var> v1 v2 ...: stage A times of scripts/r4_stage_a.py under an environment knob
dims=$1; var=$2; shift 2
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v python3 scripts/r4_stage_a.py $dims mixed 100 p 2>&1 | grep "stage_a=1" | sed 's/moffat.*mf_prep [0-9.]* //'
done
