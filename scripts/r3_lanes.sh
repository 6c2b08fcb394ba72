#!/bin/bash
run() { echo "== $*"; python scripts/variants.py run --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --steps 300 "$@" 2>&1 | grep default; }
run --streams 2
run --streams 3
run --streams 4
run --streams 2 --inflight 2
run --streams 1 --inflight 2
run --streams 1 --inflight 3
