#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "1024 2" "512 2" "256 2" "1024 4" "512 4" "256 4" "256 8" "128 8" "512 8" "64 16"; do
  set -- $cfg
  echo "== cap=$1 groups/lane=$2"; CGO_GRID_SMALL=$1 CGO_GROUPS_PER_LANE=$2 python3 scripts/latency_check.py 2>&1 | grep "prof=False"
done
