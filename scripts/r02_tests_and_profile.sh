#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/r02_gpu_tests.sh
bash scripts/profile_r02.sh
