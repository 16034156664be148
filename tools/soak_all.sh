#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python tools/soak_forms.py 150 7 2>&1 | tail -8
timeout 600 python tools/soak_batches.py 60 3 2>&1 | tail -5
timeout 600 python tools/soak_fuzz.py 1500 2>&1 | tail -5
