#!/bin/bash
# Longer randomized runs than the suite holds (dev aid; GPU): kernel forms, large batches, call patterns.
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python tests/soak/soak_forms.py ${1:-600} 11 2>&1 | tail -6
timeout 900 python tests/soak/soak_batches.py ${2:-400} 5 2>&1 | tail -4
timeout 900 python tests/soak/soak_fuzz.py ${3:-5000} 2>&1 | tail -4
