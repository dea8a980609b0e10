#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for m in 3 0; do timeout 600 python tools/debug/self_consistency3d.py 150 $m 2>&1 | tail -4; done
timeout 600 python tools/debug/self_consistency3d.py 150 3 lits 2>&1 | tail -4
