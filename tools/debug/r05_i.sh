#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug/replay_diff.py 3000 2>&1 | tail -8
timeout 600 python tools/debug/replay_diff.py 1000 2>&1 | tail -8
