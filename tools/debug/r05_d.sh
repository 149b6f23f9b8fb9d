#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/debug/wf_remote_check.py 2>&1 | tail -20
