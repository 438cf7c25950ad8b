#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/split_lab.py 1 2 3 4 2>&1 | tail -6
