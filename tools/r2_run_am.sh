#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "non_finite" 2>&1 | tail -25 | cut -c1-300
