cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests -x -q -m "gpu and not slow" 2>&1 | tail -2
