cd $GRAFT_REPO_ROOT
PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python scripts/_proj_ab.py 2>&1 | tail -6
