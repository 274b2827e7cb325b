cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_lib_abi.py -x -q 2>&1 | tail -4
PYTHONPATH=$GRAFT_REPO_ROOT timeout 600 python scripts/small_batch_latency.py 2>&1 | tail -12
