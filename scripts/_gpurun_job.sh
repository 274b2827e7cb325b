cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backbone.py -x -q -k "fused_block or batch_invariance" 2>&1 | tail -4
