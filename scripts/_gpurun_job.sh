cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_stego_pixels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15
