cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_stego_pixels.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
timeout 120 python scripts/bench_pixel_kmeans.py 2>&1 | tail -1
timeout 300 python bench.py --stego-reading upstream --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('upstream', d['value'], d['ms_per_step'])"
