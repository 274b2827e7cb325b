cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stego_pixels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
bash scripts/ab_lib.sh "b64 libwvn_hip.so" "b62 libwvn_hip.so --batch 62 --chunk 62" "b60 libwvn_hip.so --batch 60 --chunk 60" "b64_again libwvn_hip.so" 2>&1 | tee gpurun_out/ab_batch.log
timeout 300 python bench.py --stego-reading upstream --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('upstream', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_up4 -o up -- python $GRAFT_REPO_ROOT/bench.py --stego-reading upstream --no-cpu-baseline --no-overlap --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_up4.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/summarize_profile.py db gpurun_out/prof_up4/up_results.db 2>&1 | head -12
