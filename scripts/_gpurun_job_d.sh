cd $GRAFT_REPO_ROOT
for F in 0 1 0 1; do
WVN_KMEANS_ASSIGN_FORM=$F timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('assign form $F:', d['value'], d['ms_per_step'])"
done
python - <<'PY'
import ctypes, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
sys.argv=[sys.argv[0]]
args=bench.parse()
dev=torch.device("cuda:0")
fe, model, trainer = bench.make_pipeline(args, dev, "fp16", "upstream")
from wild_visual_navigation_amd import _lib
img=torch.rand(16,3,448,448,generator=torch.Generator().manual_seed(1)).to(dev)
st=(ctypes.c_ulonglong*2)()
_lib.lib().wvn_debug_kmeans_screen_stats(st,1)
_lib.lib().wvn_debug_kmeans_assign_form(3)
fe.extract_batch(img)
torch.cuda.synchronize()
_lib.lib().wvn_debug_kmeans_screen_stats(st,1)
print("bench data (synthetic weights): exact row groups", st[0], "of", st[1], f"= {100.0*st[0]/max(st[1],1):.2f} %")
PY
