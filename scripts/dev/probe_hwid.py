import os, sys
os.environ["WVN_A384_STAGGER"] = "-1"
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.backbone import pack_a384_mx
dev = torch.device("cuda:0")
M, F = 128 * 3152, 1536
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, 384, generator=g) * 1.5).to(dev)
st = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)], -1).contiguous()
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
w1p = pack_a384_mx((torch.randn(F, 384, generator=g) * 0.05).to(dev)); b1 = torch.zeros(F, device=dev)
Mp = (M + 31) // 32 * 32
hid = torch.zeros(Mp * F * 3, dtype=torch.uint8, device=dev)
dbg = torch.zeros(512 * 4 * 4, dtype=torch.int64, device=dev)
for _ in range(2):
    _lib.check(lib.wvn_debug_mlp_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), w1p.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.data_ptr() + Mp * F * 2,
                                    0, 0, 0, 0, M, F, dbg.data_ptr(), 0, _lib.stream()), "fc1")
torch.cuda.synchronize()
d = dbg.reshape(512, 4, 4).cpu()
hw = d[:, :, 0]
slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xcc = d[:, :, 1]
t0 = d[:, :, 2] - d[:, :, 2].min()
from collections import Counter
print("wave slots:", Counter(slot.reshape(-1).tolist()))
key = (xcc * 10000 + se * 1000 + sh * 100 + cu)[:, 0]
per_cu = Counter(key.tolist())
print("workgroups per physical CU:", Counter(per_cu.values()))
for b in range(6):
    print(b, "xcc", xcc[b, 0].item(), "se", se[b, 0].item(), "sh", sh[b, 0].item(), "cu", cu[b, 0].item(), "simd", simd[b].tolist(), "slot", slot[b].tolist(), "t0", t0[b].tolist())
same = [i for i in range(512) if key[i] == key[0]]
print("blocks on block 0's CU:", same, [slot[i].tolist() for i in same])
