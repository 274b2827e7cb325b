import torch, ctypes, sys
from wild_visual_navigation_amd import _lib, ops
sys.path.insert(0, "tests")
dev = torch.device("cuda:0")
from test_gpu_mlp_fused import make, unfused
for var in (0, 8):
    _lib.lib().wvn_debug_mlp_fused_var(var)
    for M in (128, 300, 515, 1000, 128 * 300 + 17, 128 * 520 + 77):
        xn, w1, b1, w2, b2, x, _ = make(M, 1536, dev)
        want = unfused(xn, w1, b1, w2, b2, x)
        w2p = w2[:, ops.vt_token_order(1536, device=dev)].contiguous()
        nbad = []
        for rep in range(3):
            got = ops.mlp_fused(xn, w1, b1, w2p, b2, x.clone())
            torch.cuda.synchronize()
            nbad.append(((got - want).abs() > 2e-3).sum().item())
        print("var", var, "M", M, "bad", nbad)
