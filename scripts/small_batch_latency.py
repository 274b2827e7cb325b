import time, torch
from wild_visual_navigation_amd.backbone import VitBackbone, synthetic_vit_state_dict
dev = torch.device("cuda:0")
sd = synthetic_vit_state_dict(depth=12, pretrain_grid=28)
for S in (448, 224):
    for B in (1, 2, 4, 8, 16, 32):
        img = torch.rand(B, 3, S, S, device=dev)
        res = {}
        for name, kw in (("fused", {}), ("no-mlp", dict(fuse_mlp=False)), ("no-qkv", dict(fuse_qkv=False)), ("unfused", dict(fuse_mlp=False, fuse_qkv=False))):
            bb = VitBackbone(sd, S, 8, 6, device=dev, precision="bf16", max_chunk=64, **kw)
            for _ in range(3):
                bb.forward_tokens(img)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                bb.forward_tokens(img)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e3
        print(S, B, {k: round(v, 3) for k, v in res.items()})
