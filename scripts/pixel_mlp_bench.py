"""Per-pixel traversability inference at BASELINE.json's 448x448 frame size: the fused kernel (csrc/pixel_mlp.hip)
beside the reference-shaped sequence on the same library (dense upsample -> SimpleMLP forward -> confidence), both fed
from patch tokens already in HBM.  Prints one JSON line.  WVN_PIXEL_WSPLIT=0 selects the single-bf16 weight variant."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_visual_navigation_amd import ops  # noqa: E402
from wild_visual_navigation_amd.cfg import ExperimentParams  # noqa: E402
from wild_visual_navigation_amd.model import get_model  # noqa: E402
from wild_visual_navigation_amd.utils import ConfidenceGenerator, Data  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    B, G, H = int(os.environ.get("PIX_B", 16)), 56, 448
    params = ExperimentParams()
    params.model.simple_mlp_cfg.input_size = 384
    model = get_model(params.model).to(dev)
    model.eval()
    cg = ConfidenceGenerator(method="latest_measurement", std_factor=0.5).to(dev)
    cg.mean[0], cg.std[0] = 0.9, 0.25
    tokens = 2.0 * torch.randn(B, G * G, 384, device=dev)
    zx = torch.zeros(B * G * G, 640, dtype=torch.bfloat16, device=dev)
    zx[:, 256:] = tokens.reshape(-1, 384).to(torch.bfloat16)
    model.pack_per_pixel()

    fused_ms = timed(lambda: model.forward_per_pixel(zx, B, G, (H, H), 0.9, 0.25, 0.5, repack=False), 20) / B

    def unfused():  # one frame: wvn_feature_extractor_node.py:319-363 on this library's exact-mode kernels
        dense = ops.upsample_bilinear(tokens[:1], G, H)
        x = dense[0].permute(1, 2, 0).reshape(-1, 384)
        pred = model.forward(Data(x=x))
        lr = ((pred[:, 1:] - x) ** 2).mean(1)
        return pred[:, 0], cg.inference_without_update(lr)

    unfused_ms = timed(unfused, 3)
    tok2d = tokens.reshape(B * G * G, 384)
    exact_ms = timed(lambda: model.forward_per_pixel_exact(tok2d, B, G, (H, H), 0.9, 0.25, 0.5), 10) / B
    # accuracy of both fused forms against the un-fused fp32 sequence on frame 0
    t_ref, c_ref = unfused()
    t_x3, c_x3, _ = model.forward_per_pixel_exact(tok2d[: G * G], 1, G, (H, H), 0.9, 0.25, 0.5)
    t_bf, c_bf, _ = model.forward_per_pixel(zx[: G * G], 1, G, (H, H), 0.9, 0.25, 0.5, repack=False)
    err = lambda a, b: round(float((a.reshape(-1) - b.reshape(-1)).abs().max()), 6)  # noqa: E731
    mfma = 82 if os.environ.get("WVN_PIXEL_WSPLIT", "1") != "0" else 62
    flops = (H * H / 32) * mfma * 32 * 32 * 16 * 2 + G * G * 384 * 256 * 2
    print(json.dumps({"frame": f"{H}x{H}", "grid": G, "batch": B, "fused_ms_per_frame": round(fused_ms, 4),
                      "fused_frames_per_s": round(1e3 / fused_ms, 1), "fused_mfma_tflops": round(flops / fused_ms / 1e9, 1),
                      "unfused_ms_per_frame": round(unfused_ms, 3), "speedup": round(unfused_ms / fused_ms, 1),
                      "exact_fused_ms_per_frame": round(exact_ms, 4), "exact_speedup": round(unfused_ms / exact_ms, 1),
                      "max_abs_err_trav": {"exact_fused": err(t_x3, t_ref), "bf16_fused": err(t_bf, t_ref)},
                      "max_abs_err_conf": {"exact_fused": err(c_x3, c_ref), "bf16_fused": err(c_bf, c_ref)},
                      "weight_split": mfma == 82}))


if __name__ == "__main__":
    main()
