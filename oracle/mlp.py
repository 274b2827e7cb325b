"""Oracle: traversability MLP, loss, confidence statistic and Adam step.  TEST INFRASTRUCTURE ONLY.

Explicit-math restatement (manual backward, no autograd) of
  wild_visual_navigation/model/simple_mlp.py:10-39          (SimpleMLP)
  wild_visual_navigation/utils/loss.py:93-160               (TraversabilityLoss.forward)
  wild_visual_navigation/utils/confidence_generator.py:78-82,182-193  (latest_measurement)
  wild_visual_navigation/traversability_estimator/traversability_estimator.py:100,464-477 (Adam step)
PINNED against the reference's own modules + torch.optim.Adam by oracle/pin_reference.py.
"""
import math
from typing import Dict, Tuple

import torch

KEYS = ("layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias", "layers.4.weight", "layers.4.bias")


def make_mlp_state_dict(D: int, hidden=(256, 32), seed: int = 42) -> Dict[str, torch.Tensor]:
    """torch.nn.Linear default init (kaiming-uniform a=sqrt(5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    for weight and bias) under a seeded generator, state-dict keys as simple_mlp.py:24-30 produces."""
    g = torch.Generator().manual_seed(seed)
    sizes = [D, hidden[0], hidden[1], 1 + D]
    sd = {}
    for li, (i, o) in zip((0, 2, 4), zip(sizes[:-1], sizes[1:])):
        b = 1.0 / math.sqrt(i)
        sd[f"layers.{li}.weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * b
        sd[f"layers.{li}.bias"] = (torch.rand(o, generator=g) * 2 - 1) * b
    return sd


def mlp_forward(sd, x: torch.Tensor, keep: bool = False):
    """simple_mlp.py:33-39: three Linears with ReLU, sigmoid on column 0 (nr_sigmoid_layers = 1)."""
    z1 = x @ sd["layers.0.weight"].T + sd["layers.0.bias"]
    h1 = torch.relu(z1)
    z2 = h1 @ sd["layers.2.weight"].T + sd["layers.2.bias"]
    h2 = torch.relu(z2)
    out = h2 @ sd["layers.4.weight"].T + sd["layers.4.bias"]
    out = torch.cat([torch.sigmoid(out[:, :1]), out[:, 1:]], dim=1)
    return (out, h1, h2) if keep else out


def confidence_from_stats(x: torch.Tensor, mean: float, std: float, std_factor: float) -> torch.Tensor:
    """confidence_generator.py:182-193 (inference_without_update)."""
    shifted = mean + std * std_factor
    lo = max(shifted - std, 0.0) if not math.isnan(std) else float("nan")
    hi = shifted + std
    xc = torch.clip(x, lo, hi)
    return (1 - (xc - lo) / (hi - lo)).float()


def loss_forward(out, x, y, y_valid, std_factor=0.5, w_trav=0.03, w_reco=0.5) -> Dict[str, torch.Tensor]:
    """loss.py:93-160 with anomaly_balanced=True, trav_cross_entropy=False, w_temp*0."""
    R, D = x.shape
    loss_reco = ((out[:, 1:] - x) ** 2).mean(dim=1)
    pos = loss_reco[y_valid]
    mean = pos.mean()
    std = pos.std()  # unbiased; NaN for < 2 positives (confidence_generator.py:80-81)
    conf = confidence_from_stats(loss_reco, float(mean), float(std), std_factor)
    trav_raw = (out[:, 0] - y) ** 2
    weighted = torch.where(y_valid, trav_raw, trav_raw * (1 - conf))
    loss_trav_conf = weighted.sum() / R
    loss_reco_mean = pos.mean()
    loss = w_trav * loss_trav_conf + w_reco * loss_reco_mean
    return {
        "loss": loss,
        "loss_reco": loss_reco_mean,
        "loss_trav": trav_raw.mean(),
        "loss_trav_confidence": loss_trav_conf,
        "confidence": conf,
        "mean": mean,
        "std": std,
        "loss_reco_rows": loss_reco,
    }


def loss_backward(sd, x, y, y_valid, out, h1, h2, conf, w_trav=0.03, w_reco=0.5) -> Dict[str, torch.Tensor]:
    """Manual gradient of loss_forward wrt the six parameter tensors (confidence is a constant:
    it is produced under torch.no_grad in loss.py:105-114)."""
    R, D = x.shape
    nv = int(y_valid.sum())
    wrow = torch.where(y_valid, torch.ones(R), 1 - conf)
    g_out = torch.zeros_like(out)
    s = out[:, 0]
    # d/dz of (sigmoid(z) - y)^2 * wrow * w_trav / R
    g_out[:, 0] = (w_trav / R) * wrow * 2 * (s - y) * s * (1 - s)
    g_out[:, 1:] = (w_reco / (nv * D)) * 2 * (out[:, 1:] - x) * y_valid[:, None].float()
    grads = {}
    grads["layers.4.weight"] = g_out.T @ h2
    grads["layers.4.bias"] = g_out.sum(0)
    g_h2 = (g_out @ sd["layers.4.weight"]) * (h2 > 0)
    grads["layers.2.weight"] = g_h2.T @ h1
    grads["layers.2.bias"] = g_h2.sum(0)
    g_h1 = (g_h2 @ sd["layers.2.weight"]) * (h1 > 0)
    grads["layers.0.weight"] = g_h1.T @ x
    grads["layers.0.bias"] = g_h1.sum(0)
    return grads


def adam_update(p, g, m, v, step: int, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8) -> None:
    """torch.optim.Adam (no amsgrad / weight decay / maximize), single-tensor formulation, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class TrainState:
    """Parameters + Adam moments + step counter + confidence statistic (mean, std)."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.m = {k: torch.zeros_like(v) for k, v in sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in sd.items()}
        self.step = 0
        self.mean = 0.0
        self.std = 1.0


def train_step(st: TrainState, x, y, y_valid, lr=1e-3, std_factor=0.5, w_trav=0.03, w_reco=0.5) -> Dict[str, float]:
    """One TraversabilityEstimator.train() body (traversability_estimator.py:464-477, 491-493)."""
    out, h1, h2 = mlp_forward(st.sd, x, keep=True)
    L = loss_forward(out, x, y, y_valid, std_factor, w_trav, w_reco)
    grads = loss_backward(st.sd, x, y, y_valid, out, h1, h2, L["confidence"], w_trav, w_reco)
    st.step += 1
    for k in KEYS:
        adam_update(st.sd[k], grads[k], st.m[k], st.v[k], st.step, lr=lr)
    st.mean, st.std = float(L["mean"]), float(L["std"])
    return {
        "loss_total": float(L["loss"]),
        "loss_trav": float(L["loss_trav"]),
        "loss_reco": float(L["loss_reco"]),
        "mean": st.mean,
        "std": st.std,
    }


# --------------------------------------------------------------------------------------------------
# Data-parallel decomposition of train_step (what wvn_mlp_train_phase_{a,b,c} compute per rank).
# Summing `stats` and `flat grads` over ranks and then calling phase_c on every rank must reproduce
# train_step on the concatenated batch (tests/test_distributed_gloo.py, tests/test_gpu_mlp.py).
# --------------------------------------------------------------------------------------------------
def flat_of(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    return torch.cat([d[k].reshape(-1) for k in KEYS])


def phase_a_local(sd, x, y_valid):
    out, h1, h2 = mlp_forward(sd, x, keep=True)
    lr = ((out[:, 1:] - x) ** 2).mean(dim=1)
    pos = lr[y_valid].double()
    stats = torch.tensor([float(pos.numel()), float(pos.sum()), float((pos * pos).sum()), float(x.shape[0])],
                         dtype=torch.float64)
    return stats, (out, h1, h2, lr)


def stats_mean_std(stats) -> Tuple[float, float]:
    n, s1, s2 = float(stats[0]), float(stats[1]), float(stats[2])
    mean = s1 / n if n > 0 else float("nan")
    var = (s2 - s1 * s1 / n) / (n - 1.0) if n > 1 else float("nan")
    std = math.sqrt(var) if var == var and var > 0 else (0.0 if var == var else float("nan"))
    return float(torch.tensor(mean, dtype=torch.float32)), float(torch.tensor(std, dtype=torch.float32))


def phase_b_local(sd, x, y, y_valid, cache, stats, std_factor=0.5, w_trav=0.03, w_reco=0.5) -> torch.Tensor:
    out, h1, h2, lr = cache
    R_tot, n_valid = float(stats[3]), float(stats[0])
    D = x.shape[1]
    mean, std = stats_mean_std(stats)
    conf = confidence_from_stats(lr, mean, std, std_factor)
    s = out[:, 0]
    diff = s - y
    raw = diff * diff
    wrow = torch.where(y_valid, torch.ones_like(raw), 1 - conf)
    g_out = torch.zeros_like(out)
    g_out[:, 0] = (w_trav / R_tot) * wrow * 2 * diff * s * (1 - s)
    g_out[:, 1:] = (w_reco / (n_valid * D)) * 2 * (out[:, 1:] - x) * y_valid[:, None].float()
    grads = {}
    grads["layers.4.weight"] = g_out.T @ h2
    grads["layers.4.bias"] = g_out.sum(0)
    g_h2 = (g_out @ sd["layers.4.weight"]) * (h2 > 0)
    grads["layers.2.weight"] = g_h2.T @ h1
    grads["layers.2.bias"] = g_h2.sum(0)
    g_h1 = (g_h2 @ sd["layers.2.weight"]) * (h1 > 0)
    grads["layers.0.weight"] = g_h1.T @ x
    grads["layers.0.bias"] = g_h1.sum(0)
    extra = torch.stack([(raw * wrow).sum(), raw.sum()])
    return torch.cat([flat_of(grads), extra])


def phase_c(st: TrainState, flat_grads: torch.Tensor, stats, lr=1e-3, w_trav=0.03, w_reco=0.5) -> Dict[str, float]:
    st.step += 1
    off = 0
    for k in KEYS:
        n = st.sd[k].numel()
        adam_update(st.sd[k], flat_grads[off:off + n].view_as(st.sd[k]), st.m[k], st.v[k], st.step, lr=lr)
        off += n
    R_tot = float(stats[3])
    reco = float(stats[1]) / float(stats[0])
    st.mean, st.std = stats_mean_std(stats)
    return {"loss_total": w_trav * float(flat_grads[off]) / R_tot + w_reco * reco,
            "loss_trav": float(flat_grads[off + 1]) / R_tot, "loss_reco": reco, "mean": st.mean, "std": st.std}
