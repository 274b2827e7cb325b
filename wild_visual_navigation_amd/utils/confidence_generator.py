"""ConfidenceGenerator (method "latest_measurement") --
wild_visual_navigation/utils/confidence_generator.py:13-212.

State = three non-trainable parameters mean[1], var[1,1], std[1] with the reference's names, so
``state_dict()`` / the ``.tmp_state_dict.pt`` hand-off (wvn_learning_node.py:381-394) stay compatible.
Inside ``TraversabilityEstimator.train`` the statistic is produced by the fused HIP kernels
(mlp.hip); the methods here are the caller-facing API (quick_start.py:207-210) on small vectors."""
import torch


class ConfidenceGenerator(torch.nn.Module):
    def __init__(self, std_factor, method="latest_measurement", log_enabled: bool = False, log_folder: str = "/tmp"):
        super().__init__()
        if method != "latest_measurement":
            raise ValueError("Unknown method (the MI355X path implements the default 'latest_measurement')")
        self.std_factor = std_factor
        self.log_enabled = log_enabled
        self.log_folder = log_folder
        self.mean = torch.nn.Parameter(torch.zeros(1, dtype=torch.float32), requires_grad=False)
        self.var = torch.nn.Parameter(torch.ones((1, 1), dtype=torch.float32), requires_grad=False)
        self.std = torch.nn.Parameter(torch.ones(1, dtype=torch.float32), requires_grad=False)

    @torch.no_grad()
    def update(self, x: torch.Tensor, x_positive: torch.Tensor, step: int = 0, log_step: bool = False):
        self.mean[0] = x_positive.mean()
        self.std[0] = x_positive.std()
        return self.inference_without_update(x)

    @torch.no_grad()
    def inference_without_update(self, x: torch.Tensor):
        if x.device != self.mean.device:
            return torch.zeros_like(x)
        shifted = self.mean + self.std * self.std_factor
        lo = torch.where(torch.isnan(shifted - self.std), shifted - self.std,
                         torch.clamp(shifted - self.std, min=0.0))
        hi = shifted + self.std
        xc = torch.minimum(torch.maximum(x, lo), hi)
        return (1 - ((xc - lo) / (hi - lo))).type(torch.float32)

    def forward(self, x: torch.Tensor):
        return self.inference_without_update(x)

    def reset(self):
        with torch.no_grad():
            self.mean[0] = 0
            self.var[0] = 1
            self.std[0] = 1

    def get_dict(self):
        return {"mean": self.mean, "var": self.var, "std": self.std}
