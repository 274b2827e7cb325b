"""TraversabilityLoss -- wild_visual_navigation/utils/loss.py:57-164.

``forward`` keeps the reference signature and return triple.  The optimisation step of
``TraversabilityEstimator.train`` does not go through this module's autograd graph: loss, gradient and
confidence statistic are produced by the fused HIP phases (include/wvn_hip.h, wvn_mlp_train_phase_*).
This class is the caller-facing view of the same arithmetic (differentiable torch ops on the device
tensors it is given) and the owner of the ConfidenceGenerator state."""
from typing import Optional

import torch
import torch.nn.functional as F

from .confidence_generator import ConfidenceGenerator
from .data import Data


class TraversabilityLoss(torch.nn.Module):
    def __init__(self, w_trav: float, w_reco: float, w_temp: float, anomaly_balanced: bool, model: torch.nn.Module,
                 method: str, confidence_std_factor: float, log_enabled: bool = False, log_folder: str = "/tmp",
                 trav_cross_entropy=False):
        super().__init__()
        if trav_cross_entropy or not anomaly_balanced:
            raise ValueError("the MI355X path implements the default loss (MSE, anomaly_balanced=True)")
        self._w_trav, self._w_reco, self._w_temp = w_trav, w_reco, w_temp
        self._anomaly_balanced = anomaly_balanced
        self.__dict__["_model"] = model  # not registered as a sub-module (mirrors usage, avoids state-dict dup)
        self._confidence_generator = ConfidenceGenerator(std_factor=confidence_std_factor, method=method,
                                                         log_enabled=log_enabled, log_folder=log_folder)

    def reset(self):
        self._confidence_generator.reset()

    # The reference registers the model as a sub-module of the loss (loss.py:75), so its ``traversability_loss_state_dict``
    # carries ``_model.layers.*`` next to ``_confidence_generator.*``.  Here the model is not a sub-module (one owner for the
    # flat parameter buffer), so the same keys are emitted as aliases on save and ignored on load: checkpoints are
    # interchangeable with the reference in both directions.
    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", "")
        for k, v in self._model.state_dict().items():
            sd[f"{prefix}_model.{k}"] = v
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        own = {k: v for k, v in state_dict.items() if not k.startswith("_model.")}
        return super().load_state_dict(own, strict=strict, **kwargs)

    def forward(self, graph: Optional[Data], res: torch.Tensor, update_generator: bool = True, step: int = 0,
                log_step: bool = False):
        D = graph.x.shape[1]
        loss_reco = F.mse_loss(res[:, -D:], graph.x, reduction="none").mean(dim=1)
        with torch.no_grad():
            if update_generator:
                confidence = self._confidence_generator.update(x=loss_reco, x_positive=loss_reco[graph.y_valid],
                                                               step=step, log_step=log_step)
            else:
                confidence = self._confidence_generator.inference_without_update(x=loss_reco)
        loss_trav_raw = F.mse_loss(res[:, :-D].squeeze(), graph.y[:], reduction="none")
        weighted = torch.where(graph.y_valid, loss_trav_raw, loss_trav_raw * (1 - confidence))
        loss_trav_confidence = weighted.sum() / graph.y.shape[0]
        loss_temp = torch.zeros_like(loss_trav_confidence)
        loss_reco_mean = loss_reco[graph.y_valid].mean()
        loss = self._w_trav * loss_trav_confidence + self._w_reco * loss_reco_mean + self._w_temp * loss_temp
        aux = {"loss_reco": loss_reco_mean, "loss_trav": loss_trav_raw.mean(), "loss_temp": loss_temp.mean(),
               "loss_trav_confidence": loss_trav_confidence, "confidence": confidence}
        return loss, aux, res

    def update_node_confidence(self, node):
        reco_loss = F.mse_loss(node.prediction[:, 1:], node.features, reduction="none").mean(dim=1)
        node.confidence = self._confidence_generator.inference_without_update(reco_loss)


class AnomalyLoss(torch.nn.Module):
    """wild_visual_navigation/utils/loss.py:16-54 belongs to the LinearRnvp anomaly-detection ablation (non-default model,
    SURVEY.md section 2: out of scope).  The name exists because callers import it unconditionally (quick_start.py:12,
    wvn_feature_extractor_node.py:12); constructing it is refused."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise ValueError("AnomalyLoss (LinearRnvp anomaly detection) is outside the MI355X hot path")
