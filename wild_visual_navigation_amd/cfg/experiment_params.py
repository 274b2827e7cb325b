"""ExperimentParams -- the slice of wild_visual_navigation/cfg/experiment_params.py:13-180 the hot path
reads (general, optimizer, loss, ablation_data_module.batch_size, model).  Plain dataclasses that
support both attribute and [] access, as the reference's OmegaConf-wrapped tree is used both ways
(traversability_estimator.py:80,85-96,100,462)."""
from dataclasses import dataclass, field
from typing import List, Optional


class _Node:
    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def keys(self):
        return [k for k in vars(self) if not k.startswith("_")]

    def __iter__(self):
        return iter(self.keys())


@dataclass
class GeneralParams(_Node):
    name: str = "debug/debug"
    model_path: Optional[str] = None
    log_confidence: bool = False
    use_threshold: bool = True


@dataclass
class OptimizerParams(_Node):
    name: str = "ADAM"
    lr: float = 0.001


@dataclass
class LossParams(_Node):
    anomaly_balanced: bool = True
    w_trav: float = 0.03
    w_reco: float = 0.5
    w_temp: float = 0.0
    method: str = "latest_measurement"
    confidence_std_factor: float = 0.5
    trav_cross_entropy: bool = False


@dataclass
class AblationDataModuleParams(_Node):
    batch_size: int = 8


@dataclass
class SimpleMlpCfgParams(_Node):
    input_size: int = 90  # 90 for stego, 384 for dino
    hidden_sizes: List[int] = field(default_factory=lambda: [256, 32, 1])
    reconstruction: bool = True


@dataclass
class OtherModelCfgParams(_Node):
    """Config slots of the non-default models (DoubleMLP / SimpleGCN / LinearRnvp, experiment_params.py:113-139): callers write
    ``input_size`` into all four unconditionally (quick_start.py:131-134); the models themselves are outside this build."""
    input_size: int = 384


@dataclass
class ModelParams(_Node):
    name: str = "SimpleMLP"
    load_ckpt: Optional[str] = None
    simple_mlp_cfg: SimpleMlpCfgParams = field(default_factory=SimpleMlpCfgParams)
    double_mlp_cfg: OtherModelCfgParams = field(default_factory=OtherModelCfgParams)
    simple_gcn_cfg: OtherModelCfgParams = field(default_factory=OtherModelCfgParams)
    linear_rnvp_cfg: OtherModelCfgParams = field(default_factory=OtherModelCfgParams)


@dataclass
class LossAnomalyParams(_Node):
    method: str = "latest_measurement"
    confidence_std_factor: float = 0.5


@dataclass
class ExperimentParams(_Node):
    general: GeneralParams = field(default_factory=GeneralParams)
    optimizer: OptimizerParams = field(default_factory=OptimizerParams)
    loss: LossParams = field(default_factory=LossParams)
    loss_anomaly: LossAnomalyParams = field(default_factory=LossAnomalyParams)
    ablation_data_module: AblationDataModuleParams = field(default_factory=AblationDataModuleParams)
    model: ModelParams = field(default_factory=ModelParams)
