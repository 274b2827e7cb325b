"""TraversabilityEstimator -- the online learner of
wild_visual_navigation/traversability_estimator/traversability_estimator.py:33-505, hot-path slice:
constructor (10 reference arguments), ``train() -> dict``, ``make_batch``, ``add_mission_node``, ``add_supervision_node``
(projection + polygon fill + fmin + label pooling on HIP kernels), ``change_device``, ``save_checkpoint`` /
``load_checkpoint`` (same dict keys, interchangeable with the reference), and the private attributes the learning node reads
(``_model``, ``_traversability_loss._confidence_generator``, ``_mission_graph``, ``_step``, ``_visualizer``).  Forward, loss, backward and Adam run in the fused HIP phases
(trainer.py); with torch.distributed initialised every rank trains on its own frames and the two
small all-reduces keep the replicas identical.
"""
import os
from threading import Lock

import torch

from ..distributed import all_ranks_ready
from ..model import get_model
from ..utils import Batch, TraversabilityLoss
from .. import ops
from .graphs import BaseGraph, DistanceWindowGraph, MaxElementsGraph
from .nodes import MissionNode, SupervisionNode
from .trainer import MlpTrainer


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class TraversabilityEstimator:
    def __init__(self, params, device: str = "cuda", max_distance: float = 3, image_distance_thr: float = None,
                 supervision_distance_thr: float = None, min_samples_for_training: int = 10,
                 vis_node_index: int = 10, mode=False, extraction_store_folder=None,
                 anomaly_detection: bool = False):
        if anomaly_detection:
            raise ValueError("anomaly_detection (LinearRnvp / AnomalyLoss) is outside the MI355X hot path")
        self._device = torch.device(device)
        self._mode = mode
        self._extraction_store_folder = extraction_store_folder
        self._min_samples_for_training = min_samples_for_training
        self._vis_node_index = vis_node_index
        self._params = params
        self._anomaly_detection = anomaly_detection
        # Local graphs (traversability_estimator.py:55-63): supervision nodes within max_distance of the newest one; every
        # mission node farther than image_distance_thr from its predecessor
        self._supervision_graph = DistanceWindowGraph(max_distance=max_distance, edge_distance=supervision_distance_thr)
        if str(mode).endswith("EXTRACT_LABELS"):
            self._mission_graph = MaxElementsGraph(edge_distance=image_distance_thr, max_elements=200)
        else:
            self._mission_graph = BaseGraph(edge_distance=image_distance_thr)
        self._vis_mission_node = None
        self._visualizer = None   # LearningVisualizer (matplotlib overlays) is outside the MI355X path; the attribute exists
        self._learning_lock = Lock()
        self._pause_training = False
        self._pause_mission_graph = False
        self._pause_supervision_graph = False
        self._debug_info_node_count = 0

        torch.manual_seed(42)  # seed_everything(42), traversability_estimator.py:78
        self._model = get_model(_get(params, "model")).to(self._device)
        self._model.train()
        loss_cfg, gen = _get(params, "loss"), _get(params, "general")
        self._traversability_loss = TraversabilityLoss(
            w_trav=_get(loss_cfg, "w_trav"), w_reco=_get(loss_cfg, "w_reco"), w_temp=_get(loss_cfg, "w_temp"),
            anomaly_balanced=_get(loss_cfg, "anomaly_balanced"), model=self._model, method=_get(loss_cfg, "method"),
            confidence_std_factor=_get(loss_cfg, "confidence_std_factor"),
            trav_cross_entropy=_get(loss_cfg, "trav_cross_entropy"), log_enabled=_get(gen, "log_confidence"),
            log_folder=_get(gen, "model_path") or "/tmp").to(self._device)
        self._optimizer = MlpTrainer(self._model, lr=_get(_get(params, "optimizer"), "lr"),
                                     std_factor=_get(loss_cfg, "confidence_std_factor"),
                                     w_trav=_get(loss_cfg, "w_trav"), w_reco=_get(loss_cfg, "w_reco"))
        self._loss = torch.tensor([torch.inf])
        self._step = 0

    # ------------------------------------------------------------------------------------------------
    @property
    def loss(self):
        return float(self._loss.detach().reshape(-1)[0].item())

    @property
    def step(self):
        return self._step

    @property
    def pause_learning(self):
        return self._pause_training

    @pause_learning.setter
    def pause_learning(self, pause: bool):
        self._pause_training = pause

    def change_device(self, device: str):
        """traversability_estimator.py:139-151: move graphs and model to another device (another GPU)."""
        self._device = torch.device(device)
        self._supervision_graph.change_device(device)
        self._mission_graph.change_device(device)
        self._model = self._model.to(device)
        self._traversability_loss = self._traversability_loss.to(device)

    def update_visualization_node(self):
        """traversability_estimator.py:153-163."""
        if self._mission_graph.get_num_nodes() <= self._vis_node_index:
            self._vis_mission_node = self._mission_graph.get_nodes()[0]
        else:
            self._vis_mission_node = self._mission_graph.get_nodes()[-self._vis_node_index]

    def add_mission_node(self, node: MissionNode, verbose: bool = False) -> bool:
        """traversability_estimator.py:166-196: graph insertion (rejected when closer than image_distance_thr to the last
        node), a fresh all-NaN supervision mask, first label pooling."""
        if self._pause_mission_graph:
            return False
        success = self._mission_graph.add_node(node)
        if success and node.use_for_training:
            if verbose:
                print(f"adding node [{node}], total nodes [{self._mission_graph.get_num_nodes()}]")
            h, w = node.feature_segments.shape[0], node.feature_segments.shape[1]
            node.supervision_mask = torch.full((3, h, w), float("nan"), dtype=torch.float32, device=self._device)
            node.update_supervision_signal()
            return True
        return False

    @torch.no_grad()
    def add_supervision_node(self, pnode: SupervisionNode) -> bool:
        """traversability_estimator.py:198-300.  A new robot pose with its traversability score: chain it into the
        supervision graph, build the footprint quadrilateral between it and the previous pose, and for every mission node
        within ``max_distance`` along the mission graph project that polygon into the node's camera, fill it with
        ``colour * traversability``, merge it into the node's supervision mask with fmin and re-pool the per-segment labels.
        Projection + fill + fmin for all nodes is ONE HIP launch (csrc/supervision.hip) that updates the masks in place --
        the reference stacks the masks into a [B,3,H,W] tensor, renders a second one and calls torch.fmin -- and the label
        pooling of all nodes is one batched launch pair."""
        if self._pause_supervision_graph or not pnode.is_valid():
            return False
        last_pnode = self._supervision_graph.get_last_node()
        success = self._supervision_graph.add_node(pnode)
        if not success:
            if last_pnode is not None:   # too close to the last pose: only its score is updated (pessimistically)
                last_pnode.update_traversability(pnode.traversability, pnode.traversability_var)
            return False
        if last_pnode is None or not last_pnode.is_valid():
            return False
        footprint = pnode.make_footprint_with_node(last_pnode)           # [N,3] world points
        last_mission_node = self._mission_graph.get_last_node()
        if last_mission_node is None or getattr(last_mission_node, "supervision_mask", None) is None:
            return False
        for node in list(self._mission_graph._nodes.keys())[self._debug_info_node_count:]:   # :238-246
            if last_mission_node.timestamp - node.timestamp > 30:
                node.clear_debug_data()
                self._debug_info_node_count += 1
            else:
                break
        mission_nodes = self._mission_graph.get_nodes_within_radius_range(
            last_mission_node, 0, self._supervision_graph.max_distance)
        if len(mission_nodes) < 1:
            return False
        shape = last_mission_node.supervision_mask.shape
        masks = []
        for m in mission_nodes:
            if getattr(m, "supervision_mask", None) is None:   # the reference substitutes zeros (:262-264, :272-273)
                m.supervision_mask = torch.zeros(shape, dtype=torch.float32, device=self._device)
            elif not m.supervision_mask.is_contiguous() or m.supervision_mask.dtype != torch.float32:
                m.supervision_mask = m.supervision_mask.float().contiguous()
            masks.append(m.supervision_mask)
        Ks = [m.image_projector.camera.intrinsics[0] for m in mission_nodes]
        poses = [m.pose_cam_in_world for m in mission_nodes]
        trav = pnode.traversability
        value = trav.to(self._device).float().reshape(-1)[:1] if isinstance(trav, torch.Tensor) else float(trav)
        ops.project_render_fmin(Ks, poses, masks, footprint.to(self._device), value)    # colour = 1 (:258)
        # re-pool the labels (:287-289).  A node without features / segments (e.g. added with use_for_training=False before its
        # features arrived) had its mask merged above but has nothing to pool: update_supervision_signal returns early for
        # it in the reference too (nodes.py:401-402)
        poolable = [m for m in mission_nodes if m.features is not None and m.feature_segments is not None]
        pooled = ops.label_pool_batched([m.supervision_mask for m in poolable], [m.segments_i32() for m in poolable],
                                        [m.num_segments() for m in poolable]) if poolable else []
        for m, (sig, val) in zip(poolable, pooled):
            m._supervision_signal, m._supervision_signal_valid = sig, val
        for m in mission_nodes:
            if str(self._mode).endswith("EXTRACT_LABELS") and self._extraction_store_folder is not None:
                p = os.path.join(self._extraction_store_folder, "supervision_mask", str(m.timestamp).replace(".", "_") + ".pt")
                os.makedirs(os.path.dirname(p), exist_ok=True)
                torch.save(torch.nan_to_num(m.supervision_mask.nanmean(axis=0)) != 0, p)
        return True

    def get_mission_nodes(self):
        return self._mission_graph.get_nodes()

    def get_supervision_nodes(self):
        return self._supervision_graph.get_nodes()

    def get_last_valid_mission_node(self):
        last = None
        for node in self._mission_graph.get_nodes():
            if node.is_valid():
                last = node
        return last

    def get_mission_node_for_visualization(self):
        return self._vis_mission_node

    def update_supervision(self, node: MissionNode, mask: torch.Tensor) -> None:
        """Merge an already rendered footprint mask into one node (torch.fmin, traversability_estimator.py:281-286) and
        re-pool its labels (:287-289) -- the per-node tail of ``add_supervision_node`` for callers that render elsewhere."""
        if node.supervision_mask is None:
            node.supervision_mask = mask.clone()
        else:
            node.supervision_mask = torch.fmin(node.supervision_mask, mask)
        node.update_supervision_signal()

    def make_batch(self, batch_size: int = 8):
        nodes = self._mission_graph.get_n_random_valid_nodes(n=batch_size)
        return Batch.from_data_list([n.as_pyg_data() for n in nodes])

    def train(self):
        """traversability_estimator.py:449-497: one optimisation step; returns the reference's dict."""
        if self._pause_training:
            all_ranks_ready(False, self._device)   # a paused replica still answers the others' readiness poll
            return {}
        num_valid_nodes = self._mission_graph.get_num_valid_nodes()
        return_dict = {"mission_graph_num_valid_node": num_valid_nodes}
        graph = None
        if num_valid_nodes > self._min_samples_for_training:
            graph = self.make_batch(_get(_get(self._params, "ablation_data_module"), "batch_size"))
        # Data-parallel replicas: the step contains two collectives, so whether it runs is itself decided collectively --
        # every rank steps iff ALL ranks have a batch (a rank that skipped alone would leave the others hanging in RCCL).
        if all_ranks_ready(graph is not None, self._device):
            if graph is not None:
                with self._learning_lock:
                    losses = self.train_on_batch(graph.x, graph.y, graph.y_valid)
                log_step = (self._step % 20) == 0
                vals = losses.tolist()  # the one host sync per step (the reference does three .item() calls)
                if log_step:
                    print(f"step: {self._step} | loss: {vals[0]:5f} | loss_trav: {vals[1]:5f} | loss_reco: {vals[2]:5f}")
                self._step += 1
                return_dict["loss_total"], return_dict["loss_trav"], return_dict["loss_reco"] = vals[0], vals[1], vals[2]
                return return_dict
        return_dict["loss_total"] = -1
        return return_dict

    def train_on_batch(self, x: torch.Tensor, y: torch.Tensor, y_valid: torch.Tensor) -> torch.Tensor:
        """Fused forward + loss + backward + Adam on this rank's rows; updates the confidence statistic.
        Returns the device tensor {total, trav, reco, conf_mean, conf_std} without synchronising."""
        losses = self._optimizer.train_step(x.to(self._device), y.to(self._device), y_valid.to(self._device))
        cg = self._traversability_loss._confidence_generator
        with torch.no_grad():
            cg.mean.copy_(losses[3:4])
            cg.std.copy_(losses[4:5])
        self._loss = losses[0:1]
        return losses

    # ------------------------------------------------------------------------------------------------
    def save_checkpoint(self, mission_path: str, checkpoint_name: str = "last_checkpoint.pt"):
        with self._learning_lock:
            self._pause_training = True
            os.makedirs(mission_path, exist_ok=True)
            checkpoint_file = os.path.join(mission_path, checkpoint_name)
            torch.save({
                "step": self._step,
                "model_state_dict": self._model.state_dict(),
                "optimizer_state_dict": self._optimizer.optimizer_state_dict(),
                "traversability_loss_state_dict": self._traversability_loss.state_dict(),
                "loss": self.loss,
            }, checkpoint_file)
            self._pause_training = False
        return checkpoint_file

    def load_checkpoint(self, checkpoint_path: str):
        with self._learning_lock:
            self._pause_training = True
            ck = torch.load(checkpoint_path, map_location=self._device, weights_only=False)
            self._model.load_state_dict(ck["model_state_dict"])
            self._optimizer.load_optimizer_state_dict(ck["optimizer_state_dict"])
            self._traversability_loss.load_state_dict(ck["traversability_loss_state_dict"])
            self._step = ck["step"]
            self._loss = torch.tensor([ck["loss"]])
            self._model.train()
            self._pause_training = False
