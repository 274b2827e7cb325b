"""Pin the oracle against the reference's OWN code and write the golden vectors.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python -m oracle.pin_reference            # checks + (re)writes tests/golden/*.pt

It imports the reference's first-party modules unmodified from /root/reference with the
unrelated, missing third-party imports (torchvision, kornia, omegaconf, stego, pytictac,
pytorch_lightning, liegroups, cv2, ...) stubbed out, runs them on seeded inputs (and on the
reference's own fixture assets/graph/graph.pt), asserts that oracle/ reproduces them, and stores
inputs + reference outputs as small fixtures so the pin travels with the repo.
"""
import importlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _passthrough_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def import_reference():
    """Make ``wild_visual_navigation`` importable from /root/reference with stubs."""
    for name in [
        "torchvision", "torchvision.transforms", "torchvision.models", "torchvision.models.feature_extraction",
        "kornia", "kornia.feature", "kornia.contrib", "kornia.geometry", "kornia.geometry.camera",
        "kornia.geometry.camera.pinhole", "kornia.geometry.linalg", "kornia.utils", "kornia.utils.draw",
        "omegaconf", "stego", "stego.backbones", "stego.backbones.backbone", "stego.stego", "stego.data",
        "pytictac", "pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.loggers",
        "pytorch_lightning.loggers.neptune", "liegroups", "liegroups.torch", "cv2", "seaborn", "skimage",
        "skimage.segmentation", "neptune", "neptune.new", "wandb", "pynvml", "fast_slic", "torch_geometric",
        "torch_geometric.data", "simple_parsing", "dataclasses_json", "yaml_include", "prettytable", "termcolor",
        "pytorch_pwc", "pytorch_pwc.network", "optuna", "torchmetrics", "h5py", "rospkg", "imageio", "pytransform3d",
        "pytransform3d.rotations", "open3d", "rospy", "tf", "tf2_ros", "cv_bridge",
    ]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = mock.MagicMock(name=name)
    sys.modules["pytictac"].accumulate_time = _passthrough_decorator
    sys.modules["pytorch_lightning.utilities"].rank_zero_only = _passthrough_decorator
    sys.modules["pytorch_lightning"].utilities = sys.modules["pytorch_lightning.utilities"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import wild_visual_navigation  # noqa: F401

    return wild_visual_navigation


def blob_segmentation(H, W, S, seed):
    """Voronoi-style integer map with S segments (stand-in for SLIC / k-means cluster maps)."""
    g = torch.Generator().manual_seed(seed)
    cy = torch.rand(S, generator=g) * H
    cx = torch.rand(S, generator=g) * W
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    d = (ys[..., None] - cy) ** 2 + (xs[..., None] - cx) ** 2
    return d.argmin(-1).to(torch.int64)


def pin_segments():
    from wild_visual_navigation.feature_extractor.segment_extractor import SegmentExtractor
    from wild_visual_navigation.feature_extractor.feature_extractor import FeatureExtractor
    from oracle import segments as O

    out = {}
    se = SegmentExtractor()
    cases = {
        "blobs64": blob_segmentation(64, 64, 12, 0),
        "blobs96x64": blob_segmentation(96, 64, 23, 1),
        "grid64": O.segment_grid(64, 64, 16)[0, 0],
        "gap": torch.where(blob_segmentation(48, 48, 9, 2) == 4, torch.tensor(8), blob_segmentation(48, 48, 9, 2)),
    }
    for name, seg in cases.items():
        seg4 = seg[None, None]
        adj_ref = se.adjacency_list(seg4)
        cen_ref = se.centers(seg4)
        adj = O.adjacency_list(seg4)
        cen = O.centers(seg4)
        assert torch.equal(adj, adj_ref), name
        assert torch.allclose(cen, cen_ref, atol=1e-4, equal_nan=True), name
        H, W = seg.shape
        D = 24
        dense = torch.randn(1, D, H, W, generator=torch.Generator().manual_seed(7))
        fe = FeatureExtractor.__new__(FeatureExtractor)
        fe._feature_type, fe._segmentation_type = "dino", "grid"
        sp_ref = FeatureExtractor.sparsify_features(fe, dense, seg)
        sp = O.sparsify_features(dense, seg)
        assert torch.allclose(sp, sp_ref, atol=1e-5, equal_nan=True), name
        out[name] = {"seg": seg.to(torch.int32), "adjacency": adj_ref, "centers": cen_ref, "dense_seed": 7,
                     "dense_D": D, "sparsified": sp_ref}
    return out


def pin_label_pool():
    from oracle import segments as O

    try:
        from wild_visual_navigation.traversability_estimator.nodes import MissionNode
    except Exception as e:  # pragma: no cover
        print("MissionNode import failed:", e)
        return None
    out = {}
    for name, (H, W, S, seed) in {"a": (48, 48, 9, 3), "b": (64, 96, 30, 4)}.items():
        seg = blob_segmentation(H, W, S, seed)
        g = torch.Generator().manual_seed(seed)
        mask = torch.rand(3, H, W, generator=g)
        mask[torch.rand(3, H, W, generator=g) < 0.6] = float("nan")
        mask[:, : H // 3] = float("nan")  # some segments end up with no label at all
        node = MissionNode.__new__(MissionNode)
        node._supervision_mask = mask
        node._features = torch.zeros(S, 4)
        node._feature_segments = seg
        node.update_supervision_signal()
        sig, valid = O.update_supervision_signal(mask, seg)
        assert torch.allclose(sig, node._supervision_signal, atol=1e-6), name
        assert torch.equal(valid, node._supervision_signal_valid), name
        out[name] = {"seg": seg.to(torch.int32), "mask": mask, "signal": node._supervision_signal,
                     "valid": node._supervision_signal_valid}
    return out


def pin_mlp():
    from wild_visual_navigation.model.simple_mlp import SimpleMLP
    from wild_visual_navigation.utils.loss import TraversabilityLoss
    from wild_visual_navigation.utils.data import Data, Batch
    from oracle import mlp as O

    out = {}
    # (a) the reference's own fixture: assets/graph/graph.pt  (PyG Data: x[100,90], y, y_valid)
    cases = {}
    try:
        import pickle

        class _U(pickle.Unpickler):
            def find_class(self, module, name):
                if module.startswith("torch_geometric"):
                    return type(name, (dict,), {"__setstate__": lambda s, st: s.update(st if isinstance(st, dict) else {})})
                return super().find_class(module, name)

        pk = types.ModuleType("pk")
        pk.Unpickler = _U
        pk.load = lambda f, **kw: _U(f).load()
        pk.__name__ = "pickle"
        obj = torch.load(os.path.join(REF, "assets/graph/graph.pt"), map_location="cpu", pickle_module=pk,
                         weights_only=False)
        store = obj.get("_store", obj)
        store = store if "x" in store else store.get("_mapping", store)
        cases["graph_pt_D90"] = (store["x"].float(), store["y"].float(), store["y_valid"].bool())
        print("loaded assets/graph/graph.pt:", {k: tuple(v.shape) for k, v in store.items() if hasattr(v, "shape")})
    except Exception as e:  # pragma: no cover
        print("could not unpickle assets/graph/graph.pt (", type(e).__name__, e, ") -> synthetic D=90 case instead")
        gg = torch.Generator().manual_seed(5)
        yv = torch.rand(100, generator=gg) < 0.16
        cases["synthetic_D90"] = (torch.randn(100, 90, generator=gg) * 1.3, yv.float() * (0.5 + 0.5 * torch.rand(100, generator=gg)), yv)
    gg = torch.Generator().manual_seed(6)
    yv = torch.rand(160, generator=gg) < 0.2
    cases["synthetic_D384"] = (torch.randn(160, 384, generator=gg), yv.float() * (0.5 + 0.5 * torch.rand(160, generator=gg)), yv)

    for name, (x, y, yv) in cases.items():
        D = x.shape[1]
        torch.manual_seed(42)  # seed_everything(42), traversability_estimator.py:78
        model = SimpleMLP(input_size=D, hidden_sizes=[256, 32, 1], reconstruction=True)
        model.train()
        loss_fn = TraversabilityLoss(w_trav=0.03, w_reco=0.5, w_temp=0.0, anomaly_balanced=True, model=model,
                                     method="latest_measurement", confidence_std_factor=0.5, log_enabled=False,
                                     log_folder="/tmp")
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        st = O.TrainState(sd0)
        # split into two "nodes" and re-batch through the reference's Batch to pin a8 as well
        h = x.shape[0] // 2
        batch = Batch.from_data_list([Data(x=x[:h], y=y[:h], y_valid=yv[:h]), Data(x=x[h:], y=y[h:], y_valid=yv[h:])])
        assert torch.equal(batch.x, x) and torch.equal(batch.y, y) and torch.equal(batch.y_valid, yv)
        traj = []
        first = None
        for step in range(10):
            res = model(batch)
            loss, aux, _ = loss_fn(batch, res, step=step, log_step=False)
            if first is None:
                first = {"res": res.detach().clone(), "confidence": aux["confidence"].clone(),
                         "loss_trav_confidence": aux["loss_trav_confidence"].detach().clone()}
            opt.zero_grad()
            loss.backward()
            opt.step()
            cg = loss_fn._confidence_generator
            ref_row = [loss.item(), aux["loss_trav"].item(), aux["loss_reco"].item(), cg.mean.item(), cg.std.item()]
            o = O.train_step(st, x, y, yv)
            got = [o["loss_total"], o["loss_trav"], o["loss_reco"], o["mean"], o["std"]]
            assert np.allclose(ref_row, got, rtol=2e-4, atol=1e-6), (name, step, ref_row, got)
            traj.append(ref_row)
        sd1 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k in sd1:
            assert torch.allclose(sd1[k], st.sd[k], atol=2e-5), (name, k, (sd1[k] - st.sd[k]).abs().max())
        o0 = O.mlp_forward(sd0, x)
        assert torch.allclose(o0, first["res"], atol=1e-5)
        out[name] = {"x": x, "y": y, "y_valid": yv, "sd0": sd0, "sd10": sd1, "traj": torch.tensor(traj),
                     "res0": first["res"], "confidence0": first["confidence"]}
        print(f"  {name}: 10 Adam steps pinned, loss {traj[0][0]:.5f} -> {traj[-1][0]:.5f}")
    return out


def pin_confidence():
    from wild_visual_navigation.utils.confidence_generator import ConfidenceGenerator
    from oracle import mlp as O

    out = {}
    g = torch.Generator().manual_seed(11)
    for sf in (0.5, 1.0):
        cg = ConfidenceGenerator(std_factor=sf, method="latest_measurement")
        x = torch.rand(300, generator=g) * 3
        pos = x[:40]
        ref = cg.update(x, pos, step=0)
        mine = O.confidence_from_stats(x, float(pos.mean()), float(pos.std()), sf)
        assert torch.allclose(ref, mine, atol=1e-6)
        out[f"sf{sf}"] = {"x": x, "n_pos": 40, "confidence": ref, "mean": cg.mean.clone(), "std": cg.std.clone()}
    return out


def _kornia_transform_points(trans_01, points_1):
    """Restatement of kornia.geometry.linalg.transform_points (kornia is absent; the reference's meshes.py calls it)."""
    ones = torch.ones_like(points_1[..., :1])
    ph = torch.cat([points_1, ones], dim=-1)
    out = torch.matmul(ph, trans_01.transpose(-1, -2))
    z = out[..., -1:]
    scale = torch.where(z.abs() > 1e-8, 1.0 / (z + 1e-8), torch.ones_like(z))
    return scale * out[..., :-1]


def _pose(x, y, yaw, z=0.0):
    import math

    T = torch.eye(4)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
    T[0, 3], T[1, 3], T[2, 3] = x, y, z
    return T


def pin_geometry():
    """First-party geometry / bookkeeping of the supervision path, executed from the reference:
    meshes.make_plane / make_polygon_from_points / make_dense_plane, SupervisionNode.make_footprint_with_node,
    ImageProjector's scaled camera matrix, BaseGraph / DistanceWindowGraph / MaxElementsGraph eviction and radius queries
    (real networkx).  Third-party pieces they touch are stubbed by restatements: kornia transform_points (above) and the
    liegroups SE(3) distance (translation norm; the pin cases use planar poses whose relative rotation is about z, for which
    |log(T)[:3]| is computed by the product's own se3 helper and cross-checked against the closed form below)."""
    import wild_visual_navigation.utils.meshes as RM

    RM.transform_points = _kornia_transform_points
    from wild_visual_navigation.traversability_estimator import nodes as RN, graphs as RG

    RN.make_plane, RN.make_polygon_from_points, RN.make_dense_plane = RM.make_plane, RM.make_polygon_from_points, RM.make_dense_plane
    out = {"planes": [], "footprints": [], "graphs": {}, "projector": []}
    for kw in (dict(x=0.0, y=0.6, grid_size=2), dict(x=1.0, y=0.6, grid_size=25), dict(y=0.4, z=0.3, grid_size=3)):
        pose = _pose(1.0, -2.0, 0.7, 0.2)
        full = {k: v for k, v in kw.items()}
        out["planes"].append({"kw": full, "pose": pose, "points": RM.make_plane(pose=pose, **full)})
    sq = torch.tensor([[0.0, 0, 0], [1, 0, 0], [1, 2, 0.5], [0, 2, 0.5]])
    out["polygon"] = {"points": sq, "grid_size": 10, "out": RM.make_polygon_from_points(sq, grid_size=10)}
    out["dense"] = {"pose": _pose(0.3, 0.1, -0.4), "out": RM.make_dense_plane(y=0.4, z=0.3, pose=_pose(0.3, 0.1, -0.4), grid_size=5)}

    def sup(t, x, y, yaw):
        return RN.SupervisionNode(timestamp=t, pose_base_in_world=_pose(x, y, yaw), pose_footprint_in_base=_pose(0, 0, 0, -0.3),
                                  width=0.6, length=1.0, height=0.4, supervision=torch.ones(1),
                                  traversability=torch.tensor([0.7]), traversability_var=torch.tensor([0.1]))

    a, b = sup(1.0, 0.0, 0.0, 0.0), sup(2.0, 0.9, 0.2, 0.3)
    out["footprints"].append({"prev": (1.0, 0.0, 0.0, 0.0), "cur": (2.0, 0.9, 0.2, 0.3), "side_prev": a.get_side_points(),
                              "side_cur": b.get_side_points(), "footprint": b.make_footprint_with_node(a)})

    # graphs: distance = translation norm (valid for the planar, small-rotation pin poses to 1e-3; stated in the fixture)
    def dist(self, other):
        return torch.linalg.norm(self.pose_base_in_world[:3, 3] - other.pose_base_in_world[:3, 3])

    RN.BaseNode.distance_to = dist
    xs = [0.0, 0.05, 0.5, 1.1, 1.15, 2.0, 3.5, 3.6, 5.2, 6.0]
    def run(graph):
        acc = []
        for i, x in enumerate(xs):
            acc.append(bool(graph.add_node(RN.BaseNode(float(i), _pose(x, 0.0, 0.0)))))
        return acc, [n.timestamp for n in graph.get_nodes()]
    g = RG.BaseGraph(edge_distance=0.2)
    acc, kept = run(g)
    q = [n.timestamp for n in g.get_nodes_within_radius_range(g.get_last_node(), 0, 2.6)]
    out["graphs"]["base"] = {"xs": xs, "edge_distance": 0.2, "accepted": acc, "kept": kept, "radius": 2.6, "within": q}
    g = RG.DistanceWindowGraph(edge_distance=0.2, max_distance=2.0)
    acc, kept = run(g)
    out["graphs"]["distance_window"] = {"xs": xs, "edge_distance": 0.2, "max_distance": 2.0, "accepted": acc, "kept": kept}
    g = RG.MaxElementsGraph(edge_distance=0.2, max_elements=4)
    acc, kept = run(g)
    out["graphs"]["max_elements"] = {"xs": xs, "edge_distance": 0.2, "max_elements": 4, "accepted": acc, "kept": kept}

    # ImageProjector: scaled camera matrix (kornia.PinholeCamera is a mock here: read its call arguments)
    import wild_visual_navigation.image_projector.image_projector as RI

    for (h, w, nh, nw) in ((1080, 1440, 448, None), (540, 720, 224, 224), (1080, 1440, 448, 600)):
        K = torch.eye(4)[None].clone()
        K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2] = 1000.0, 1100.0, 700.0, 500.0
        RI.PinholeCamera.reset_mock()
        RI.ImageProjector(K, torch.tensor(h), torch.tensor(w), new_h=nh, new_w=nw)
        sK = RI.PinholeCamera.call_args[0][0]
        out["projector"].append({"h": h, "w": w, "new_h": nh, "new_w": nw, "K": K, "sK": sK.clone(),
                                 "sh": int(RI.PinholeCamera.call_args[0][2]), "sw": int(RI.PinholeCamera.call_args[0][3])})
    return out


def pin_checkpoint():
    """Key layout of the reference's checkpoint pieces (traversability_estimator.py:377-429): SimpleMLP.state_dict,
    TraversabilityLoss.state_dict (which carries ``_model.*`` because the loss registers the model as a sub-module) and
    torch.optim.Adam.state_dict after three steps -- a small D = 16 instance."""
    from wild_visual_navigation.model.simple_mlp import SimpleMLP
    from wild_visual_navigation.utils.loss import TraversabilityLoss
    from wild_visual_navigation.utils.data import Data

    torch.manual_seed(42)
    D = 16
    model = SimpleMLP(input_size=D, hidden_sizes=[256, 32, 1], reconstruction=True)
    loss_fn = TraversabilityLoss(w_trav=0.03, w_reco=0.5, w_temp=0.0, anomaly_balanced=True, model=model,
                                 method="latest_measurement", confidence_std_factor=0.5, log_enabled=False, log_folder="/tmp")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    empty_opt = opt.state_dict()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(40, D, generator=g)
    yv = torch.rand(40, generator=g) < 0.3
    y = yv.float() * 0.8
    for step in range(3):
        b = Data(x=x, y=y, y_valid=yv)
        loss, _, _ = loss_fn(b, model(b), step=step, log_step=False)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return {"step": 3, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
            "traversability_loss_state_dict": loss_fn.state_dict(), "loss": float(loss.item()),
            "_empty_optimizer_state_dict": empty_opt, "_D": D}


def main():
    import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    print("pinning segments / pooling ...")
    torch.save(pin_segments(), os.path.join(GOLDEN, "segments.pt"))
    lp = pin_label_pool()
    if lp is not None:
        torch.save(lp, os.path.join(GOLDEN, "label_pool.pt"))
    print("pinning confidence ...")
    torch.save(pin_confidence(), os.path.join(GOLDEN, "confidence.pt"))
    print("pinning MLP / loss / Adam ...")
    torch.save(pin_mlp(), os.path.join(GOLDEN, "mlp_train.pt"))
    print("pinning supervision geometry / graphs / checkpoint layout ...")
    torch.save(pin_checkpoint(), os.path.join(GOLDEN, "ref_checkpoint.pt"))
    torch.save(pin_geometry(), os.path.join(GOLDEN, "geometry.pt"))
    print("oracle pinned; fixtures in", GOLDEN)


if __name__ == "__main__":
    main()
