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
    print("oracle pinned; fixtures in", GOLDEN)


if __name__ == "__main__":
    main()
