"""Traversability MLP on the GPU vs (a) the golden vectors produced by the REFERENCE's own
SimpleMLP + TraversabilityLoss + torch.optim.Adam (tests/golden/mlp_train.pt, incl. the reference's
fixture assets/graph/graph.pt) and (b) the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import mlp as OM
from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd.cfg import ExperimentParams
from wild_visual_navigation_amd.model import SimpleMLP
from wild_visual_navigation_amd.traversability_estimator import MissionNode, MlpTrainer, TraversabilityEstimator
from wild_visual_navigation_amd.utils import ConfidenceGenerator, Data

pytestmark = pytest.mark.gpu


def _model(sd0, D, dev):
    m = SimpleMLP(D, [256, 32, 1], True)
    m.load_state_dict(sd0)
    return m.to(dev)


@pytest.mark.parametrize("case", ["graph_pt_D90", "synthetic_D384"])
def test_forward_matches_reference(dev, golden, case):
    c = golden("mlp_train.pt")[case]
    m = _model(c["sd0"], c["x"].shape[1], dev)
    out = m.forward(Data(x=c["x"].to(dev))).cpu()
    assert out.shape == c["res0"].shape and (out - c["res0"]).abs().max().item() < 1e-5
    assert list(m.state_dict()) == list(c["sd0"])


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", ["graph_pt_D90", "synthetic_D384"])
def test_ten_adam_steps_match_reference_trajectory(dev, golden, case, fused):
    """fused: the four-launch step of csrc/mlp_train.hip (what the trainer runs at these sizes); not fused: the general path."""
    c = golden("mlp_train.pt")[case]
    x, y, yv = c["x"].to(dev), c["y"].to(dev), c["y_valid"].to(dev)
    m = _model(c["sd0"], x.shape[1], dev)
    tr = MlpTrainer(m, lr=1e-3, std_factor=0.5, w_trav=0.03, w_reco=0.5, fused=fused)
    traj = []
    for step in range(10):
        losses = tr.train_step(x, y, yv, want_confidence=(step == 0))
        if step == 0:
            assert torch.allclose(tr.last_confidence.cpu(), c["confidence0"], atol=1e-5)
        traj.append(losses.cpu().tolist())
    traj = np.array(traj)
    assert np.allclose(traj, c["traj"].numpy(), rtol=3e-4, atol=2e-6), np.abs(traj - c["traj"].numpy()).max()
    for k, v in m.state_dict().items():
        assert torch.allclose(v.cpu(), c["sd10"][k], atol=3e-5), (k, (v.cpu() - c["sd10"][k]).abs().max())


def test_large_batch_step_matches_oracle(dev):
    """R = 64 frames x 196 grid cells (BASELINE config 3 row count), D = 384: exercises the split-K
    weight-gradient path and the big-R reductions."""
    R, D = 64 * 196, 384
    g = torch.Generator().manual_seed(3)
    x = torch.randn(R, D, generator=g)
    yv = torch.rand(R, generator=g) < 0.16
    y = yv.float() * (0.5 + 0.5 * torch.rand(R, generator=g))
    sd0 = OM.make_mlp_state_dict(D, seed=42)
    m = _model(sd0, D, dev)
    tr = MlpTrainer(m)
    st = OM.TrainState(sd0)
    for _ in range(2):
        got = tr.train_step(x.to(dev), y.to(dev), yv.to(dev)).cpu()
        want = OM.train_step(st, x, y, yv)
    assert abs(got[0].item() - want["loss_total"]) < 2e-6 and abs(got[1].item() - want["loss_trav"]) < 2e-6
    assert abs(got[3].item() - want["mean"]) < 2e-6 and abs(got[4].item() - want["std"]) < 2e-6
    for k in st.sd:
        assert torch.allclose(m.state_dict()[k].cpu(), st.sd[k], atol=2e-5), k


@pytest.mark.parametrize("R,D", [(1280, 90), (2048, 384), (77, 384), (800, 91), (33, 90)])
def test_four_launch_step_equals_general_path_and_oracle(dev, R, D):
    """The bench / live-node sizes through both paths and the CPU oracle (three steps), ragged last row tile, odd D; then a
    compacted batch whose row count lives on the device (rows_dev): both paths ignore the rows behind it; and bit-reproducible."""
    g = torch.Generator().manual_seed(R + D)
    x = torch.randn(R, D, generator=g)
    yv = torch.rand(R, generator=g) < 0.16
    yv[:2] = True
    y = yv.float() * (0.5 + 0.5 * torch.rand(R, generator=g))
    sd0 = OM.make_mlp_state_dict(D, seed=42)
    ma, mb = _model(sd0, D, dev), _model(sd0, D, dev)
    ta, tb = MlpTrainer(ma, fused=True), MlpTrainer(mb, fused=False)
    st = OM.TrainState(sd0)
    for _ in range(3):
        la = ta.train_step(x.to(dev), y.to(dev), yv.to(dev), want_confidence=True).cpu()
        lb = tb.train_step(x.to(dev), y.to(dev), yv.to(dev), want_confidence=True).cpu()
        want = OM.train_step(st, x, y, yv)
        assert torch.allclose(ta.last_confidence, tb.last_confidence, atol=1e-5)
    assert torch.allclose(la, lb, rtol=1e-5, atol=2e-6), (la, lb)
    assert abs(la[0].item() - want["loss_total"]) < 2e-6 and abs(la[2].item() - want["loss_reco"]) < 2e-6
    for k in st.sd:
        assert torch.allclose(ma.state_dict()[k].cpu(), st.sd[k], atol=2e-5), k
        assert torch.allclose(ma.state_dict()[k], mb.state_dict()[k], atol=1e-5), k
    assert int(ta.sync_word.item()) == 0                                      # the arrival counter is back at zero
    # rows_dev: the first n rows are real, garbage behind them
    n = R - R // 3
    rows_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    xg = x.clone()
    xg[n:] = 1e6
    mc, md, me = _model(sd0, D, dev), _model(sd0, D, dev), _model(sd0, D, dev)
    lc = MlpTrainer(mc, fused=True).train_step(xg.to(dev), y.to(dev), yv.to(dev), rows_dev=rows_dev).cpu()
    ld = MlpTrainer(md, fused=True).train_step(x[:n].to(dev), y[:n].to(dev), yv[:n].to(dev)).cpu()
    le = MlpTrainer(me, fused=False).train_step(xg.to(dev), y.to(dev), yv.to(dev), rows_dev=rows_dev).cpu()
    assert torch.equal(lc, ld)                                                # the fused path walks the same tiles: identical bits
    assert torch.allclose(lc, le, rtol=1e-5, atol=2e-6)
    for k in st.sd:
        assert torch.equal(mc.state_dict()[k], md.state_dict()[k]), k
        assert torch.allclose(mc.state_dict()[k], me.state_dict()[k], atol=1e-5), k
    mf = _model(sd0, D, dev)
    lf = MlpTrainer(mf, fused=True).train_step(xg.to(dev), y.to(dev), yv.to(dev), rows_dev=rows_dev).cpu()
    assert torch.equal(lc, lf) and all(torch.equal(mc.state_dict()[k], mf.state_dict()[k]) for k in st.sd)


def test_compact_segment_rows(dev):
    """ops.compact_segment_rows: rows (b, s < nseg[b]) front to back, zeros behind, count on the device."""
    B, S, D = 5, 7, 12
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(B, S, D, generator=g)
    side = torch.rand(B, S, 2, generator=g)
    nseg = torch.tensor([7, 0, 3, 5, 1], dtype=torch.int32)
    feat[1] = float("nan")
    x, so, cnt = ops.compact_segment_rows(feat.to(dev), nseg.to(dev), side.to(dev))
    keep = (torch.arange(S)[None] < nseg[:, None]).reshape(-1)
    n = int(keep.sum())
    assert int(cnt.item()) == n == 16
    assert torch.equal(x[:n].cpu(), feat.reshape(B * S, D)[keep]) and torch.equal(so[:n].cpu(), side.reshape(B * S, 2)[keep])
    assert (x[n:] == 0).all() and (so[n:] == 0).all()


def test_single_labelled_row_gives_nan_std_like_reference(dev):
    D = 90
    x = torch.randn(8, D, generator=torch.Generator().manual_seed(0))
    yv = torch.zeros(8, dtype=torch.bool)
    yv[2] = True
    m = _model(OM.make_mlp_state_dict(D), D, dev)
    losses = MlpTrainer(m).train_step(x.to(dev), yv.float().to(dev), yv.to(dev)).cpu()
    assert torch.isnan(losses[4]) and torch.isnan(losses[0])  # torch.std of one sample is NaN (quirk kept)


def test_confidence_kernel_and_per_pixel_inference(dev, golden):
    """quick_start.py:183-212 per-pixel branch on a small dense map."""
    from wild_visual_navigation_amd import _lib

    c = golden("mlp_train.pt")["graph_pt_D90"]
    m = _model(c["sd10"], 90, dev).eval()
    H = 24
    dense = torch.randn(1, 90, H, H, generator=torch.Generator().manual_seed(4))
    xx = dense[0].permute(1, 2, 0).reshape(-1, 90).contiguous()
    xd = xx.to(dev)
    pred = m.forward(Data(x=xd))
    want = OM.mlp_forward(c["sd10"], xx)
    assert (pred.cpu() - want).abs().max().item() < 1e-5
    cg = ConfidenceGenerator(0.5)
    cg.mean[0], cg.std[0] = 1.1, 0.3
    lr = ((want[:, 1:] - xx) ** 2).mean(1)
    want_conf = cg.inference_without_update(lr)
    trav = torch.empty(H * H, device=dev)
    conf = torch.empty(H * H, device=dev)
    _lib.check(_lib.lib().wvn_mlp_confidence(pred.data_ptr(), 91, xd.data_ptr(), 90, 1.1, 0.3, 0.5,
                                             trav.data_ptr(), conf.data_ptr(), H * H, 90, _lib.stream()))
    assert torch.allclose(conf.cpu(), want_conf, atol=1e-5) and torch.allclose(trav.cpu(), want[:, 0], atol=1e-5)


def test_traversability_estimator_train_loop_and_checkpoint(dev, tmp_path):
    p = ExperimentParams()
    p.model.simple_mlp_cfg.input_size = 90
    te = TraversabilityEstimator(p, device=dev, min_samples_for_training=2)
    assert te.train() == {"mission_graph_num_valid_node": 0, "loss_total": -1}
    g = torch.Generator().manual_seed(0)
    S, H = 12, 48
    for i in range(6):
        n = MissionNode(timestamp=float(i))
        n.features = torch.randn(S, 90, generator=g).to(dev)
        n.feature_segments = (torch.arange(H * H).reshape(H, H) * S // (H * H)).to(dev)
        mask = torch.full((3, H, H), float("nan"))
        mask[:, : H // 2] = 0.5 + 0.5 * torch.rand(3, H // 2, H, generator=g)
        assert te.add_mission_node(n)                       # fresh all-NaN mask + first pooling (nothing labelled yet)
        assert not n.is_valid() and torch.isnan(n.supervision_mask).all()
        te.update_supervision(n, mask.to(dev))              # fmin merge of a rendered mask + re-pooling
        assert n.supervision_signal.shape == (S,) and n.supervision_signal_valid.any() and n.is_valid()
    first = te.train()
    assert set(first) == {"mission_graph_num_valid_node", "loss_total", "loss_trav", "loss_reco"}
    for _ in range(30):
        last = te.train()
    assert last["loss_total"] < first["loss_total"] and te.step == 31
    cg = te._traversability_loss._confidence_generator
    assert torch.isfinite(cg.mean).all() and torch.isfinite(cg.std).all()
    f = te.save_checkpoint(str(tmp_path))
    ck = torch.load(f, weights_only=False)
    assert set(ck) == {"step", "model_state_dict", "optimizer_state_dict", "traversability_loss_state_dict", "loss"}
    te2 = TraversabilityEstimator(p, device=dev, min_samples_for_training=2)
    te2.load_checkpoint(f)
    for k, v in te._model.state_dict().items():
        assert torch.equal(v, te2._model.state_dict()[k])
    assert te2.step == 31 and te2._optimizer.step == te._optimizer.step
    # the live weights hand-off file of the learning node (wvn_learning_node.py:381-394) loads with strict=False
    sd = te._model.state_dict()
    sd["confidence_generator"] = cg.get_dict()
    m = SimpleMLP(90, [256, 32, 1], True)
    m.load_state_dict(sd, strict=False)
