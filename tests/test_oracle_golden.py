"""Oracle vs the golden vectors produced from the reference's own code (oracle/pin_reference.py)."""
import numpy as np
import torch

from oracle import mlp as OM, segments as OS


def test_segments_adjacency_centers_sparsify(golden):
    cases = golden("segments.pt")
    assert len(cases) >= 4
    for name, c in cases.items():
        seg = c["seg"].long()
        assert torch.equal(OS.adjacency_list(seg[None, None]), c["adjacency"]), name  # bit-exact integer work
        assert torch.allclose(OS.centers(seg[None, None]), c["centers"], atol=1e-4, equal_nan=True), name
        H, W = seg.shape
        dense = torch.randn(1, c["dense_D"], H, W, generator=torch.Generator().manual_seed(c["dense_seed"]))
        assert torch.allclose(OS.sparsify_features(dense, seg), c["sparsified"], atol=1e-5, equal_nan=True), name


def test_gap_segment_gives_nan_row(golden):
    c = golden("segments.pt")["gap"]
    assert torch.isnan(c["sparsified"][4]).all()  # id 4 was removed from the map: empty mean -> NaN row


def test_label_pool(golden):
    for name, c in golden("label_pool.pt").items():
        sig, valid = OS.update_supervision_signal(c["mask"], c["seg"].long())
        assert torch.allclose(sig, c["signal"], atol=1e-6), name
        assert torch.equal(valid, c["valid"]), name
        assert (~valid).any() and valid.any()


def test_confidence(golden):
    for name, c in golden("confidence.pt").items():
        sf = float(name[2:])
        got = OM.confidence_from_stats(c["x"], float(c["mean"]), float(c["std"]), sf)
        assert torch.allclose(got, c["confidence"], atol=1e-6), name


def test_mlp_forward_and_adam_trajectory(golden):
    cases = golden("mlp_train.pt")
    assert "graph_pt_D90" in cases, "the reference's own fixture assets/graph/graph.pt must be part of the pin"
    for name, c in cases.items():
        out0 = OM.mlp_forward(c["sd0"], c["x"])
        assert torch.allclose(out0, c["res0"], atol=1e-5), name
        st = OM.TrainState(c["sd0"])
        for step in range(10):
            o = OM.train_step(st, c["x"], c["y"], c["y_valid"])
            got = [o["loss_total"], o["loss_trav"], o["loss_reco"], o["mean"], o["std"]]
            assert np.allclose(got, c["traj"][step].numpy(), rtol=2e-4, atol=1e-6), (name, step)
        for k, v in c["sd10"].items():
            assert torch.allclose(st.sd[k], v, atol=2e-5), (name, k)


def test_mlp_data_parallel_decomposition(golden):
    c = golden("mlp_train.pt")["synthetic_D384"]
    x, y, yv = c["x"], c["y"], c["y_valid"]
    st1, st2 = OM.TrainState(c["sd0"]), OM.TrainState(c["sd0"])
    cuts = [0, 37, 90, x.shape[0]]  # ragged shards
    for _ in range(3):
        a = OM.train_step(st1, x, y, yv)
        parts = [OM.phase_a_local(st2.sd, x[i:j], yv[i:j]) for i, j in zip(cuts[:-1], cuts[1:])]
        stats = sum(p[0] for p in parts)
        grads = sum(OM.phase_b_local(st2.sd, x[i:j], y[i:j], yv[i:j], p[1], stats)
                    for (i, j), p in zip(zip(cuts[:-1], cuts[1:]), parts))
        b = OM.phase_c(st2, grads, stats)
        for k in a:
            assert abs(a[k] - b[k]) < 2e-5, (k, a[k], b[k])
    for k in st1.sd:
        assert torch.allclose(st1.sd[k], st2.sd[k], atol=2e-5), (st1.sd[k] - st2.sd[k]).abs().max()  # Adam amplifies fp32 summation-order noise
