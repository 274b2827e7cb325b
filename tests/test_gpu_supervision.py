"""Supervision-mask path on the GPU: projection + convex-polygon fill + fmin merge for all mission nodes in one
launch (csrc/supervision.hip), batched label pooling, and TraversabilityEstimator.add_supervision_node end to end -- each against
the CPU oracle (oracle/supervision.py, oracle/segments.py), bit for bit for the masks.  Also: the general pooling kernels are
deterministic (64-bit fixed-point / fixed-order sums)."""
import math

import numpy as np
import pytest
import torch

from oracle import segments as OS, supervision as OSV
from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd.cfg import ExperimentParams
from wild_visual_navigation_amd.image_projector import ImageProjector
from wild_visual_navigation_amd.traversability_estimator import MissionNode, SupervisionNode, TraversabilityEstimator

pytestmark = pytest.mark.gpu


def _pose(x, y, yaw, z=0.0):
    T = torch.eye(4)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
    T[0, 3], T[1, 3], T[2, 3] = x, y, z
    return T


def _cam_looking_forward_down(x, y, yaw, pitch=0.5, h=0.8):
    """camera frame: z forward, x right, y down; mounted at height h, pitched down by `pitch`."""
    base = _pose(x, y, yaw, h)
    R = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])         # camera axes in the base frame
    cp, sp = math.cos(pitch), math.sin(pitch)
    Rp = torch.tensor([[1.0, 0, 0], [0, cp, sp], [0, -sp, cp]])                       # tilt the optical axis DOWN (camera y points down)
    T = torch.eye(4)
    T[:3, :3] = R @ Rp
    return base @ T


def _K(f, cx, cy):
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = f
    K[0, 2], K[1, 2] = cx, cy
    return K


def test_project_render_fmin_matches_oracle_bit_for_bit(dev):
    g = torch.Generator().manual_seed(0)
    H, W, n = 120, 160, 5
    Ks = [_K(110.0 + 7 * i, 80.0 + i, 60.0 - i) for i in range(n)]
    poses = [_cam_looking_forward_down(0.2 * i, -0.1 * i, 0.1 * i - 0.2, 0.45 + 0.03 * i) for i in range(n)]
    poses[4] = _cam_looking_forward_down(5.0, 0.0, 0.0, 0.5)            # beyond the footprint, looking away: every point behind the camera -> untouched
    quad = torch.tensor([[1.2, 0.35, 0.0], [1.2, -0.35, 0.0], [2.4, -0.45, 0.0], [2.4, 0.45, 0.0]])
    from wild_visual_navigation_amd.utils import make_polygon_from_points

    pts = make_polygon_from_points(quad, grid_size=10)                 # 40 points, like make_footprint_with_node
    masks0 = []
    for i in range(n):
        m = torch.full((3, H, W), float("nan"))
        if i % 2 == 0:
            m[:, :, : W // 2] = torch.rand(3, H, W // 2, generator=g)
        masks0.append(m)
    masks = [m.clone().to(dev) for m in masks0]
    proj, depth = ops.project_render_fmin([k.to(dev) for k in Ks], [p.to(dev) for p in poses], masks, pts.to(dev), 0.65,
                                          want_projected=True)
    proj, depth = proj.cpu().numpy(), depth.cpu().numpy()
    changed = 0
    for i in range(n):
        want_p, want_z = OSV.project_points_raw(Ks[i].numpy(), poses[i].numpy(), pts.numpy())   # raw: finite behind the camera too
        assert np.array_equal(proj[i], want_p) and np.array_equal(depth[i], want_z), f"projected points of node {i}"
        want = OSV.render_fmin(masks0[i].numpy(), Ks[i].numpy(), poses[i].numpy(), pts.numpy(), 0.65)
        got = masks[i].cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), f"mask of node {i}"
        changed += int((~np.isnan(got[0]) & np.isnan(masks0[i].numpy()[0])).sum())
    assert changed > 2000                                               # the polygon is visible in the first four cameras
    assert torch.equal(torch.isnan(masks[4].cpu()), torch.isnan(masks0[4]))
    # value from device memory (traversability tensors live on the GPU in the learning node): same result, no host sync
    masks2 = [m.clone().to(dev) for m in masks0]
    ops.project_render_fmin([k.to(dev) for k in Ks], [p.to(dev) for p in poses], masks2, pts.to(dev), torch.tensor([0.65], device=dev))
    for a, b in zip(masks, masks2):
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))


def test_image_projector_project_and_render(dev):
    """The class-level call of image_projector.py:152-197 (tests/test_image_projector.py in the reference only plots)."""
    K = _K(150.0, 112.0, 112.0)[None].to(dev)
    ip = ImageProjector(K, torch.tensor(224), torch.tensor(224))
    pose = _cam_looking_forward_down(0, 0, 0)[None].to(dev)
    quad = torch.tensor([[1.0, 0.3, 0.0], [1.0, -0.3, 0.0], [2.0, -0.3, 0.0], [2.0, 0.3, 0.0]], device=dev)[None]
    masks, overlay, proj, valid = ip.project_and_render(pose, quad, torch.tensor([1.0, 0.5, 0.25]), image=torch.zeros(1, 3, 224, 224, device=dev))
    inside = OSV.fill_mask(OSV.project_points(ip.camera.intrinsics[0].cpu().numpy(), pose[0].cpu().numpy(), quad[0].cpu().numpy()), 224, 224)
    assert inside.sum() > 500
    assert torch.equal(~torch.isnan(masks[0, 0]).cpu(), torch.from_numpy(inside))
    assert float(masks[0, 1][torch.from_numpy(inside).to(dev)].unique()) == 0.5 and torch.isnan(masks[0, 2][~torch.from_numpy(inside).to(dev)]).all()
    assert valid.all() and proj.shape == (1, 4, 2)
    assert float(overlay[0, 0][torch.from_numpy(inside).to(dev)].min()) == 1.0 and float(overlay[0, 0][~torch.from_numpy(inside).to(dev)].max()) == 0.0
    p2, v2, vz = ip.project(pose, quad)
    assert torch.equal(torch.nan_to_num(p2), torch.nan_to_num(proj)) and vz.all()
    # a point behind the camera: project() returns its (finite) pinhole coordinates with valid_z False, as the reference's
    # ImageProjector.project does (image_projector.py:126-150); project_and_render NaNs it (:180)
    behind = torch.tensor([[[1.0, 0.3, 0.0], [-3.0, 0.2, 0.0], [2.0, -0.3, 0.0], [2.0, 0.3, 0.0]]], device=dev)
    p3, v3, vz3 = ip.project(pose, behind)
    raw, z = OSV.project_points_raw(ip.camera.intrinsics[0].cpu().numpy(), pose[0].cpu().numpy(), behind[0].cpu().numpy())
    assert np.array_equal(p3[0].cpu().numpy(), raw) and torch.isfinite(p3).all()
    assert vz3[0].tolist() == [True, False, True, True] and not bool(v3[0, 1])
    _, _, p4, _ = ip.project_and_render(pose, behind, torch.tensor([1.0, 0.5, 0.25]))
    assert torch.isnan(p4[0, 1]).all() and torch.equal(p4[0, [0, 2, 3]], p3[0, [0, 2, 3]])


def test_untraversable_plane_and_infinite_vertices(dev):
    """a supervision node with is_untraversable=True hands add_supervision_node the 10 x 10 x 10 dense plane of
    make_dense_plane (1000 points; the kernel's vertex arrays are dynamic LDS now), and vertices at infinity (z' ~ 0) must
    poison their scan lines the way torch's min / max / clamp propagate NaN -- both bit for bit against the oracle."""
    from wild_visual_navigation_amd.traversability_estimator import SupervisionNode

    H, W = 120, 160

    def sup(t, x, untrav):
        return SupervisionNode(timestamp=t, pose_base_in_world=_pose(x, 0, 0), pose_footprint_in_base=_pose(0, 0, 0, 0.0), width=0.7,
                               length=1.0, height=0.4, supervision=torch.ones(1), traversability=torch.tensor([0.2]),
                               traversability_var=torch.tensor([0.1]), is_untraversable=untrav,
                               twist_in_base=torch.tensor([0.8, 0.1, 0.0]))

    fp = sup(1.0, 2.4, True).make_footprint_with_node(sup(0.0, 1.6, False))
    assert fp.shape == (1000, 3)
    Ks = [_K(115.0, 80.0, 60.0), _K(130.0, 82.0, 58.0)]
    poses = [_cam_looking_forward_down(0.0, 0.0, 0.0), _cam_looking_forward_down(0.4, 0.2, -0.2)]
    masks0 = [torch.full((3, H, W), float("nan")) for _ in range(2)]
    masks = [m.clone().to(dev) for m in masks0]
    ops.project_render_fmin([k.to(dev) for k in Ks], [p.to(dev) for p in poses], masks, fp.to(dev), 0.2)
    touched = 0
    for i in range(2):
        want = OSV.render_fmin(masks0[i].numpy(), Ks[i].numpy(), poses[i].numpy(), fp.numpy(), 0.2)
        assert np.array_equal(masks[i].cpu().numpy(), want, equal_nan=True), i
        touched += int((~np.isnan(want[0])).sum())
    assert touched > 500
    # a vertex that projects to v = +inf (camera frame == world frame; y = 1e38 overflows the projection): the edge that STARTS at
    # it evaluates to (y - inf) * 0 + x0 = NaN on every row it is active on -- torch's min / max over the edges propagate that
    # NaN and those scan lines stay unfilled; fminf / fmaxf would have dropped the NaN and filled them from the other edges
    K1 = _K(100.0, 80.0, 60.0)
    pts = torch.tensor([[-0.3, -0.2, 1.0], [0.3, -0.2, 1.0], [0.1, 1e38, 1.0], [-0.3, 0.25, 1.0]])
    uv = OSV.project_points(K1.numpy(), torch.eye(4).numpy(), pts.numpy())
    assert np.isinf(uv[2, 1]) and np.isfinite(uv[[0, 1, 3]]).all()
    m0 = torch.full((1, H, W), float("nan"))
    m = [m0.clone().to(dev)]
    ops.project_render_fmin([K1.to(dev)], [torch.eye(4).to(dev)], m, pts.to(dev), 0.5)
    want = OSV.render_fmin(m0.numpy(), K1.numpy(), torch.eye(4).numpy(), pts.numpy(), 0.5)
    assert np.array_equal(m[0].cpu().numpy(), want, equal_nan=True)
    assert (~np.isnan(want[0, 40:85])).any() and np.isnan(want[0, 85:]).all()     # rows 40..84 filled, the poisoned rows below are not


def test_label_pool_batched_equals_single_and_golden(dev, golden):
    c = golden("label_pool.pt")["a"]
    masks, segs, ns = [], [], []
    g = torch.Generator().manual_seed(3)
    H, W = c["mask"].shape[1:]
    for i in range(4):
        m = c["mask"].clone()
        if i:
            m[torch.rand(3, H, W, generator=g) < 0.3] = float("nan")
        masks.append(m.to(dev).contiguous())
        segs.append(c["seg"].to(dev).to(torch.int32).contiguous())
        ns.append(int(c["seg"].max()) + 1 + (i % 2))            # ragged segment counts
    out = ops.label_pool_batched(masks, segs, ns)
    for i in range(4):
        sig, val = ops.label_pool(masks[i], segs[i], ns[i])
        assert torch.equal(sig, out[i][0]) and torch.equal(val, out[i][1])
    assert torch.allclose(out[0][0].cpu(), c["signal"], atol=1e-6) and torch.equal(out[0][1].cpu(), c["valid"])


def test_pooling_kernels_are_deterministic(dev):
    """The three general pooling kernels used fp32 atomics in round 1 (run-to-run differences in the last bits); they now sum
    in 64-bit fixed point / fixed order: repeated runs are bit-identical, and still equal the oracle."""
    g = torch.Generator().manual_seed(5)
    H, G, D, S = 224, 28, 384, 37
    seg = torch.randint(0, S, (1, H, H), generator=g)
    seg[0, :40] = -1
    tok = torch.randn(1, G * G, D, generator=g)
    a = ops.segpool_bilinear_mean(seg.to(dev), tok.to(dev), G, S)
    for _ in range(5):
        assert torch.equal(ops.segpool_bilinear_mean(seg.to(dev), tok.to(dev), G, S), a)
    from oracle import interfaces as OI

    dense = OI.upsample_bilinear_ac(tok.reshape(1, G, G, D).permute(0, 3, 1, 2), H)
    want = OS.sparsify_features(dense, seg[0].clamp_min(0) if False else torch.where(seg[0] < 0, torch.tensor(S), seg[0]))[:S]
    assert (a[0].cpu() - want).abs().max().item() < 2e-5
    dense_t = torch.randn(1, 96 * 96, 130, generator=g).to(dev)
    seg2 = torch.randint(0, 50, (1, 96 * 96), generator=g).to(dev)
    b = ops.segmean_tokens(seg2, dense_t, 50)
    for _ in range(5):
        assert torch.equal(ops.segmean_tokens(seg2, dense_t, 50), b)
    ref = torch.stack([dense_t[0][seg2[0] == s].mean(0) for s in range(50)])
    assert (b[0] - ref).abs().max().item() < 1e-5
    mask = torch.rand(3, H, H, generator=g)
    mask[torch.rand(3, H, H, generator=g) < 0.5] = float("nan")
    s0, v0 = ops.label_pool(mask.to(dev), seg[0].clamp_min(0).to(dev), S)
    for _ in range(5):
        s1, v1 = ops.label_pool(mask.to(dev), seg[0].clamp_min(0).to(dev), S)
        assert torch.equal(s0, s1) and torch.equal(v0, v1)


def test_add_supervision_node_end_to_end(dev):
    """traversability_estimator.py:198-300: three mission nodes along a path, two supervision nodes; the footprint between
    the two robot poses must land in every mission node's mask (fmin with what was there), the per-segment labels must be
    the oracle's pooling of the oracle's masks, and a supervision node closer than supervision_distance_thr only updates the
    previous node's traversability."""
    H = W = 128
    p = ExperimentParams()
    p.model.simple_mlp_cfg.input_size = 90
    te = TraversabilityEstimator(p, device=dev, max_distance=3.0, image_distance_thr=0.2, supervision_distance_thr=0.1,
                                 min_samples_for_training=1)
    g = torch.Generator().manual_seed(1)
    S = 16
    seg = (torch.arange(H)[:, None] // 32 * 4 + torch.arange(W)[None] // 32).to(torch.int64)   # 4 x 4 grid of segments
    K = _K(120.0, 64.0, 64.0)[None]
    nodes = []
    for i, x in enumerate((0.0, 0.5, 0.55, 1.0)):       # the third is closer than image_distance_thr to the second: rejected
        n = MissionNode(timestamp=float(i), pose_base_in_world=_pose(x, 0, 0), pose_cam_in_world=_cam_looking_forward_down(x, 0, 0).to(dev),
                        image_projector=ImageProjector(K.to(dev), torch.tensor(H), torch.tensor(W)))
        n.features = torch.randn(S, 90, generator=g).to(dev)
        n.feature_segments = seg.to(dev)
        ok = te.add_mission_node(n)
        assert ok == (i != 2)
        if ok:
            nodes.append(n)
            assert torch.isnan(n.supervision_mask).all() and not n.is_valid()

    def sup(t, x, trav):
        return SupervisionNode(timestamp=t, pose_base_in_world=_pose(x, 0, 0), pose_footprint_in_base=_pose(0, 0, 0, 0.0),
                               width=0.7, length=1.0, height=0.4, supervision=torch.ones(1),
                               traversability=torch.tensor([trav]), traversability_var=torch.tensor([0.1]))

    s0, s1, s1b, s2 = sup(10.0, 2.0, 0.9), sup(11.0, 2.8, 0.6), sup(11.5, 2.85, 0.3), sup(12.0, 3.4, 0.8)
    assert te.add_supervision_node(s0) is False                      # no previous supervision node yet
    assert te.add_supervision_node(s1) is True
    fp1 = s1.make_footprint_with_node(s0)
    want = {}
    # the radius query returns the nodes AROUND the newest mission node, not that node itself (graphs.py:169:
    # `nodes = sorted(list(length)[1:])`): its mask stays untouched, exactly as in the reference
    newest, nodes = nodes[-1], nodes[:-1]
    assert torch.isnan(newest.supervision_mask).all()
    for n in nodes:
        m = np.full((3, H, W), np.nan, dtype=np.float32)
        want[n.timestamp] = OSV.render_fmin(m, K[0].numpy(), n.pose_cam_in_world.cpu().numpy(), fp1.cpu().numpy(), np.float32(0.6))
        assert np.array_equal(n.supervision_mask.cpu().numpy(), want[n.timestamp], equal_nan=True), n.timestamp
        sig, val = OS.update_supervision_signal(torch.from_numpy(want[n.timestamp]), seg)
        assert torch.allclose(n.supervision_signal.cpu(), sig, atol=1e-6) and torch.equal(n.supervision_signal_valid.cpu(), val)
    assert any(n.is_valid() for n in nodes)
    assert te.add_supervision_node(s1b) is False                     # 5 cm from s1 (< supervision_distance_thr): not added ...
    assert float(s1.traversability) == pytest.approx(0.3)            # ... but s1 takes the lower traversability
    assert te.add_supervision_node(s2) is True                       # second footprint, fmin-merged into the first
    fp2 = s2.make_footprint_with_node(s1)
    assert torch.isnan(newest.supervision_mask).all()
    for n in nodes:
        w2 = OSV.render_fmin(want[n.timestamp], K[0].numpy(), n.pose_cam_in_world.cpu().numpy(), fp2.cpu().numpy(), np.float32(0.8))
        assert np.array_equal(n.supervision_mask.cpu().numpy(), w2, equal_nan=True)
    res = te.train()
    assert res["loss_total"] != -1 and te.step == 1
    # a mission node in range that has no features / segments (its mask is merged, nothing is pooled, no exception),
    # and a segment map whose ids run past the feature rows (signal length = seg.max() + 1 as in the reference, nodes.py:413)
    bare = MissionNode(timestamp=20.0, pose_base_in_world=_pose(1.4, 0, 0), pose_cam_in_world=_cam_looking_forward_down(1.4, 0, 0).to(dev),
                       image_projector=ImageProjector(K.to(dev), torch.tensor(H), torch.tensor(W)), use_for_training=False)
    assert te.add_mission_node(bare) is False and bare.supervision_mask is None     # in the graph, but not prepared for training (:182-196)
    wide = MissionNode(timestamp=21.0, pose_base_in_world=_pose(1.8, 0, 0), pose_cam_in_world=_cam_looking_forward_down(1.8, 0, 0).to(dev),
                       image_projector=ImageProjector(K.to(dev), torch.tensor(H), torch.tensor(W)))
    wide.features = torch.randn(S, 90, generator=g).to(dev)
    wide.feature_segments = (seg + 3).to(dev)                       # ids 3 .. 18 against 16 feature rows
    assert te.add_mission_node(wide) is True and wide.num_segments() == S + 3
    newest2 = MissionNode(timestamp=22.0, pose_base_in_world=_pose(2.2, 0, 0), pose_cam_in_world=_cam_looking_forward_down(2.2, 0, 0).to(dev),
                          image_projector=ImageProjector(K.to(dev), torch.tensor(H), torch.tensor(W)))
    newest2.features = torch.randn(S, 90, generator=g).to(dev)
    newest2.feature_segments = seg.to(dev)
    assert te.add_mission_node(newest2) is True
    assert te.add_supervision_node(sup(13.0, 4.2, 0.7)) is True
    assert bare.supervision_signal is None and bare.supervision_mask is not None     # zeros substituted (:262-264), merged, nothing pooled
    assert wide.supervision_signal.shape == (S + 3,)
    sig, val = OS.update_supervision_signal(wide.supervision_mask.cpu(), seg + 3)
    assert torch.allclose(wide.supervision_signal.cpu(), sig, atol=1e-6) and torch.equal(wide.supervision_signal_valid.cpu(), val)
