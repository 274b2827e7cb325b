"""Host-side layout helpers of the round-3 kernels (CPU only): the bit-2 <-> bit-3 index order of the fused weights, the fragment
layout of the LayerNorm hand-over, the gather tables of the fused frame ingest (incl. the mirror pass)."""
import torch

from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd.feature_extractor.transforms import ingest_tables, resize_nearest_center_crop


def test_swap23_order_is_an_involution_inside_groups_of_16():
    for n in (16, 384, 1536):
        p = ops.vt_token_order(n)
        assert torch.equal(p[p], torch.arange(n))                        # applying it twice is the identity
        assert torch.equal(p // 16, torch.arange(n) // 16)               # it never leaves its aligned group of 16
        assert p[:16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]


def test_fragment_layout_round_trip():
    """unpack_row_fragments inverts the layout include/wvn_hip.h documents for wvn_proj_mlp_resident's xn_next: fragment (R, s)
    = 1 KB, lane l = row 32 R + (l & 31), the 8 values at columns 16 s + 4 (l >> 5) + {0..3} and 16 s + 8 + 4 (l >> 5) + {0..3}."""
    M = 70
    rows = torch.arange(M * 384, dtype=torch.float32).reshape(M, 384)
    G = (M + 31) // 32
    frag = torch.zeros(G, 24, 64, 8)
    for R in range(G):
        for s in range(24):
            for lane in range(64):
                r, hi = 32 * R + (lane & 31), lane >> 5
                if r < M:
                    frag[R, s, lane, :4] = rows[r, 16 * s + 4 * hi:16 * s + 4 * hi + 4]
                    frag[R, s, lane, 4:] = rows[r, 16 * s + 8 + 4 * hi:16 * s + 8 + 4 * hi + 4]
    assert torch.equal(ops.unpack_row_fragments(frag.reshape(-1), M), rows)


def test_ingest_tables_are_the_image_op_and_its_mirror():
    for (h, w, size) in ((224, 299, 224), (1080, 1440, 448), (64, 64, 64), (300, 200, 128)):
        t, tm = ingest_tables(h, w, size, "cpu"), ingest_tables(h, w, size, "cpu", flip=True)
        img = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(h + w))
        want = resize_nearest_center_crop(img, size)
        got = img[..., t.rows.long(), :][..., t.cols.long()]
        assert torch.equal(got, want)
        assert torch.equal(img[..., tm.rows.long(), :][..., tm.cols.long()], want.flip(-1))
        assert torch.equal(tm.cols, t.cols.flip(0)) and torch.equal(tm.rows, t.rows)
