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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: the weight images the MX / fp8 A-stationary kernels stream (byte orders as the docstrings of backbone.py state them)
# ---------------------------------------------------------------------------------------------------------------------------
def test_mx_split_is_fp16_plus_scaled_e5m2_residue():
    from wild_visual_navigation_amd.backbone import MX_RES_SCALE, mx_split
    w = torch.randn(8, 64, generator=torch.Generator().manual_seed(0)) * 3
    h, l8, h8 = mx_split(w)
    assert torch.equal(h, w.to(torch.float16))
    res = l8.view(torch.float8_e5m2).float() / MX_RES_SCALE
    # hi + residue carries ~14 bits: the e5m2 residue has 3 significant bits of the fp16 rounding error (|error| <= 2^-11 |w|)
    assert ((h.float() + res - w).abs() <= 2.0 ** -13 * w.abs() + 1e-9).all()
    assert torch.equal(h8.view(torch.float8_e5m2).float(), w.to(torch.float8_e5m2).float())


def test_pack_a384_mx_both_images_byte_for_byte():
    """Image 0: plane 0 = fp16 rows, plane 1 = per 128-k slice sixteen chunks 2 (4 which + 2 mm + x) + h; image 1: [N / 32][3 slices][32 chunk images][32 rows][16 B]."""
    from wild_visual_navigation_amd.backbone import mx_split, pack_a384_mx
    N = 64
    w = torch.randn(N, 384, generator=torch.Generator().manual_seed(1))
    h, l8, h8 = mx_split(w)
    p = pack_a384_mx(w)
    assert p.shape == (2, 2, N, 768) and p.dtype == torch.uint8
    hb = h.contiguous().view(torch.uint8).reshape(N, 768)
    assert torch.equal(p[0, 0], hb)
    planes = (l8, h8)
    img1 = p[1].reshape(N // 32, 3, 32, 32, 16)
    gen = torch.Generator().manual_seed(2)
    for _ in range(200):
        n, ks, which, mm, x, hh, sp, j = (int(torch.randint(0, m, (1,), generator=gen)) for m in (N, 3, 2, 2, 2, 2, 2, 8))
        k = 128 * ks + 64 * mm + 16 * (2 * x + sp) + 8 * hh + j
        byte = int(planes[which][n, k])
        assert int(p[0, 1, n, ks * 256 + 16 * (2 * (4 * which + 2 * mm + x) + hh) + 8 * sp + j]) == byte
        assert int(img1[n // 32, ks, 16 + 8 * mm + 4 * which + 2 * x + hh, n % 32, 8 * sp + j]) == byte
        c, jj = int(torch.randint(0, 16, (1,), generator=gen)), int(torch.randint(0, 16, (1,), generator=gen))
        assert int(img1[n // 32, ks, c, n % 32, jj]) == int(hb[n, 256 * ks + 16 * c + jj])


def test_pack_a768_fp8_byte_for_byte():
    """[N / 32 tiles][12 k-steps][2 halves x][64 lanes = (hi, row)][16 B]: byte j of lane (hi, row) in (s, x) = W[32 tile + row, 64 s + 32 hi + 16 x + j]."""
    from wild_visual_navigation_amd.backbone import pack_a768_fp8
    N = 96
    wq = torch.randint(0, 256, (N, 768), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    p = pack_a768_fp8(wq)
    assert p.shape == (N // 32, 12, 2, 2, 32, 16)
    gen = torch.Generator().manual_seed(4)
    for _ in range(300):
        t, s, x, hi, row, j = (int(torch.randint(0, m, (1,), generator=gen)) for m in (N // 32, 12, 2, 2, 32, 16))
        assert int(p[t, s, x, hi, row, j]) == int(wq[32 * t + row, 64 * s + 32 * hi + 16 * x + j])
