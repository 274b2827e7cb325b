"""Oracle (TEST INFRASTRUCTURE ONLY): the pixel-resolution cosine k-means of the STEGO stage, stated through its LINEARITY.

The points of that k-means are x_p = rinv_p * sum_t w_{p,t} code_t: the H x H bilinearly up-sampled (align_corners=True) patch
codes, normalised (stego_interface.py:94-100 as this build reads it; oracle/interfaces.py::kmeans_cosine_labels_pixels is the
direct statement).  Two facts remove ~95 % of its arithmetic without changing what it computes:

  * rinv_p > 0 does not move an argmax and <., c_k> is linear in the four taps, so
        argmax_k <x_p, c_k> = argmax_k  sum_t w_{p,t} S[t, k],      S = code . c^T          ([G*G, K]: one small product per pass)
    -- per pixel a bilinear interpolation of K similarities instead of C channels followed by K dot products of length C;
  * the centroid sums are linear in the code as well:
        sum_{p in k} x_p = sum_t A[k, t] code_t,                   A[k, t] = sum_{p in k} rinv_p w_{p,t}
    -- a [K, G*G] table of summed tap weights per pass (the identity SURVEY.md 8(a7) states for segment pooling), then one small
    product, instead of re-creating and adding H*H rows of C channels.

WHAT THIS DEFINITION IS ANCHORED TO: its operation ORDER -- ROW_GROUP, the band chains, P0 before P1 -- was chosen together with
csrc/stego_linear.hip's tiling and changes when that tiling does (round 5 changed ROW_GROUP 4 -> 2 in the commit that re-tiled the kernel), so
"the GPU is bit-exact against this file" says the kernel computes THIS order, not that the order is canonical.  What the pair is held to is the
DIRECT statement (oracle/interfaces.py::kmeans_cosine_labels_pixels, which has no tiling to tune): identical points bit for bit, final centroids
within fp32 summation noise, and label maps that differ only within the float tolerance of a decision boundary --
tests/test_oracle_stego.py::test_linear_and_direct_statements_at_the_headline_shape_pinned_numbers pins the numbers at G = 56, H = 448, K = 20
(0 of 200 704 pixels, centroid distance 2.6e-7): re-run it whenever this file is edited.

The two statements are the same function of exact arithmetic; in fp32 they round differently, so their label maps can differ at
pixels whose two best similarities are closer than the rounding error (tests/test_oracle_stego.py holds the two against each
other with oracle/segmap_agreement.py's tolerance).  THIS file fixes every operation order of the linear form; csrc/stego_linear.hip
follows it operation for operation, so the GPU's labels and centroids can be compared bit for bit.

Definition (all fp32, one rounding per stated operation; `chain` = acc = fma(a, b, acc) from +0 in ascending index order):
  taps      o -> (i0, i1, w0, w1) of oracle/interfaces.py::upsample_bilinear_fixed (ATen's align_corners coordinates)
  rinv_p    1 / max(sqrt(chain_d v_d v_d), 1e-12), v = the fixed-order bilinear interpolation of the code at p   [as the direct form]
  c_k^0     v * rinv at pixel floor((2k+1) P / (2K))                                                            [as the direct form]
  pass:     S[t, k]    = chain_d code[t, d] c_k[d]
            sim[p, k]  = bilerp_fixed(S[., k]) at p;   label_p = first maximum over k
            u0 = rinv_p * wx0(x), u1 = rinv_p * wx1(x)
            U[y, k, j] = sum (plain adds from +0) over the pixels x of row y with label k, ascending x, of u0 if j0(x) == j, then u1 if j1(x) == j
            band b = the rows y with i0(y) == b:   P0[b, k, j] = chain_y wy0(y) U[y, k, j],   P1[b, k, j] = chain_y wy1(y) U[y, k, j]
            A[k, i, j] = plain adds from +0, bands ascending, P0 before P1, of every P_s[b] whose patch row (b for s = 0, i1 of the band for s = 1) is i
            R[i, k, d] = chain_j A[k, i, j] code[i, j, d]
            Q[g, k, d] = plain adds from +0 over the patch rows i of group g (ROW_GROUP = 2 consecutive rows), ascending, of R[i, k, d]
            sums[k, d] = plain adds from +0 over g ascending of Q[g, k, d]
            c_k        = sums_k * (1 / max(sqrt(chain_d sums_k[d]^2), 1e-12)) if the cluster has members, else unchanged
  final labels = one more assignment.
"""
from typing import Tuple

import numpy as np

from . import interfaces as OI

f32 = np.float32
ROW_GROUP = 2   # patch rows whose row sums are added before the groups are (csrc/stego_linear.hip: LIN_RG)


def taps(G: int, H: int, align_corners: bool = True):
    """(i0, i1, w0, w1) per output index -- the coordinates of upsample_bilinear_fixed (csrc/common.h: lerp_tap / lerp_tap_ac)."""
    return OI.bilinear_taps_fixed(G, H, align_corners)


def pixel_rinv_and_init(code: np.ndarray, G: int, H: int, K: int, align_corners: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """rinv [H, H] and the initial centroids [K, C]: exactly the direct form's (the dense rows exist here once, never in a pass)."""
    dense = OI.upsample_bilinear_fixed(code.reshape(G, G, -1), H, align_corners)        # [H, H, C]
    n2 = OI._seq_dot_f32(dense, dense)
    n = np.maximum(np.sqrt(n2).astype(np.float32), f32(1e-12))
    rinv = (f32(1.0) / n).astype(np.float32)
    P = H * H
    init = [((2 * k + 1) * P) // (2 * K) for k in range(K)]
    flat = dense.reshape(P, -1)
    cent = (flat[init] * rinv.reshape(P)[init][:, None]).astype(np.float32)
    return rinv, cent


def assign(code: np.ndarray, cent: np.ndarray, G: int, H: int, align_corners: bool = True) -> np.ndarray:
    S = OI._seq_dot_f32(code[:, None, :], cent[None, :, :])                              # [G*G, K]
    sim = OI.upsample_bilinear_fixed(S.reshape(G, G, -1), H, align_corners)              # [H, H, K]: the same fixed-order interpolation
    return np.argmax(sim, axis=2).astype(np.int32)                                       # first maximum == lowest index


def centroid_sums(code: np.ndarray, lab: np.ndarray, rinv: np.ndarray, G: int, H: int, K: int, align_corners: bool = True) -> np.ndarray:
    i0, i1, w0, w1 = taps(G, H, align_corners)
    C = code.shape[1]
    ys = np.arange(H)
    U = np.zeros((H, K, G), dtype=np.float32)
    for x in range(H):                                                                   # ascending x; tap 0 before tap 1
        U[ys, lab[:, x], i0[x]] = (U[ys, lab[:, x], i0[x]] + (rinv[:, x] * w0[x]).astype(np.float32)).astype(np.float32)
        U[ys, lab[:, x], i1[x]] = (U[ys, lab[:, x], i1[x]] + (rinv[:, x] * w1[x]).astype(np.float32)).astype(np.float32)
    P0 = np.zeros((G, K, G), dtype=np.float32)
    P1 = np.zeros((G, K, G), dtype=np.float32)
    for y in range(H):                                                                   # ascending y inside every band
        b = i0[y]
        P0[b] = OI._fma32(np.full_like(U[y], w0[y]), U[y], P0[b])
        P1[b] = OI._fma32(np.full_like(U[y], w1[y]), U[y], P1[b])
    A = np.zeros((K, G, G), dtype=np.float32)
    for b in range(G):
        A[:, b, :] = (A[:, b, :] + P0[b]).astype(np.float32)
        ib = b + (1 if b < G - 1 else 0)
        A[:, ib, :] = (A[:, ib, :] + P1[b]).astype(np.float32)
    cmap = code.reshape(G, G, C)
    R = np.zeros((G, K, C), dtype=np.float32)                                            # [i, k, d]
    for j in range(G):
        a = np.ascontiguousarray(A[:, :, j].T)[:, :, None]                              # [i, k, 1]
        R = OI._fma32(np.broadcast_to(a, R.shape), np.broadcast_to(cmap[:, j, None, :], R.shape), R)
    sums = np.zeros((K, C), dtype=np.float32)
    for i0_ in range(0, G, ROW_GROUP):                                                   # two levels: rows inside a group, then the groups
        q = np.zeros((K, C), dtype=np.float32)
        for i in range(i0_, min(G, i0_ + ROW_GROUP)):
            q = (q + R[i]).astype(np.float32)
        sums = (sums + q).astype(np.float32)
    return sums


def kmeans_pixels_linear_c(code_tokens: np.ndarray, G: int, H: int, K: int, iters: int = OI.KMEANS_ITERS, want_rows: bool = False,
                           align_corners: bool = True):
    """The C restatement (oracle/kmeans_linear_ref.c in oracle/_build/libwvn_oracle.so): (labels, centroids[, normalised rows [H*H, C]])
    or None when the library has not been built."""
    h = OI._oracle_lib()
    if h is None or not hasattr(h, "wvn_oracle_kmeans_pixels_linear_ac"):
        return None
    code = np.ascontiguousarray(code_tokens, dtype=np.float32)
    C = code.shape[1]
    labels = np.empty(H * H, dtype=np.int32)
    cent = np.empty((K, C), dtype=np.float32)
    x = np.empty((H * H, C), dtype=np.float32) if want_rows else None
    if h.wvn_oracle_kmeans_pixels_linear_ac(code.ctypes.data, G, H, C, K, iters, labels.ctypes.data, cent.ctypes.data,
                                            x.ctypes.data if want_rows else None, 1 if align_corners else 0) != 0:
        raise MemoryError("oracle k-means (linear form)")
    return (labels, cent, x) if want_rows else (labels, cent)


def kmeans_pixels_linear(code_tokens: np.ndarray, G: int, H: int, K: int, iters: int = OI.KMEANS_ITERS, force_numpy: bool = False,
                         align_corners: bool = True):
    """code_tokens [G*G, C] fp32 -> (labels int32 [H*H] not compacted, final centroids [K, C] fp32).  Through the C restatement when
    it is built (tests/test_oracle_stego.py holds it against the numpy statement below), else -- and with force_numpy -- the numpy
    statement itself."""
    if not force_numpy:
        r = kmeans_pixels_linear_c(code_tokens, G, H, K, iters, align_corners=align_corners)
        if r is not None:
            return r
    code = np.ascontiguousarray(code_tokens, dtype=np.float32)
    rinv, cent = pixel_rinv_and_init(code, G, H, K, align_corners)
    for _ in range(iters):
        lab = assign(code, cent, G, H, align_corners)
        sums = centroid_sums(code, lab, rinv, G, H, K, align_corners)
        cnt = np.bincount(lab.reshape(-1), minlength=K)
        cent = np.where((cnt > 0)[:, None], OI._normalize_rows_f32(sums), cent).astype(np.float32)
    return assign(code, cent, G, H, align_corners).reshape(-1), cent


def kmeans_cosine_labels_pixels_linear(code_tokens: np.ndarray, G: int, H: int, K: int, iters: int = OI.KMEANS_ITERS, align_corners: bool = True) -> np.ndarray:
    return kmeans_pixels_linear(code_tokens, G, H, K, iters, align_corners=align_corners)[0]
