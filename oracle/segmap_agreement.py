"""Oracle (TEST INFRASTRUCTURE ONLY): what "bit-exact segment-index maps" can mean behind a floating-point backbone.

north_star asks for segment maps that equal the reference's bit for bit, and for tokens within 1e-3.  The k-means of the STEGO
stage (stego_interface.py:94-109) is an integer-valued function of float input: a pixel whose two best centroids are closer than
the float tolerance of the code may legitimately land on either side, so the end-to-end statement that CAN be checked is

    every pixel where the GPU's map differs from the oracle's lies within the float tolerance of a decision boundary.

Made rigorous: let x_o, x_g be the normalised code rows of the oracle / of the GPU path (fixed-order bilinear up-sampling of the
respective patch codes), c_o, c_g the final centroids of the deterministic k-means run on each, eps_x = max_p ||x_g - x_o||,
eps_c = max_k ||c_g,k - c_o,k||.  All vectors have unit length, so |<x_g, c_g,k> - <x_o, c_o,k>| <= eps_x + eps_c =: eps for every
(p, k).  If the GPU labels p with k_g while the oracle labels it k_o, then <x_g, c_g,kg> >= <x_g, c_g,ko>, hence

    margin(p) := <x_o, c_o,ko> - <x_o, c_o,kg>  <=  2 eps.

`analyse` measures eps and the margins of all mismatching pixels and reports whether the inequality holds (it must, as long as the
GPU's integer stage equals the oracle's on the GPU's code -- which the caller checks separately, bit for bit), plus the weaker
figures the round-3 verdict asked for: the agreement rate and the oracle's top-2 margin at the mismatching pixels against
4 x max|code_g - code_o|.  Cluster ids are comparable between the two runs because the initial centroids sit at fixed pixel
positions (oracle/interfaces.py::kmeans_cosine_labels); an id permutation would show up as an agreement near 1 / K, not as a pass.
"""
from typing import Dict, Optional

import numpy as np

from . import interfaces as OI


def kmeans_pixels_full(code_tokens: np.ndarray, G: int, H: int, K: int, iters: int = OI.KMEANS_ITERS, form: str = "linear", align_corners: bool = True):
    """labels [H*H] (not compacted), final centroids [K, C], normalised rows [H*H, C] of the pixel-resolution k-means of one frame in
    the statement `form` ("linear": oracle/kmeans_linear.py, the product default; "direct": oracle/interfaces.py).
    Needs the C restatements (oracle/_build/libwvn_oracle.so)."""
    h = OI._oracle_lib()
    if h is None:
        raise RuntimeError("oracle/_build/libwvn_oracle.so is not built (python -m oracle.build_oracle)")
    if form == "linear":
        from . import kmeans_linear

        r = kmeans_linear.kmeans_pixels_linear_c(np.asarray(code_tokens, dtype=np.float32), G, H, K, iters, want_rows=True, align_corners=align_corners)
        if r is None:
            raise RuntimeError("oracle/_build/libwvn_oracle.so predates oracle/kmeans_linear_ref.c (python -m oracle.build_oracle)")
        return r
    dense = np.ascontiguousarray(OI.upsample_bilinear_fixed(code_tokens.reshape(G, G, -1), H, align_corners).reshape(H * H, -1), dtype=np.float32)
    P, C = dense.shape
    labels = np.empty(P, dtype=np.int32)
    cent = np.empty((K, C), dtype=np.float32)
    x = np.empty((P, C), dtype=np.float32)
    if h.wvn_oracle_kmeans_cosine_ex(dense.ctypes.data, P, C, K, iters, labels.ctypes.data, cent.ctypes.data, x.ctypes.data) != 0:
        raise MemoryError("oracle k-means")
    return labels, cent, x


def analyse(ocode: np.ndarray, gcode: np.ndarray, G: int, H: int, K: int, glabels: Optional[np.ndarray] = None, form: str = "linear") -> Dict:
    """ocode / gcode: [G*G, C] fp32 patch codes of ONE frame from the oracle / from the GPU path.  glabels (optional): the GPU's
    uncompacted or compacted label map [H*H]; compared with the oracle k-means of gcode after ascending relabelling.  form: the
    statement of the k-means both runs use (the linear form's argmax is taken on ||v_p|| <x_p, c_k>: the same decision up to the
    fp32 rounding of a similarity, which the 2e-6 slack below covers)."""
    lo, co, xo = kmeans_pixels_full(np.asarray(ocode, dtype=np.float32), G, H, K, form=form)
    lg, cg, xg = kmeans_pixels_full(np.asarray(gcode, dtype=np.float32), G, H, K, form=form)
    out = {"pixels": int(lo.size)}
    if glabels is not None:
        out["gpu_integer_stage_exact"] = bool(np.array_equal(OI.relabel_ascending(lg), OI.relabel_ascending(np.asarray(glabels).reshape(-1))))
    eps_x = float(np.sqrt(((xg.astype(np.float64) - xo) ** 2).sum(1)).max())
    eps_c = float(np.sqrt(((cg.astype(np.float64) - co) ** 2).sum(1)).max())
    eps = eps_x + eps_c
    mism = np.nonzero(lo != lg)[0]
    out.update(agreement=float((OI.relabel_ascending(lo) == OI.relabel_ascending(lg)).mean()), agreement_raw_ids=float(1.0 - mism.size / lo.size),
               mismatching=int(mism.size), eps_x=eps_x, eps_c=eps_c, max_abs_code=float(np.abs(gcode - ocode).max()))
    if mism.size:
        sims = xo[mism].astype(np.float64) @ co.astype(np.float64).T                 # [m, K] oracle similarities of the mismatching pixels
        best = sims[np.arange(mism.size), lo[mism]]
        margin = best - sims[np.arange(mism.size), lg[mism]]                          # >= 0 up to the fp32 chain's rounding
        srt = np.sort(sims, axis=1)
        top2 = srt[:, -1] - srt[:, -2]
        out.update(max_margin=float(margin.max()), max_top2_margin=float(top2.max()),
                   margin_over_eps_hist=np.histogram(margin / max(eps, 1e-30), bins=[0, 0.01, 0.03, 0.1, 0.3, 1.0, 2.0, np.inf])[0].tolist())
    else:
        out.update(max_margin=0.0, max_top2_margin=0.0, margin_over_eps_hist=[0] * 7)
    out["bound_2eps"] = 2 * eps
    out["within_float_tolerance"] = bool(out["max_margin"] <= 2 * eps + 2e-6)          # (2e-6: fp64 dot against the fp32 fma chains / interpolated table)
    out["top2_within_4x_code_error"] = bool(out["max_top2_margin"] < 4 * out["max_abs_code"])
    return out
