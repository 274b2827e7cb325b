"""SegmentExtractor -- wild_visual_navigation/feature_extractor/segment_extractor.py:11-92 on HIP
kernels (bit-exact adjacency list, exact-integer centre sums)."""
import torch

from .. import ops


class SegmentExtractor(torch.nn.Module):
    def __init__(self):
        super().__init__()

    @torch.no_grad()
    def adjacency_list(self, seg: torch.Tensor) -> torch.Tensor:
        """seg [1,1,H,W] int64 -> [E,2] int64, sorted by left + right*(max+1) like the reference."""
        assert seg.shape[0] == 1 and len(seg.shape) == 4, f"{seg.shape}"
        n_seg = int(seg.max().item()) + 1  # same host sync the reference performs (`div = seg.max() + 1`)
        return ops.seg_adjacency(seg[0, 0], n_seg)

    @torch.no_grad()
    def centers(self, seg: torch.Tensor) -> torch.Tensor:
        """seg [1,1,H,W] -> [S,2] fp32 (x, y)."""
        assert seg.shape[0] == 1 and len(seg.shape) == 4
        n_seg = int(seg.max().item()) + 1
        return ops.seg_centers(seg[0, 0], n_seg)
