"""Minimal PyG-like containers -- wild_visual_navigation/utils/data.py:11-58.

``Batch.from_data_list`` concatenates per-node tensors along dim 0 and offsets ``edge_index`` by the
node pointer.  Unlike the reference (which mutates and returns the *class*, data.py:39-58 -- not
thread-safe) an instance is returned; callers only read ``.x .y .y_valid .edge_index .ptr .batch .ba``."""
from typing import List

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


class Batch:
    @classmethod
    def from_data_list(cls, list_of_data: List[Data]):
        if len(list_of_data) == 0:
            return None
        out = cls()
        first = list_of_data[0]
        keys = ["x"] + [k for k in vars(first) if k[0] != "_" and k != "x" and getattr(first, k) is not None]
        sizes = [int(d.x.shape[0]) for d in list_of_data]
        ptr = [0]
        for s in sizes:
            ptr.append(ptr[-1] + s)
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
        for k in keys:
            if k == "edge_index":
                out.edge_index = torch.cat([getattr(d, k) + out.ptr[j].item() for j, d in enumerate(list_of_data)], dim=-1)
            else:
                setattr(out, k, torch.cat([getattr(d, k) for d in list_of_data], dim=0))
        out.ba = out.x.shape[0]
        return out
