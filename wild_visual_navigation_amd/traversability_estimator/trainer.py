"""One optimisation step of the online traversability MLP on HIP kernels, with the two tiny
exchanges that make it data-parallel:

    phase A  forward + per-row reconstruction loss + LOCAL statistic {n_lab, sum, sum^2, R}  (fp64)
       -> all-reduce(sum) of 4 doubles      (the loss normalisers / confidence mean,std are GLOBAL)
    phase B  loss gradient + backward GEMMs -> flat gradient [n_param + 2] fp32
       -> all-reduce(sum) of 478 KB (D=384) / 138 KB (D=90)   -- RCCL over xGMI, latency-bound
    phase C  Adam + losses; every rank applies the identical update => replicas stay bit-identical

Local contributions are sums (not means) scaled by the global normalisers inside phase B, so the
N-GPU trajectory equals the 1-GPU trajectory on the concatenated batch up to fp32 summation order.
Reference arithmetic: traversability_estimator.py:464-477, loss.py:93-160, torch.optim.Adam.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from .. import _lib
from ..distributed import allreduce_sum_
from ..model.simple_mlp import SimpleMLP




class MlpTrainer:
    def __init__(self, model: SimpleMLP, lr: float = 1e-3, std_factor: float = 0.5, w_trav: float = 0.03,
                 w_reco: float = 0.5, process_group=None, fused: bool = True):
        self.model = model
        self.lr, self.std_factor, self.w_trav, self.w_reco = lr, std_factor, w_trav, w_reco
        self.group = process_group
        self.step = 0
        self._state_dev = None
        self.last_confidence: Optional[torch.Tensor] = None
        self.comm_events = None   # set to [] to collect (start, end) CUDA-event pairs around the two all-reduces of every step
        # True: the four-launch step of csrc/mlp_train.hip where its geometry applies (256 / 32 hidden units, <= 2048 rows);
        # False: the general path (one kernel per stage, 17-20 launches).  Same arithmetic, other summation orders.
        self.fused = fused

    def _state(self, dev):
        if self._state_dev != dev:
            n = self.model.flat_params().numel()
            self.grads = torch.zeros(n + 2, dtype=torch.float32, device=dev)
            if self._state_dev is not None:   # change_device in the middle of a mission: the Adam moments move with the model
                self.m, self.v = self.m.to(dev), self.v.to(dev)
            else:
                self.m = torch.zeros(n, dtype=torch.float32, device=dev)
                self.v = torch.zeros(n, dtype=torch.float32, device=dev)
            self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
            self.losses = torch.zeros(5, dtype=torch.float32, device=dev)
            self.sync_word = torch.zeros(1, dtype=torch.int32, device=dev)   # arrival counter of the fused forward (zero between steps)
            self._state_dev = dev

    # Adam moments in torch.optim.Adam.state_dict() shape, for save/load_checkpoint compatibility
    def optimizer_state_dict(self) -> Dict:
        """``torch.optim.Adam.state_dict()`` shape: empty ``state`` before the first step (as the reference's optimizer),
        and every param-group key ``Adam.load_state_dict`` of current and older torch versions expects."""
        ps = self.model._params_in_order()
        st, off = {}, 0
        if self._state_dev is not None and self.step > 0:
            for i, p in enumerate(ps):
                n = p.numel()
                st[i] = {"step": torch.tensor(float(self.step)), "exp_avg": self.m[off:off + n].view_as(p).clone(),
                         "exp_avg_sq": self.v[off:off + n].view_as(p).clone()}
                off += n
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(ps)))}
        return {"state": st, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        ps = self.model._params_in_order()
        self._state(ps[0].device)
        self.m.zero_()
        self.v.zero_()
        self.step = 0
        off = 0
        for i, p in enumerate(ps):
            n = p.numel()
            if i in sd["state"]:
                self.m[off:off + n] = sd["state"][i]["exp_avg"].reshape(-1).to(self.m.device)
                self.v[off:off + n] = sd["state"][i]["exp_avg_sq"].reshape(-1).to(self.v.device)
                self.step = int(sd["state"][i]["step"])
            off += n
        self.lr = sd["param_groups"][0]["lr"]

    def _timed_allreduce(self, t: torch.Tensor) -> None:
        if self.comm_events is None:
            allreduce_sum_(t, self.group)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        allreduce_sum_(t, self.group)   # RCCL runs on its own stream; the current stream waits for it, so b brackets it
        b.record()
        self.comm_events.append((a, b))

    @torch.no_grad()
    def train_step(self, x: torch.Tensor, y: torch.Tensor, y_valid: torch.Tensor,
                   want_confidence: bool = False, rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [R,D] fp32, y [R] fp32, y_valid [R] bool (this rank's rows).  Returns the device tensor
        losses[5] = {total, loss_trav, loss_reco, conf_mean, conf_std} (no host sync here).
        ``rows_dev`` (int32 [1] on the device): only the first rows_dev[0] of the R rows are real (a batch compacted by
        ``ops.compact_segment_rows``: its row count never visits the host); the rest contribute nothing."""
        _lib.require_cuda(x, "x")
        lib = _lib.lib()
        dev = x.device
        self._state(dev)
        x = x.float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        y = y.float().contiguous()
        yv = y_valid.to(torch.uint8).contiguous()
        R = x.shape[0]
        d = self.model.desc
        flat = self.model.flat_params()
        ws = self.model._workspace(max(R, 1))
        st = _lib.stream()
        conf = torch.empty(R, dtype=torch.float32, device=dev) if want_confidence else None

        # A rank whose shard is empty this step (ragged frame sharding) still takes part in both collectives, contributing
        # zeros: every rank makes the same sequence of RCCL calls whatever its row count.
        rd = _lib.ptr(rows_dev)
        if R > 0:
            _lib.check(lib.wvn_mlp_train_phase_a_rows(C.byref(d), flat.data_ptr(), x.data_ptr(), x.stride(0), yv.data_ptr(), R, rd,
                                                      self.stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                                      self.sync_word.data_ptr() if self.fused else 0, st), "phase_a")
        else:
            self.stats.zero_()
        self._timed_allreduce(self.stats)
        if R > 0:
            _lib.check(lib.wvn_mlp_train_phase_b_rows(C.byref(d), flat.data_ptr(), x.data_ptr(), x.stride(0), y.data_ptr(),
                                                      yv.data_ptr(), R, rd, self.stats.data_ptr(), self.std_factor, self.w_trav,
                                                      self.w_reco, self.grads.data_ptr(), _lib.ptr(conf), ws.data_ptr(),
                                                      ws.numel(), int(self.fused), st), "phase_b")
        else:
            self.grads.zero_()
        self._timed_allreduce(self.grads)
        self.step += 1
        _lib.check(lib.wvn_mlp_train_phase_c(C.byref(d), flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                             self.v.data_ptr(), self.step, self.lr, self.stats.data_ptr(), self.w_trav,
                                             self.w_reco, self.losses.data_ptr(), st), "phase_c")
        self.last_confidence = conf
        return self.losses
