"""SimpleMLP -- wild_visual_navigation/model/simple_mlp.py:10-39 on HIP kernels.

Same constructor, same ``state_dict`` keys (``layers.{0,2,4}.{weight,bias}``), same ``forward(Data)``
contract ([R, 1+D] with the sigmoid applied to column 0).  All six parameter tensors are views into
ONE flat fp32 buffer [W1|b1|W2|b2|W3|b3] -- the layout libwvn_hip's MLP entry points, the Adam kernel
and the data-parallel gradient all-reduce operate on.
"""
import ctypes as C
from typing import List, Optional

import torch

from .. import _lib
from ..utils.data import Data


class SimpleMLP(torch.nn.Module):
    def __init__(self, input_size: int = 64, hidden_sizes: List[int] = [255], reconstruction: bool = False):
        super().__init__()
        hidden_sizes = list(hidden_sizes)  # the reference mutates its argument (simple_mlp.py:21-22); we do not
        self.nr_sigmoid_layers = hidden_sizes[-1]
        if reconstruction:
            hidden_sizes[-1] = hidden_sizes[-1] + input_size
        if len(hidden_sizes) != 3 or self.nr_sigmoid_layers != 1 or not reconstruction:
            raise ValueError("the MI355X path implements the default SimpleMLP(D, [h1, h2, 1], reconstruction=True)")
        self.input_size = input_size
        layers, i = [], input_size
        for hs in hidden_sizes[:-1]:
            layers += [torch.nn.Linear(i, hs), torch.nn.ReLU()]
            i = hs
        layers.append(torch.nn.Linear(i, hidden_sizes[-1]))
        self.layers = torch.nn.Sequential(*layers)
        self.output_features = hidden_sizes[-1]
        self.desc = _lib.MlpDesc(input_size, hidden_sizes[0], hidden_sizes[1], 0)
        self._flat: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._pix_packed: Optional[torch.Tensor] = None
        self._pix_packed_x3: Optional[torch.Tensor] = None
        self._pix_ws: Optional[torch.Tensor] = None

    # ---- flat parameter storage --------------------------------------------------------------------
    def _params_in_order(self):
        return [self.layers[0].weight, self.layers[0].bias, self.layers[2].weight, self.layers[2].bias,
                self.layers[4].weight, self.layers[4].bias]

    def flat_params(self) -> torch.Tensor:
        """The flat buffer the six parameters alias (re-packed if .to()/load broke the aliasing)."""
        ps = self._params_in_order()
        ok = self._flat is not None and self._flat.device == ps[0].device
        if ok:
            off = 0
            for p in ps:
                if p.data_ptr() != self._flat.data_ptr() + 4 * off or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = flat[off: off + p.numel()].view_as(p)
                off += p.numel()
            self._flat = flat
        return self._flat

    def _workspace(self, rows: int) -> torch.Tensor:
        need = _lib.lib().wvn_mlp_workspace_bytes(C.byref(self.desc), rows)
        dev = self.layers[0].weight.device
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    # ---- forward --------------------------------------------------------------------------------------
    def forward(self, data: Data) -> torch.Tensor:
        x = data.x
        _lib.require_cuda(x, "data.x")
        x = x.float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        R = x.shape[0]
        flat = self.flat_params()
        out = torch.empty(R, self.output_features, dtype=torch.float32, device=x.device)
        ws = self._workspace(R)
        rc = _lib.lib().wvn_mlp_forward(C.byref(self.desc), flat.data_ptr(), x.data_ptr(), x.stride(0), R,
                                        out.data_ptr(), 0, 0, ws.data_ptr(), ws.numel(), _lib.stream())
        _lib.check(rc, "wvn_mlp_forward")
        return out

    # ---- fused per-pixel inference ---------------------------------------------------
    X_COL = 256   # include/wvn_hip.h: WVN_PIXEL_X_COL

    @property
    def ZX_COLS(self) -> int:
        """Row length of the zx buffer: 640 for 384-d DINO features, 384 for the 90-d STEGO code (0: unsupported size)."""
        return int(_lib.lib().wvn_pixel_mlp_zx_cols(C.byref(self.desc)))

    def pack_per_pixel(self) -> torch.Tensor:
        """Re-pack the current parameters for ``forward_per_pixel`` (bf16, MFMA fragment order).  Call after every
        parameter change (the node reloads weights at 1 Hz, wvn_feature_extractor_node.py:407-432)."""
        flat = self.flat_params()
        _lib.require_cuda(flat, "parameters")
        n = _lib.lib().wvn_pixel_mlp_pack_bytes(C.byref(self.desc))
        if n == 0:
            raise _lib.WvnError("fused per-pixel inference needs SimpleMLP(384 | 90, [256, 32, 1], reconstruction=True)")
        if self._pix_packed is None or self._pix_packed.device != flat.device:
            self._pix_packed = torch.empty(n, dtype=torch.uint8, device=flat.device)
        rc = _lib.lib().wvn_pixel_mlp_pack(C.byref(self.desc), flat.data_ptr(), self._pix_packed.data_ptr(), _lib.stream())
        _lib.check(rc, "wvn_pixel_mlp_pack")
        return self._pix_packed

    @torch.no_grad()
    def forward_per_pixel(self, zx: torch.Tensor, batch: int, grid: int, out_hw, mean: float = 0.0, std: float = 1.0,
                          std_factor: float = 0.5, want_loss: bool = False, repack: bool = True,
                          conf_state: Optional[torch.Tensor] = None):
        """What wvn_feature_extractor_node.py:319-363 computes with prediction_per_pixel -- upsample, forward, column 0,
        reconstruction confidence -- from the PATCH tokens, without the dense feature tensor:
        ``zx`` [batch*grid*grid, ZX_COLS] bf16 with the features from column 256 on (zero-padded to the row end for the 90-d
        STEGO code; columns [0, 256) are scratch) ->
        (trav [batch,H,W], conf [batch,H,W], loss_reco [batch,H,W] | None), fp32.  ``conf_state``: optional fp32 device
        tensor {mean, std, std_factor} read by the kernel instead of the three floats (for HIP-graph capture)."""
        _lib.require_cuda(zx, "zx")
        if zx.dtype != torch.bfloat16 or zx.dim() != 2 or zx.shape[0] != batch * grid * grid or zx.stride(1) != 1 \
                or zx.shape[1] < self.ZX_COLS:
            raise _lib.WvnError(f"zx must be bf16 [batch*grid*grid, >= {self.ZX_COLS}], got {tuple(zx.shape)} {zx.dtype}")
        packed = self.pack_per_pixel() if (repack or self._pix_packed is None) else self._pix_packed
        H, W = out_hw
        trav = torch.empty(batch, H, W, dtype=torch.float32, device=zx.device)
        conf = torch.empty_like(trav)
        loss = torch.empty_like(trav) if want_loss else None
        rc = _lib.lib().wvn_pixel_mlp_infer(C.byref(self.desc), packed.data_ptr(), zx.data_ptr(), zx.stride(0), batch, grid,
                                            H, W, float(mean), float(std), float(std_factor),
                                            conf_state.data_ptr() if conf_state is not None else 0, trav.data_ptr(),
                                            conf.data_ptr(), loss.data_ptr() if want_loss else 0, _lib.stream())
        _lib.check(rc, "wvn_pixel_mlp_infer")
        return trav, conf, loss

    @torch.no_grad()
    def forward_per_pixel_exact(self, tokens: torch.Tensor, batch: int, grid: int, out_hw, mean: float = 0.0,
                                std: float = 1.0, std_factor: float = 0.5, want_loss: bool = False,
                                conf_state: Optional[torch.Tensor] = None):
        """Exact-mode form of ``forward_per_pixel`` (hi + lo split MFMA operands, fp32 layer-1 GEMM at token resolution):
        ``tokens`` [batch*grid*grid, 384] fp32 (the backbone's final patch tokens) -> (trav, conf, loss_reco | None),
        within 1e-3 of the reference sequence on the dense fp32 features."""
        _lib.require_cuda(tokens, "tokens")
        if tokens.dtype != torch.float32 or tokens.dim() != 2 or tokens.shape[0] != batch * grid * grid or tokens.stride(1) != 1 \
                or tokens.shape[1] < self.input_size:
            raise _lib.WvnError(f"tokens must be fp32 [batch*grid*grid, >= {self.input_size}], got {tuple(tokens.shape)} {tokens.dtype}")
        h = _lib.lib()
        flat = self.flat_params()
        n = h.wvn_pixel_mlp_exact_pack_bytes(C.byref(self.desc))
        if n == 0:
            raise _lib.WvnError("fused per-pixel inference needs SimpleMLP(384, [256, 32, 1], reconstruction=True)")
        if self._pix_packed_x3 is None or self._pix_packed_x3.device != flat.device:
            self._pix_packed_x3 = torch.empty(n, dtype=torch.uint8, device=flat.device)
        _lib.check(h.wvn_pixel_mlp_exact_pack(C.byref(self.desc), flat.data_ptr(), self._pix_packed_x3.data_ptr(), _lib.stream()),
                   "wvn_pixel_mlp_exact_pack")
        need = h.wvn_pixel_mlp_exact_workspace_bytes(C.byref(self.desc), batch, grid)
        if self._pix_ws is None or self._pix_ws.numel() < need or self._pix_ws.device != flat.device:
            self._pix_ws = torch.empty(need, dtype=torch.uint8, device=flat.device)
        H, W = out_hw
        trav = torch.empty(batch, H, W, dtype=torch.float32, device=tokens.device)
        conf = torch.empty_like(trav)
        loss = torch.empty_like(trav) if want_loss else None
        rc = h.wvn_pixel_mlp_infer_exact(C.byref(self.desc), flat.data_ptr(), self._pix_packed_x3.data_ptr(), tokens.data_ptr(),
                                         tokens.stride(0), batch, grid, H, W, float(mean), float(std), float(std_factor),
                                         conf_state.data_ptr() if conf_state is not None else 0, trav.data_ptr(), conf.data_ptr(),
                                         loss.data_ptr() if want_loss else 0, self._pix_ws.data_ptr(), self._pix_ws.numel(),
                                         _lib.stream())
        _lib.check(rc, "wvn_pixel_mlp_infer_exact")
        return trav, conf, loss
