"""The two process-boundary mechanisms between WVN's feature-extractor process (A) and learning process (B), SURVEY.md 8f-4:

1. ``ImageFeatures`` wire format (wvn_feature_extractor_node.py:373-393 -> wvn_learning_node.py:651-656).  ``pack_image_features``
   lays (feat, seg) out on the GPU exactly as the ROS message carries them (csrc/wire.hip) and brings ONE contiguous buffer to
   the host; ``ImageFeaturesPacket`` exposes the byte arrays / layout fields to fill the message with (zero-copy numpy views) and
   ``unpack_image_features`` is the learner's side.  ``reference_roundtrip`` is the reference's own encode / decode (Python list of
   floats) restated for the parity test.

2. Weights hand-off B -> A (wvn_learning_node.py:381-394 writes ``.tmp_state_dict.pt`` at 1 Hz, wvn_feature_extractor_node.py:
   407-442 polls it).  ``FileWeightsHandoff`` keeps that file protocol (same keys, incl. ``confidence_generator: {mean, var, std}``)
   but writes atomically (temp file + rename; the reference removes the old file first, so a reader can find nothing or a partial
   file).  ``DeviceWeightsHandoff`` is the single-process mode SURVEY.md asks for: extractor and learner share the GPU, the
   119 489 parameters + 3 confidence scalars go through a device-resident double buffer with a version counter -- a 478 KB
   device copy ordered by an event, no file, no host sync."""
import os
import tempfile
import threading
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

WIRE_HEADER = 64
WIRE_MAGIC = 0x464E5657


class ImageFeaturesPacket:
    """Host image of one ``ImageFeatures`` message: ``buffer`` (uint8, pinned when it came from ``pack_image_features``)."""

    def __init__(self, buffer: np.ndarray):
        self.buffer = buffer
        hdr = np.frombuffer(buffer[:WIRE_HEADER].tobytes(), dtype=np.uint32)
        if int(hdr[0]) != WIRE_MAGIC:
            raise _lib.WvnError("not an ImageFeatures wire buffer")
        ints = hdr.view(np.int32)
        self.H, self.W, self.S, self.D = int(ints[2]), int(ints[3]), int(ints[4]), int(ints[5])
        self._seg_off, self._feat_off = int(hdr[6]), int(hdr[7])

    # -- sensor_msgs/Image feature_segments (numpy_to_ros_image(seg.astype(np.int32), "passthrough")) --
    @property
    def segments(self) -> np.ndarray:
        return self.buffer[self._seg_off: self._seg_off + 4 * self.H * self.W].view(np.int32).reshape(self.H, self.W)

    def image_fields(self) -> Dict:
        return {"height": self.H, "width": self.W, "encoding": "32SC1", "is_bigendian": 0, "step": 4 * self.W,
                "data": self.buffer[self._seg_off: self._seg_off + 4 * self.H * self.W]}

    # -- std_msgs/Float32MultiArray features --
    @property
    def features(self) -> np.ndarray:
        return self.buffer[self._feat_off: self._feat_off + 4 * self.S * self.D].view(np.float32).reshape(self.S, self.D)

    def multiarray_fields(self) -> Dict:
        return {"dim": [{"label": "n", "size": self.S, "stride": self.S * self.D},
                        {"label": "feat", "size": self.D, "stride": self.D}], "data_offset": 0,
                "data": self.features.reshape(-1)}      # float32 array: what the wire carries (no Python list of floats)


def pack_image_features(feat: torch.Tensor, seg: torch.Tensor) -> ImageFeaturesPacket:
    """feat [S,D] fp32 (row stride allowed), seg [H,W] int64 / int32, both on the GPU -> host packet (one D2H copy)."""
    _lib.require_cuda(feat, "feat")
    _lib.require_cuda(seg, "seg")
    if seg.dtype not in (torch.int64, torch.int32) or not seg.is_contiguous() or feat.dtype != torch.float32 or feat.stride(1) != 1:
        raise _lib.WvnError("pack_image_features: feat fp32 [S,D] (unit column stride), seg contiguous int64 / int32 [H,W]")
    S, D = feat.shape
    H, W = seg.shape
    n = _lib.lib().wvn_wire_bytes(H, W, S, D)
    dev_buf = torch.empty(n, dtype=torch.uint8, device=feat.device)
    _lib.check(_lib.lib().wvn_wire_pack(seg.data_ptr(), int(seg.dtype == torch.int64), feat.data_ptr(), feat.stride(0),
                                        dev_buf.data_ptr(), H, W, S, D, _lib.stream()), "wvn_wire_pack")
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    host.copy_(dev_buf, non_blocking=True)
    torch.cuda.current_stream().synchronize()   # the publisher hands the bytes to the transport next
    return ImageFeaturesPacket(host.numpy())


def unpack_image_features(packet, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Learner side: packet (or its raw uint8 buffer) -> (features [S,D] fp32, feature_segments [H,W] int64) on ``device``."""
    if not isinstance(packet, ImageFeaturesPacket):
        packet = ImageFeaturesPacket(np.frombuffer(bytes(packet), dtype=np.uint8) if not isinstance(packet, np.ndarray) else packet)
    dev = torch.device(device)
    host = np.ascontiguousarray(packet.buffer)
    if not host.flags.writeable:     # bytes handed over by a transport: torch wants a writable array to wrap
        host = host.copy()
    buf = torch.from_numpy(host).to(dev, non_blocking=True)
    feat = torch.empty(packet.S, packet.D, dtype=torch.float32, device=dev)
    seg = torch.empty(packet.H, packet.W, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().wvn_wire_unpack(buf.data_ptr(), seg.data_ptr(), 0, feat.data_ptr(), packet.H, packet.W, packet.S,
                                          packet.D, _lib.stream()), "wvn_wire_unpack")
    return feat, seg


def reference_roundtrip(feat_np: np.ndarray, seg_np: np.ndarray):
    """What the reference does end to end (publisher :376-391, subscriber :651-656), on host arrays: for the parity test."""
    seg_msg = seg_np.astype(np.int32)
    data = feat_np.flatten().tolist()                                        # Float32MultiArray.data (serialised as float32)
    dims = (feat_np.shape[0], feat_np.shape[1])
    wire = np.asarray(data, dtype=np.float32)                                # what actually travels
    feat_back = np.array(wire.tolist(), dtype=float).reshape(dims).astype(np.float32)
    return feat_back, seg_msg


# ------------------------------------------------------------------------------------------------------------------------
class FileWeightsHandoff:
    """``.tmp_state_dict.pt`` protocol of wvn_learning_node.py:381-394 / wvn_feature_extractor_node.py:407-442."""

    def __init__(self, root_dir: str, name: str = ".tmp_state_dict.pt"):
        self.path = os.path.join(root_dir, name)

    def publish(self, model: torch.nn.Module, confidence_generator) -> str:
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        sd["confidence_generator"] = {k: v.detach().cpu() for k, v in confidence_generator.get_dict().items()}
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(self.path), prefix=".tmp_state_dict.", suffix=".part")
        os.close(fd)
        torch.save(sd, tmp)
        os.replace(tmp, self.path)       # atomic: a polling reader sees the old file or the new one, never none / half
        return self.path

    def consume(self, model: torch.nn.Module, confidence_generator) -> bool:
        """Loads if the file exists and its weights differ (load_model, :421-437).  Returns True when something was loaded."""
        if not os.path.exists(self.path):
            return False
        new = torch.load(self.path, map_location="cpu", weights_only=False)
        k = list(model.state_dict().keys())[-1]
        if k not in new or not (model.state_dict()[k].cpu() != new[k]).any():
            return False
        model.load_state_dict({a: b for a, b in new.items() if a != "confidence_generator"}, strict=False)
        cg = new.get("confidence_generator")
        if cg is not None:
            with torch.no_grad():
                confidence_generator.var.copy_(cg["var"])
                confidence_generator.mean.copy_(cg["mean"])
                confidence_generator.std.copy_(cg["std"])
        return True


class DeviceWeightsHandoff:
    """Single-process hand-off through GPU memory: ``publish`` (learner thread / stream) copies the flat parameter buffer and the
    confidence state into the back slot of a double buffer and flips the version; ``consume`` (extractor thread / stream) copies
    the front slot into its own model when the version moved.  Stream-ordered through an event; no file, no host round trip."""

    def __init__(self, n_params: int, device):
        self.dev = torch.device(device)
        self.slots = torch.zeros(2, n_params + 3, dtype=torch.float32, device=self.dev)
        self.version = 0
        self._seen = 0
        self._ready: Optional[torch.cuda.Event] = None
        self._read_done = [None, None]     # per slot: event after the consumer's last copy out of it
        self._lock = threading.Lock()      # version / events change together (publisher and consumer may be different threads)

    def publish(self, model, confidence_generator) -> int:
        with self._lock:   # held while the copies are ENQUEUED (microseconds): slot choice, events and version move together
            back = (self.version + 1) & 1
            if self._read_done[back] is not None:   # a consumer copy out of this slot may still be in flight on another stream
                torch.cuda.current_stream().wait_event(self._read_done[back])
            n = self.slots.shape[1] - 3
            with torch.no_grad():
                self.slots[back, :n].copy_(model.flat_params())
                self.slots[back, n:n + 1].copy_(confidence_generator.mean.reshape(-1))
                self.slots[back, n + 1:n + 2].copy_(confidence_generator.var.reshape(-1))
                self.slots[back, n + 2:n + 3].copy_(confidence_generator.std.reshape(-1))
            self._ready = torch.cuda.Event()
            self._ready.record()
            self.version += 1
            return self.version

    def consume(self, model, confidence_generator) -> bool:
        with self._lock:
            if self.version == self._seen or self._ready is None:
                return False
            torch.cuda.current_stream().wait_event(self._ready)
            front = self.version & 1
            n = self.slots.shape[1] - 3
            with torch.no_grad():
                model.flat_params().copy_(self.slots[front, :n])
                confidence_generator.mean.copy_(self.slots[front, n:n + 1])
                confidence_generator.var.copy_(self.slots[front, n + 1:n + 2].reshape(1, 1))
                confidence_generator.std.copy_(self.slots[front, n + 2:n + 3])
            self._read_done[front] = torch.cuda.Event()
            self._read_done[front].record()
            self._seen = self.version
            return True
