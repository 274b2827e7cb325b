"""WVNMode -- wild_visual_navigation/utils/operation_modes.py:9-35."""
from enum import Enum


class WVNMode(Enum):
    DEBUG = 0
    ONLINE = 1
    EXTRACT_LABELS = 2

    def from_string(string):
        modes = {"debug": WVNMode.DEBUG, "online": WVNMode.ONLINE, "extract_labels": WVNMode.EXTRACT_LABELS}
        if string not in modes:
            raise ValueError("Invalid WVNMode string")
        return modes[string]
