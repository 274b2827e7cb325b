"""Mission-node store with the two queries ``TraversabilityEstimator.train`` needs
(graphs.py:137-143: get_num_valid_nodes / get_n_random_valid_nodes).  The reference's networkx
distance graph (edge thresholds, radius queries) is Python bookkeeping outside the hot path."""
import random
from threading import Lock
from typing import List


class MissionGraph:
    def __init__(self, max_elements: int = None):
        self._nodes: List = []
        self._lock = Lock()
        self._max = max_elements

    def add_node(self, node) -> bool:
        with self._lock:
            self._nodes.append(node)
            if self._max is not None and len(self._nodes) > self._max:
                self._nodes.pop(0)
        return True

    def get_nodes(self):
        with self._lock:
            return list(self._nodes)

    def get_num_nodes(self):
        return len(self._nodes)

    def get_valid_nodes(self):
        return [n for n in self.get_nodes() if n.is_valid()]

    def get_num_valid_nodes(self):
        return len(self.get_valid_nodes())

    def get_n_random_valid_nodes(self, n=None):
        nodes = self.get_valid_nodes()
        random.shuffle(nodes)
        return nodes if n is None else nodes[:n]
