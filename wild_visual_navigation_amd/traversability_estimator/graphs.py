"""Mission / supervision graphs -- wild_visual_navigation/traversability_estimator/graphs.py:14-316: ``BaseGraph`` (nodes
chained by distance-weighted edges, rejected when closer than ``edge_distance`` to the last one), ``MaxElementsGraph``,
``TemporalWindowGraph`` and ``DistanceWindowGraph`` with the reference's eviction rules and queries.  The reference builds on
networkx (an undirected graph keyed by node objects, insertion-ordered); this restates the handful of operations it uses --
insertion-ordered node dict, adjacency dict, Dijkstra with a cutoff -- so networkx is not a dependency.  Python bookkeeping at
camera / robot-state rate; no kernel work here."""
import heapq
import random
from threading import Lock


class BaseGraph:
    def __init__(self, edge_distance: float = 0.0):
        self._nodes = {}        # node -> attribute dict, insertion-ordered (like nx.Graph._node)
        self._adj = {}          # node -> {neighbour: distance}
        self._first_node = None
        self._last_added_node = None
        self._edge_distance = edge_distance
        self._lock = Lock()

    def __str__(self):
        return f"Graph with {len(self._nodes)} nodes and {self.get_num_edges()} edges"

    def __getstate__(self):
        state = self.__dict__.copy()
        del state["_lock"]
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._lock = Lock()

    def change_device(self, device):
        for n in self._nodes:
            n.change_device(device)

    def add_node(self, node) -> bool:
        """graphs.py:57-86: skip nodes closer than edge_distance to the last added one; chain the rest."""
        with self._lock:
            d = None
            if self._last_added_node is not None:
                d = float(node.distance_to(self._last_added_node))
                if self._edge_distance is not None and d < self._edge_distance:
                    return False
            if node not in self._nodes:
                self._nodes[node] = {"timestamp": node.timestamp}
                self._adj[node] = {}
            if self._last_added_node is not None and self._last_added_node in self._nodes:
                self._adj[node][self._last_added_node] = d
                self._adj[self._last_added_node][node] = d
            elif self._last_added_node is None:
                self._first_node = node
        self._last_added_node = node
        return True

    def add_edge(self, node1, node2):
        with self._lock:
            d = float(node1.distance_to(node2))
            self._adj[node1][node2] = d
            self._adj[node2][node1] = d
        return True

    def clear(self):
        with self._lock:
            self._nodes.clear()
            self._adj.clear()

    def get_first_node(self):
        return self._first_node

    def get_last_node(self):
        return self._last_added_node

    def get_previous_node(self, node):
        prev = [k for k in self._adj.get(node, {}) if k < node]
        return prev[0] if len(prev) == 1 else None

    def get_num_nodes(self):
        with self._lock:
            return len(self._nodes)

    def get_num_valid_nodes(self):
        with self._lock:
            return sum(bool(n.is_valid()) for n in self._nodes)

    def get_num_edges(self):
        return sum(len(v) for v in self._adj.values()) // 2

    def get_nodes(self):
        with self._lock:
            return sorted(self._nodes)

    def get_valid_nodes(self):
        with self._lock:
            return sorted(n for n in self._nodes if n.is_valid())

    def get_n_random_valid_nodes(self, n=None):
        nodes = self.get_valid_nodes()
        random.shuffle(nodes)
        return nodes if n is None else nodes[:n]

    def get_node_with_timestamp(self, timestamp: float, eps: float = 1e-12):
        with self._lock:
            nodes = sorted(n for n in self._nodes if abs(n.timestamp - timestamp) < eps)
        return nodes[0] if nodes else None

    def get_nodes_within_radius_range(self, node, min_radius: float, max_radius: float, time_eps: float = 1,
                                      metric: str = "dijkstra"):
        """graphs.py:152-182.  "dijkstra": every node whose shortest-path length (edge weight = distance) from the node with
        the matching timestamp is <= max_radius, the query node excluded; "pose": min_radius <= |distance_to| < max_radius."""
        closest = self.get_node_with_timestamp(node.timestamp, eps=time_eps)
        nodes = []
        try:
            with self._lock:
                if metric == "dijkstra":
                    if closest is None or closest not in self._adj:
                        raise KeyError("query node not in graph")
                    dist = {closest: 0.0}
                    heap = [(0.0, id(closest), closest)]
                    done = set()
                    while heap:
                        d, _, u = heapq.heappop(heap)
                        if u in done:
                            continue
                        done.add(u)
                        for v, w in self._adj[u].items():
                            nd = d + w
                            if nd <= max_radius and nd < dist.get(v, float("inf")):
                                dist[v] = nd
                                heapq.heappush(heap, (nd, id(v), v))
                    nodes = [n for n in dist if n is not closest]
                elif metric == "pose":
                    nodes = [o for o in self._nodes if min_radius <= abs(float(o.distance_to(node))) < max_radius]
        except Exception as e:
            print(f"[get_nodes_within_radius_range] Exception: {e}")
        return sorted(nodes)

    def get_nodes_within_timespan(self, t_ini: float, t_end: float, open_interval: bool = False):
        with self._lock:
            if open_interval:
                return [n for n in self._nodes if t_ini < n.timestamp < t_end]
            return [n for n in self._nodes if t_ini <= n.timestamp <= t_end]

    def remove_nodes(self, nodes: list):
        with self._lock:
            for n in nodes:
                if n in self._nodes:
                    for nb in self._adj.pop(n):
                        self._adj[nb].pop(n, None)
                    del self._nodes[n]

    def remove_nodes_within_radius_range(self, node, min_radius: float = 0, max_radius: float = float("inf"),
                                         metric: str = "dijkstra"):
        """graphs.py:205-223: walk the nodes in insertion order and drop them while they are farther than min_radius
        (translation distance) from ``node``; stop at the first one that is not."""
        import torch

        to_remove = []
        for n in list(self._nodes.keys()):
            if torch.linalg.norm(n.pose_base_in_world[:3, 3].cpu() - node.pose_base_in_world[:3, 3].cpu()) > min_radius:
                to_remove.append(n)
            else:
                break
        self.remove_nodes(to_remove)

    def remove_nodes_within_timestamp(self, t_ini: float, t_end: float):
        self.remove_nodes(self.get_nodes_within_timespan(t_ini, t_end, open_interval=False))


class MaxElementsGraph(BaseGraph):
    """graphs.py:232-259: a FIFO of at most max_elements nodes."""

    def __init__(self, edge_distance: float = None, max_elements: int = -1):
        super().__init__(edge_distance=edge_distance)
        self._max_elements = max_elements

    def add_node(self, node):
        out = super().add_node(node)
        if len(self._nodes) > self._max_elements:
            self.remove_nodes([next(iter(self._nodes))])
        return out


class TemporalWindowGraph(BaseGraph):
    """graphs.py:262-285."""

    def __init__(self, edge_distance: float = None, time_window: float = float("inf")):
        super().__init__(edge_distance=edge_distance)
        self._time_window = time_window

    def add_node(self, node):
        out = super().add_node(node)
        self.remove_nodes_within_timestamp(0, node.timestamp - self._time_window)
        return out


class DistanceWindowGraph(BaseGraph):
    """graphs.py:288-316: keeps the nodes within max_distance of the newest one."""

    def __init__(self, edge_distance: float = None, max_distance: float = float("inf")):
        super().__init__(edge_distance=edge_distance)
        self._max_distance = max_distance

    @property
    def max_distance(self):
        return self._max_distance

    def add_node(self, node):
        out = super().add_node(node)
        self.remove_nodes_within_radius_range(node, min_radius=self._max_distance, max_radius=float("inf"), metric="pose")
        return out


MissionGraph = BaseGraph  # round-1 name of the mission-node store


def _chain(graph, n: int):
    """n BaseNodes, one per second, 0.1 m apart along x, added in order."""
    import torch

    from .nodes import BaseNode

    nodes = []
    for i in range(n):
        T = torch.eye(4)
        T[0, 3] = i / 10.0
        nodes.append(BaseNode(timestamp=i, pose_base_in_world=T))
        graph.add_node(nodes[-1])
    return nodes


def run_base_graph():
    """The reference's graph self-check (graphs.py:318-369, called by its tests/test_traversability_estimator.py): ten chained
    states; node list, counts, radius query, time-span queries (open / closed), removal, mutability of the stored nodes."""
    graph = BaseGraph()
    nodes = _chain(graph, 10)
    assert nodes == graph.get_nodes()
    assert graph.get_num_nodes() == 10 and graph.get_num_edges() == 9
    query = graph.get_node_with_timestamp(5.0)
    for n in graph.get_nodes_within_radius_range(query, min_radius=0, max_radius=0.2):
        assert float(query.distance_to(n)) <= 0.2 + 1e-6
    assert len(graph.get_nodes_within_timespan(0.0, 3.0, open_interval=True)) == 2
    closed = graph.get_nodes_within_timespan(0.0, 3.0, open_interval=False)
    assert len(closed) == 4
    graph.remove_nodes(closed)
    assert graph.get_num_nodes() == 6
    for n in graph.get_nodes():
        before = n.timestamp
        n.timestamp = 2
        assert before != n.timestamp


def run_temporal_window_graph():
    """graphs.py:372-392: fifty states into a 25-second window; the oldest STORED state is never older than the window.  (The
    reference's version asserts this on ``get_first_node()``, which it sets on the first insertion only and never moves -- that
    function is not called by the reference's tests and fails as written; the stored-node form is what the class guarantees.)"""
    import torch

    from .nodes import BaseNode

    graph = TemporalWindowGraph(edge_distance=0.0, time_window=25)
    for i in range(50):
        T = torch.eye(4)
        T[0, 3] = i / 10.0
        graph.add_node(BaseNode(timestamp=i, pose_base_in_world=T))
        assert graph.get_nodes()[0].timestamp >= i - 25
