"""Federation topology generator (reference: fedstellar/utils/topologymanager.py:16-355).

Own numpy implementation (networkx/matplotlib are only imported for ``draw_graph``):

* fully connected when ``undirected_neighbor_num >= n_nodes - 1`` (reference ``:30-35,303-318``)
* ring = ring lattice with one neighbour per side (the reference's Watts-Strogatz(n, 2, p=0),
  ``:213-228``), optionally with 10 % random extra symmetric links
* "random symmetric" = ring U ring-lattice(k) (the reference's WS(n, k, p=0) is deterministic,
  ``:230-254``) plus, as an extension, ``extra_random_links`` seeded Erdos-Renyi links so that a
  genuinely random graph can be requested
* random asymmetric (``:256-301``), star / server (``:121-125``), custom matrix (``:130-131``)
"""
from __future__ import annotations

import json
import random
from typing import List, Optional

import numpy as np

from ..role import Role


def ring_lattice(n: int, k: int) -> np.ndarray:
    """Adjacency of a ring lattice where each node links to its k//2 nearest neighbours per side."""
    a = np.zeros((n, n), dtype=np.float32)
    half = max(int(k) // 2, 0)
    for i in range(n):
        for d in range(1, half + 1):
            j = (i + d) % n
            if j != i:
                a[i, j] = a[j, i] = 1
    return a


class TopologyManager:
    def __init__(self, scenario_name=None, n_nodes=5, b_symmetric=True, undirected_neighbor_num=5, topology=None,
                 seed: Optional[int] = None):
        self.scenario_name = scenario_name
        self.n_nodes = int(n_nodes)
        self.b_symmetric = b_symmetric
        self.undirected_neighbor_num = undirected_neighbor_num
        self.topology = [] if topology is None else topology
        self.nodes = np.zeros((self.n_nodes, 3), dtype=np.int32)
        self.seed = seed
        self._rng = random.Random(seed)
        self._nprng = np.random.RandomState(seed)
        if self.undirected_neighbor_num < 2:
            raise ValueError("undirected_neighbor_num must be greater than 2")
        self.b_fully_connected = bool(self.undirected_neighbor_num >= self.n_nodes - 1 and self.b_symmetric)

    def __getstate__(self):
        return {"scenario_name": self.scenario_name, "n_nodes": self.n_nodes, "topology": self.topology,
                "nodes": self.nodes}

    def __setstate__(self, state):
        self.scenario_name = state["scenario_name"]
        self.n_nodes = state["n_nodes"]
        self.topology = state["topology"]
        self.nodes = state["nodes"]

    # ------------------------------------------------------------------ generators
    def generate_topology(self, extra_random_links: float = 0.0):
        if self.b_fully_connected:
            self._fully_connected()
        elif self.b_symmetric:
            self._random_symmetric(extra_random_links)
        else:
            self._random_asymmetric()

    def generate_server_topology(self):
        t = np.zeros((self.n_nodes, self.n_nodes), dtype=np.float32)
        t[0, :] = 1
        t[:, 0] = 1
        np.fill_diagonal(t, 0)
        self.topology = t

    def generate_ring_topology(self, increase_convergence=False):
        t = ring_lattice(self.n_nodes, 2)
        if increase_convergence:
            for i in range(self.n_nodes):
                for j in range(self.n_nodes):
                    if t[i, j] == 0 and self._rng.random() < 0.1:
                        t[i, j] = t[j, i] = 1
        np.fill_diagonal(t, 0)
        self.topology = t

    def generate_custom_topology(self, topology):
        self.topology = np.asarray(topology, dtype=np.float32)

    def _fully_connected(self):
        t = np.ones((self.n_nodes, self.n_nodes), dtype=np.float32)
        np.fill_diagonal(t, 0)
        self.topology = t

    def _random_symmetric(self, extra_random_links: float = 0.0):
        t = np.maximum(ring_lattice(self.n_nodes, 2), ring_lattice(self.n_nodes, int(self.undirected_neighbor_num)))
        if extra_random_links > 0 and self.n_nodes > 3:
            # expected `extra_random_links` additional undirected links per node
            p = min(1.0, extra_random_links / max(self.n_nodes - 3, 1))
            upper = np.triu(self._nprng.random_sample((self.n_nodes, self.n_nodes)) < p, k=1)
            t = np.maximum(t, (upper | upper.T).astype(np.float32))
        np.fill_diagonal(t, 0)
        self.topology = t

    def _random_asymmetric(self):
        t = np.maximum(ring_lattice(self.n_nodes, 2), ring_lattice(self.n_nodes, int(self.undirected_neighbor_num)))
        np.fill_diagonal(t, 0)
        out_links = set()
        n = self.n_nodes
        for i in range(n):
            zeros = [j for j in range(n) if t[i, j] == 0]
            pick = self._nprng.randint(2, size=len(zeros))
            for sel, j in zip(pick, zeros):
                if sel == 1 and (j * n + i) not in out_links:
                    t[i, j] = 1
                    out_links.add(i * n + j)
        np.fill_diagonal(t, 0)
        self.topology = t

    # ------------------------------------------------------------------ accessors
    def get_matrix_adjacency_from_neighbors(self, neighbors):
        m = np.zeros((self.n_nodes, self.n_nodes), dtype=np.float32)
        for i in range(self.n_nodes):
            for j in range(self.n_nodes):
                if i in neighbors[j]:
                    m[i, j] = 1
        return m

    def get_topology(self):
        return self.topology

    def get_nodes(self):
        return self.nodes

    @staticmethod
    def get_coordinates(random_geo=True):
        if not random_geo:
            return None
        if random.randint(0, 1) == 0:   # Spain
            return random.uniform(36.0, 43.0), random.uniform(-9.0, 4.0)
        return random.uniform(45.8, 47.8), random.uniform(5.9, 10.5)   # Switzerland

    def add_nodes(self, nodes):
        self.nodes = nodes

    def update_nodes(self, config_participants):
        self.nodes = [(n["network_args"]["ip"], n["network_args"]["port"], n["device_args"]["role"],
                       n["network_args"].get("ipdemo")) for n in config_participants]

    def get_node(self, node_idx):
        return self.nodes[node_idx]

    def get_neighbors(self, node_idx):
        idx = [i for i in range(self.n_nodes) if self.topology[node_idx][i] == 1]
        return idx, [self.nodes[i] for i in idx]

    def get_neighbors_index(self, node_idx) -> List[int]:
        return [i for i in range(self.n_nodes) if self.topology[node_idx][i] == 1]

    def get_neighbors_string(self, node_idx):
        _, data = self.get_neighbors(node_idx)
        return " ".join(f"{d[0]}:{d[1]}" for d in data)

    # ------------------------------------------------------------------ rendering / export
    def draw_graph(self, plot=False, path=None):
        """Role-coloured PNG (reference ``:48-109``); needs networkx + matplotlib."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        import networkx as nx
        g = nx.from_numpy_array(np.asarray(self.topology))
        pos = nx.spring_layout(g, k=0.15, iterations=20, seed=42)
        fig = plt.figure(num="Network topology", dpi=100, figsize=(6, 6), frameon=False)
        ax = fig.add_axes([0, 0, 1, 1])
        ax.set_xlim([-1.3, 1.3])
        ax.set_ylim([-1.3, 1.3])
        colors = {Role.AGGREGATOR: "orange", Role.SERVER: "green", Role.TRAINER: "#6182bd", Role.PROXY: "purple"}
        color_map, labels = [], {}
        for k in range(self.n_nodes):
            node = self.nodes[k]
            role = str(node[2])
            color_map.append(colors.get(role, "red"))
            host = node[3] if len(node) > 3 and node[3] not in (None, "127.0.0.1") else node[0]
            labels[k] = f"P{k}\n{host}:{node[1]}"
        nx.draw_networkx_nodes(g, pos, node_color=color_map, linewidths=2)
        nx.draw_networkx_labels(g, pos, labels, font_size=10, font_weight="bold")
        nx.draw_networkx_edges(g, pos, width=2)
        roles = {str(n[2]) for n in self.nodes}
        for role, name in ((Role.AGGREGATOR, "Aggregator"), (Role.SERVER, "Server"), (Role.TRAINER, "Trainer"),
                           (Role.PROXY, "Proxy"), (Role.IDLE, "Idle")):
            if role in roles:
                plt.scatter([], [], c=colors.get(role, "red"), label=name)
        plt.legend()
        if path is not None:
            plt.savefig(f"{path}", dpi=100, bbox_inches="tight", pad_inches=0)
        if plot:
            plt.show()
        plt.close(fig)

    @staticmethod
    def update_topology_3d_json(participants, path=None):
        """nodes / links document of the 3-D graph; written to ``path`` when one is given, always returned."""
        data = {"nodes": [], "links": []}
        for node in participants:
            ipport = node["network_args"]["ip"] + ":" + str(node["network_args"]["port"])
            data["nodes"].append({"uid": node["device_args"]["uid"], "id": node["device_args"]["idx"],
                                  "ipport": ipport, "role": node["device_args"]["role"],
                                  "ip": node["network_args"]["ip"], "port": str(node["network_args"]["port"]),
                                  "start": node["device_args"]["start"]})
        for node in participants:
            src = node["network_args"]["ip"] + ":" + str(node["network_args"]["port"])
            for nb in node["network_args"]["neighbors"].split(" "):
                if nb != "":
                    data["links"].append({"source": src, "target": nb, "value": 1})
        if path is not None:
            with open(path, "w") as f:
                json.dump(data, f, sort_keys=False, indent=2)
        return data
