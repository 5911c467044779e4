"""Controller: prepares a scenario and launches its nodes (reference: fedstellar/controller.py:67-503).

Same responsibilities and method names: ``start()`` (banner, environment, web server + statistics
server, idle loop), ``load_configurations_and_start_nodes()`` (scenario naming, topology creation,
rewriting of every ``participant_<i>.json`` with neighbours / idx / uid / geo / dirs, exactly one
start node, topology.png, node launch), ``create_topology``, ``start_nodes_cmd`` (one OS process per
node), ``start_nodes_docker`` (generated compose file), and the static ``killports / killport /
killdockers`` helpers.

MI355X addition: ``start_nodes_batched`` — ONE process per GPU hosting all nodes of the scenario in
a :class:`~p2pfl_amd.runtime.federation.BatchedFederation` (``--engine batched``, the default
whenever the scenario's dataset/model pair is supported by the batched engine). ``generate_participants``
builds the participant files from a handful of arguments so that a 1000-node scenario does not need
the web UI.
"""
from __future__ import annotations

import glob
import hashlib
import json
import logging
import os
import re
import shutil
import signal
import subprocess
import sys
import time
from datetime import datetime

from .config import Config, default_participant
from .role import default_roles
from .utils.topologymanager import TopologyManager

BATCHED_PAIRS = {("MNIST", "MLP"), ("MNIST", "CNN"), ("FEMNIST", "CNN")}
_ANSI = re.compile(r"\x1B\[[0-?]*[ -/]*[@-~]")


class TermEscapeCodeFormatter(logging.Formatter):
    """Strip ANSI colour codes for the controller's own log file (reference controller.py:23-43)."""

    def format(self, record):
        return _ANSI.sub("", super().format(record))


def signal_handler(sig, frame):
    logging.info("You pressed Ctrl+C! Cleaning up...")
    Controller.killports("tensorboard")
    Controller.killports("node_start")
    Controller.killdockers()
    sys.exit(0)


class Controller:
    def __init__(self, args):
        g = lambda k, d=None: getattr(args, k, d)
        self.scenario_name = g("scenario_name")
        self.start_date_scenario = None
        self.cloud = g("cloud")
        self.federation = g("federation", "DFL")
        self.topology = g("topology", "fully")
        self.webserver = g("webserver", False)
        self.webserver_port = g("webport", 5000)
        self.statistics_port = g("statsport", 5100)
        self.simulation = g("simulation", True)
        self.docker = g("docker")
        self.config_dir = g("config")
        self.log_dir = g("logs")
        self.env_path = g("env")
        self.python_path = g("python") or sys.executable
        self.matrix = g("matrix")
        self.engine = g("engine", "auto")          # auto | batched | process
        self.gpus = g("gpus")                      # number of GPUs for the batched engine (None: all visible)
        # batched engine, one rank: serve the node protocol for the co-resident nodes (runtime/host.py) and open
        # TCP gateways ("IDX[:PORT]") through which reference-protocol nodes peer with virtual node IDX
        self.control_plane = bool(g("control_plane", False))
        self.gateways = list(g("gateway", None) or [])
        self.wait_start = g("wait_start", None)
        self.encrypt = bool(g("encrypt", False))
        self.network_subnet = g("network_subnet")
        self.network_gateway = g("network_gateway")
        self.config = Config(entity="controller")
        self.topologymanager = None
        self.n_nodes = 0
        self.processes = []
        self.mender = None
        if not self.simulation:
            from .config.mender import Mender
            self.mender = Mender()

    # ------------------------------------------------------------------ long-running services
    def start(self):
        banner = r"""
              ____ ____        __ _                      _
        _ __ |___ \  _ \ / _| |   __ _ _ __ ___   __| |
       | '_ \  __) | |_) | |_| |  / _` | '_ ` _ \ / _` |
       | |_) |/ __/|  __/|  _| | | (_| | | | | | | (_| |
       | .__/|_____|_|   |_| |_|  \__,_|_| |_| |_|\__,_|
       |_|    MI355X-native decentralized federated learning
        """
        print("\x1b[0;36m" + banner + "\x1b[0m")
        self._load_dotenv(self.env_path)
        os.environ["FEDSTELLAR_ROOT"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.environ["FEDSTELLAR_LOGS_DIR"] = str(self.log_dir)
        os.environ["FEDSTELLAR_CONFIG_DIR"] = str(self.config_dir)
        os.environ["FEDSTELLAR_PYTHON_PATH"] = str(self.python_path)
        os.environ["FEDSTELLAR_STATISTICS_PORT"] = str(self.statistics_port)
        signal.signal(signal.SIGINT, signal_handler)
        if self.webserver:
            self.run_webserver()
            self.run_statistics()
        elif self.mender is not None:
            logging.info("[Mender.module] devices: %s", self.mender.get_devices())
        else:
            self.load_configurations_and_start_nodes()
        logging.info("Press Ctrl+C for exit")
        while True:
            time.sleep(1)
            if not self.webserver and self.processes and all(p.poll() is not None for p in self.processes):
                logging.info("all node processes finished")
                return

    @staticmethod
    def _load_dotenv(path):
        if not path or not os.path.exists(path):
            return
        with open(path) as f:
            for line in f:
                line = line.strip()
                if line and not line.startswith("#") and "=" in line:
                    k, v = line.split("=", 1)
                    os.environ.setdefault(k.strip(), v.strip().strip('"').strip("'"))

    def run_webserver(self):
        app = os.path.join(os.path.dirname(os.path.abspath(__file__)), "webserver", "app.py")
        cmd = [self.python_path, app, "--port", str(self.webserver_port)]
        logging.info("Running web server: %s", " ".join(cmd))
        self.processes.append(subprocess.Popen(cmd, env=os.environ.copy()))

    def run_statistics(self):
        """TensorBoard over the scenario log dir when it is installed (reference controller.py:184-202);
        otherwise the web server's own ``/scenario/<name>/statistics`` CSV views are the only UI."""
        tb = shutil.which("tensorboard")
        if tb is None:
            logging.info("tensorboard is not installed: statistics are served by the web server from metrics.csv")
            return
        cmd = [tb, "--host", "0.0.0.0", "--port", str(self.statistics_port), "--logdir", str(self.log_dir),
               "--reload_interval", "1"]
        self.processes.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))

    # ------------------------------------------------------------------ process hygiene
    @staticmethod
    def killports(term="python"):
        """Terminate OUR child processes whose command line contains ``term`` (never a global pattern kill)."""
        try:
            import psutil
            me = psutil.Process()
            for child in me.children(recursive=True):
                try:
                    if term in " ".join(child.cmdline()):
                        child.terminate()
                except (psutil.NoSuchProcess, psutil.AccessDenied):
                    pass
        except ImportError:
            pass

    @staticmethod
    def killport(port):
        try:
            import psutil
            for c in psutil.net_connections(kind="tcp"):
                if c.laddr and c.laddr.port == int(port) and c.status == psutil.CONN_LISTEN and c.pid:
                    if c.pid != os.getpid():
                        psutil.Process(c.pid).terminate()
        except Exception as e:
            logging.debug("killport(%s): %s", port, e)

    @staticmethod
    def killdockers():
        if shutil.which("docker") is None:
            return
        try:
            ids = subprocess.run(["docker", "ps", "-q", "--filter", "name=fedstellar"], capture_output=True,
                                 text=True, timeout=10).stdout.split()
            if ids:
                subprocess.run(["docker", "kill"] + ids, timeout=30)
                subprocess.run(["docker", "rm"] + ids, timeout=30)
        except Exception as e:
            logging.debug("killdockers: %s", e)

    @staticmethod
    def remove_files_by_scenario(scenario_name):
        """Delete ``<config_dir>/<scenario>`` and ``<log_dir>/<scenario>`` - and nothing else: the name must be one
        plain path component (never "." / ".." / a path) and the target must resolve to a strict descendant of
        its root."""
        name = str(scenario_name or "")
        if not name or set(name) <= {"."} or name != os.path.basename(name) or os.sep in name or "/" in name:
            raise ValueError(f"refusing to remove scenario {scenario_name!r}")
        for root in (os.environ.get("FEDSTELLAR_CONFIG_DIR"), os.environ.get("FEDSTELLAR_LOGS_DIR")):
            if not root:
                continue
            real_root, target = os.path.realpath(root), os.path.realpath(os.path.join(root, name))
            if target == real_root or os.path.commonpath([real_root, target]) != real_root:
                raise ValueError(f"refusing to remove {target!r}: outside {real_root!r}")
            shutil.rmtree(target, ignore_errors=True)

    # ------------------------------------------------------------------ scenario generation
    @staticmethod
    def generate_participants(config_dir, n_nodes, federation="DFL", dataset="MNIST", model="MLP", rounds=3, epochs=3,
                              base_port=45000, ip="127.0.0.1", accelerator="auto", controller="127.0.0.1:5000",
                              overrides=None, start_idx=0):
        """Write ``participant_<i>.json`` for an ``n_nodes`` scenario (what the web UI's deployment
        designer produces, webserver/app.py:640-673); roles follow the UI defaults per federation."""
        os.makedirs(config_dir, exist_ok=True)
        roles = default_roles(federation, n_nodes)
        files = []
        for i in range(n_nodes):
            p = default_participant()
            p["scenario_args"].update(federation=federation, rounds=rounds, n_nodes=n_nodes, controller=controller)
            p["device_args"].update(idx=i, role=roles[i], start=(i == start_idx), accelerator=accelerator)
            p["network_args"].update(ip=ip, port=base_port + i)
            p["data_args"]["dataset"] = dataset
            p["model_args"]["model"] = model
            p["training_args"]["epochs"] = epochs
            for k, v in (overrides or {}).items():
                if isinstance(v, dict) and isinstance(p.get(k), dict):
                    p[k].update(v)
                else:
                    p[k] = v
            path = os.path.join(config_dir, f"participant_{i}.json")
            with open(path, "w") as f:
                json.dump(p, f, indent=2)
            files.append(path)
        return files

    def load_configurations_and_start_nodes(self):
        if not self.scenario_name:
            self.scenario_name = f'fedstellar_{self.federation}_{datetime.now().strftime("%d_%m_%Y_%H_%M_%S")}'
        if os.path.basename(os.path.normpath(self.config_dir)) != self.scenario_name:
            scen_dir = os.path.join(self.config_dir, self.scenario_name)
            if not os.path.isdir(scen_dir) and glob.glob(os.path.join(self.config_dir, "participant_*.json")):
                os.makedirs(scen_dir, exist_ok=True)   # flat config dir: move the files under the scenario name
                for f in glob.glob(os.path.join(self.config_dir, "participant_*.json")):
                    shutil.copy(f, scen_dir)
            self.config_dir = scen_dir
        os.makedirs(self.config_dir, exist_ok=True)
        os.makedirs(os.path.join(self.log_dir, self.scenario_name), exist_ok=True)
        self.start_date_scenario = datetime.now().strftime("%d/%m/%Y %H:%M:%S")
        logging.info("Generating the scenario %s at %s", self.scenario_name, self.start_date_scenario)

        participant_files = sorted(glob.glob(os.path.join(self.config_dir, "participant_*.json")),
                                   key=lambda p: int(os.path.basename(p).split("_")[1].split(".")[0]))
        if not participant_files:
            # a scenario described by ONE file (config/scenario.py) is expanded into the per-participant documents
            from .config.scenario import SCENARIO_FILE, emit_participants
            for cand in (self.config_dir, os.path.dirname(os.path.normpath(self.config_dir))):
                if os.path.exists(os.path.join(cand, SCENARIO_FILE)):
                    participant_files = emit_participants(os.path.join(cand, SCENARIO_FILE), self.config_dir)
                    break
        if not participant_files:
            raise ValueError("No participant files found in config folder")
        self.config.set_participants_config(participant_files)
        self.n_nodes = len(participant_files)
        self.topologymanager = self.create_topology(matrix=self.matrix) if self.matrix is not None else self.create_topology()

        # token the nodes present to the controller's ingest routes (status / log push carry no login session)
        import secrets
        token = secrets.token_hex(16)
        with open(os.path.join(self.config_dir, ".token"), "w") as f:
            f.write(token)
        is_start_node, idx_start_node = False, 0
        for i in range(self.n_nodes):
            path = os.path.join(self.config_dir, f"participant_{i}.json")
            with open(path) as f:
                pc = json.load(f)
            pc["scenario_args"]["controller_token"] = token
            pc["scenario_args"]["federation"] = self.federation
            pc["scenario_args"]["n_nodes"] = self.n_nodes
            pc["scenario_args"]["simulation"] = bool(self.simulation)
            pc["network_args"]["neighbors"] = self.topologymanager.get_neighbors_string(i)
            pc["scenario_args"]["name"] = self.scenario_name
            pc["scenario_args"]["start_time"] = self.start_date_scenario
            pc["scenario_args"]["encrypt"] = self.encrypt
            pc["device_args"]["idx"] = i
            pc["device_args"]["uid"] = hashlib.sha1(
                (str(pc["network_args"]["ip"]) + str(pc["network_args"]["port"]) + str(self.scenario_name)).encode()).hexdigest()
            if not pc["geo_args"].get("fixed"):       # random in Spain / Switzerland (controller.py:263) unless the designer placed it
                pc["geo_args"]["latitude"], pc["geo_args"]["longitude"] = TopologyManager.get_coordinates(random_geo=True)
            pc["tracking_args"]["log_dir"] = self.log_dir
            pc["tracking_args"]["config_dir"] = self.config_dir
            if pc["device_args"]["start"]:
                if is_start_node:
                    raise ValueError("Only one node can be start node")
                is_start_node, idx_start_node = True, i
            with open(path, "w") as f:
                json.dump(pc, f, sort_keys=False, indent=2)
        if not is_start_node:
            raise ValueError("No start node found")
        # the same scenario as ONE file (defaults + per-node deltas) beside the per-participant documents
        from .config.scenario import read_participant_files, write_scenario
        from .config.schema import validate_scenario
        validate_scenario(read_participant_files(self.config_dir))      # typed checks across the whole scenario
        write_scenario(self.config_dir)
        self.config.set_participants_config(participant_files)
        self.topologymanager.update_nodes(self.config.participants)
        try:
            self.topologymanager.draw_graph(path=os.path.join(self.log_dir, self.scenario_name, "topology.png"), plot=False)
        except Exception as e:   # rendering is cosmetic (needs networkx + matplotlib)
            logging.info("topology rendering skipped: %s", e)

        if not self.simulation:
            logging.info("Simulation mode is disabled, waiting for nodes to start...")
            return
        if self.docker:
            self.start_nodes_docker(idx_start_node)
        elif self._use_batched():
            self.start_nodes_batched()
        else:
            self.start_nodes_cmd(idx_start_node)

    def _use_batched(self) -> bool:
        p0 = self.config.participants[0]
        pair = (p0["data_args"]["dataset"], p0["model_args"]["model"])
        homogeneous = all((p["data_args"]["dataset"], p["model_args"]["model"]) == pair for p in self.config.participants)
        if self.engine == "process":
            if homogeneous and any(
                    p["device_args"].get("accelerator") in ("gpu", "auto") for p in self.config.participants):
                # compatibility path: every OS process drives its own one-client engine (K = 1 launches, a host
                # round trip per payload, the TCP pacing constants) - two orders of magnitude below the batched path
                logging.warning("--engine process with a GPU accelerator runs ONE client per kernel launch and per "
                                "process; %s / %s runs client-batched in the one-process-per-GPU runtime (hand-written "
                                "kernels for the MNIST / FEMNIST models, torch.func.vmap steps for the others): pass "
                                "--engine batched to host all %d nodes in one process per GPU",
                                pair[0], pair[1], self.n_nodes)
            return False
        if self.engine == "batched":
            # every family of the factory runs batched (kernel-level specs on the HIP engine, the others as
            # nn.Modules on stock ops with their state in the slab) - but all nodes must run the same one
            if not homogeneous:
                raise ValueError("--engine batched needs the same dataset / model on every node")
            return True
        # auto: every homogeneous scenario runs client-batched in one process per GPU (BATCHED_PAIRS on the hand-written
        # kernels, the other families as torch.func.vmap steps / hipGraph replays: 20-120 x the per-node loop for the
        # small models); mixed scenarios need one process per node
        return homogeneous

    def create_topology(self, matrix=None):
        import numpy as np
        n = self.n_nodes
        if matrix is not None:
            tm = TopologyManager(topology=np.array(matrix), scenario_name=self.scenario_name, n_nodes=n, b_symmetric=True,
                                 undirected_neighbor_num=max(n - 1, 2))
        elif self.topology == "fully":
            tm = TopologyManager(scenario_name=self.scenario_name, n_nodes=n, b_symmetric=True,
                                 undirected_neighbor_num=max(n - 1, 2))
            tm.generate_topology()
        elif self.topology == "ring":
            tm = TopologyManager(scenario_name=self.scenario_name, n_nodes=n, b_symmetric=True)
            tm.generate_ring_topology(increase_convergence=True)
        elif self.topology == "random":
            tm = TopologyManager(scenario_name=self.scenario_name, n_nodes=n, b_symmetric=True, undirected_neighbor_num=3)
            tm.generate_topology()
        elif self.topology == "star" and self.federation == "CFL":
            tm = TopologyManager(scenario_name=self.scenario_name, n_nodes=n, b_symmetric=True)
            tm.generate_server_topology()
        else:
            raise ValueError("Unknown topology type: {}".format(self.topology))
        tm.add_nodes([(p["network_args"]["ip"], p["network_args"]["port"], "undefined", p["network_args"].get("ipdemo"))
                      for p in self.config.participants])
        return tm

    # ------------------------------------------------------------------ launchers
    def start_node(self, idx):
        cmd = [self.python_path, "-u", "-m", "p2pfl_amd.node_start", os.path.join(self.config_dir, f"participant_{idx}.json")]
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        with open(os.path.join(self.log_dir, self.scenario_name, f"participant_{idx}.stdout"), "w") as log:
            self.processes.append(subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env))   # the child keeps its own descriptor

    def start_nodes_cmd(self, idx_start_node):
        """One OS process per node; the start node last (reference controller.py:473-485)."""
        for idx in range(self.n_nodes):
            if idx != idx_start_node:
                self.start_node(idx)
        time.sleep(float(os.environ.get("P2PFL_LAUNCH_GAP", "1")))
        self.start_node(idx_start_node)

    def start_nodes_batched(self):
        """One process per GPU hosting every node of the scenario (the MI355X execution model)."""
        try:
            import torch
            n_gpu = torch.cuda.device_count()
        except Exception:
            n_gpu = 0
        world = int(self.gpus) if self.gpus else max(1, min(n_gpu, self.n_nodes))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root)
        args = [self.config_dir] + (["--encrypt"] if self.encrypt else [])
        if self.control_plane or self.gateways:
            # (one FederationHost per rank, joined by the host-side rank bridge: runtime/host.py, runtime/bridge.py)
            args += ["--control-plane"] + [x for g in self.gateways for x in ("--gateway", str(g))]
            if self.wait_start is not None:
                args += ["--wait-start", str(self.wait_start)]
        if world > 1:
            cmd = [self.python_path, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                   "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 1000),
                   "-m", "p2pfl_amd.runtime.launch"] + args
        else:
            cmd = [self.python_path, "-u", "-m", "p2pfl_amd.runtime.launch"] + args
        logging.info("Launching batched federation: %s", " ".join(cmd))
        with open(os.path.join(self.log_dir, self.scenario_name, "batched_launcher.log"), "w") as log:
            self.processes.append(subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env))

    def start_nodes_docker(self, idx_start_node):
        """Generate a compose file (one ROCm container per node, static IPs on a bridge network) and
        bring it up (reference controller.py:347-454 does the same with the CUDA image)."""
        subnet = self.network_subnet or "192.168.50.0/24"
        gateway = self.network_gateway or "192.168.50.1"
        base = subnet.rsplit(".", 1)[0]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lines = ["services:"]
        for idx, p in enumerate(self.config.participants):
            ip = f"{base}.{idx + 2}"
            delay = 10 if idx == idx_start_node else 0
            lines += [
                f"  participant{idx}:",
                "    image: p2pfl-amd",
                "    restart: \"no\"",
                "    devices: [\"/dev/kfd\", \"/dev/dri\"]",
                "    group_add: [\"video\"]",
                "    volumes:",
                f"      - {root}:/p2pfl_amd_root",
                f"      - {os.path.abspath(self.config_dir)}:/config",
                f"      - {os.path.abspath(self.log_dir)}:/logs",
                f"    command: /bin/bash -c \"sleep {delay} && cd /p2pfl_amd_root && python3 -m p2pfl_amd.node_start /config/participant_{idx}.json\"",
                "    networks:",
                "      fedstellar-net:",
                f"        ipv4_address: {ip}",
            ]
            # containers talk over the bridge: rewrite the participant's address book accordingly
            path = os.path.join(self.config_dir, f"participant_{idx}.json")
            with open(path) as f:
                pc = json.load(f)
            pc["network_args"]["ip"] = ip
            pc["tracking_args"]["log_dir"], pc["tracking_args"]["config_dir"] = "/logs", "/config"
            with open(path, "w") as f:
                json.dump(pc, f, indent=2)
        lines += ["networks:", "  fedstellar-net:", "    driver: bridge", "    ipam:", "      config:",
                  f"        - subnet: {subnet}", f"          gateway: {gateway}"]
        # neighbours must point at the container addresses
        ips = {i: f"{base}.{i + 2}" for i in range(self.n_nodes)}
        for idx in range(self.n_nodes):
            path = os.path.join(self.config_dir, f"participant_{idx}.json")
            with open(path) as f:
                pc = json.load(f)
            nbrs = self.topologymanager.get_neighbors_index(idx)
            pc["network_args"]["neighbors"] = " ".join(f"{ips[j]}:{self.config.participants[j]['network_args']['port']}" for j in nbrs)
            with open(path, "w") as f:
                json.dump(pc, f, indent=2)
        compose = os.path.join(self.config_dir, "docker-compose.yml")
        with open(compose, "w") as f:
            f.write("\n".join(lines) + "\n")
        if shutil.which("docker") is None:
            logging.warning("docker is not installed: compose file written to %s, nothing started", compose)
            return compose
        self.processes.append(subprocess.Popen(["docker", "compose", "-f", compose, "up", "-d"]))
        return compose

    def wait(self, timeout=None):
        """Block until every launched process exited; returns their exit codes."""
        t0 = time.time()
        for p in self.processes:
            left = None if timeout is None else max(0.0, timeout - (time.time() - t0))
            try:
                p.wait(left)
            except subprocess.TimeoutExpired:
                p.terminate()
        return [p.returncode for p in self.processes]
