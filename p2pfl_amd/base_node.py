"""BaseNode: the networking half of a federation node (reference: fedstellar/base_node.py:34-539).

Owns the listening TCP socket, the neighbour table (one :class:`NodeConnection` per peer), the
heartbeater and the gossiper, and routes their events. Handshake and messages are the reference's
(``CONNECT ip port full force`` on a fresh socket that then becomes the peer link, optional
``CONNECT_TO`` fan-out for "full" connects). Fixes w.r.t. the reference:

* encryption is decided by ONE flag on both ends (the reference's acceptor tests ``self.encrypt``
  while its connector tests ``not self.simulation``, base_node.py:248 vs :348, which deadlocks the
  handshake), and the AES session key is wrapped with the peer's RSA-OAEP public key instead of
  being sent in clear (base_node.py:256,354);
* heartbeater and gossiper read the neighbour table through a locked accessor instead of sharing
  the raw list object (base_node.py:173-176).
"""
from __future__ import annotations

import logging
import os
import shutil
import socket
import subprocess
import sys
import threading

from .communication_protocol import CommunicationProtocol
from .config import Config
from .encrypter import AESCipher, RSACipher
from .gossiper import Gossiper
from .heartbeater import Heartbeater
from .node_connection import NodeConnection
from .utils.observer import Events, Observer

_RSA_KEY_B64_LEN = 216     # base64(DER SPKI) of an RSA-1024 key
_RSA_BLOCK_LEN = 128


def _recv_line(s: socket.socket, limit: int) -> bytes:
    """The handshake line, up to and including its newline, and NOT A BYTE MORE: a dialling node may send
    START_LEARNING / its first PARAMS fragments right behind ``CONNECT ...\n``; a plain ``recv(BLOCK_SIZE)`` here
    (what the reference does, base_node.py:229) swallows them with the handshake and the peer never starts."""
    out = b""
    while len(out) < limit:
        peek = s.recv(limit - len(out), socket.MSG_PEEK)
        if not peek:
            break
        pos = peek.find(b"\n")
        take = len(peek) if pos < 0 else pos + 1
        out += _recv_exact(s, take)
        if pos >= 0:
            break
    return out


def _recv_exact(s: socket.socket, n: int) -> bytes:
    out = b""
    while len(out) < n:
        chunk = s.recv(n - len(out))
        if not chunk:
            raise ConnectionError("peer closed during handshake")
        out += chunk
    return out


class _DebugOnly(logging.Filter):
    def filter(self, record):
        return record.levelno == logging.DEBUG


class BaseNode(threading.Thread, Observer):
    """
    Args:
        experiment_name: scenario name (log directory).
        hostdemo: address shown in demos (optional).
        host / port: listen address; port ``None``/0 picks a free one.
        encrypt: encrypt every peer link (RSA key exchange + AES-128 per link).
        config: :class:`Config`; defaults are used when missing.
    """

    def __init__(self, experiment_name, hostdemo=None, host="127.0.0.1", port=None, encrypt=False, config=None):
        self.experiment_name = experiment_name
        threading.Thread.__init__(self, daemon=True)
        Observer.__init__(self)
        self._terminate_flag = threading.Event()
        self.hostdemo = hostdemo
        self.host = socket.gethostbyname(host)
        self.port = port
        self.encrypt = bool(encrypt)
        if config is None or isinstance(config, type):
            config = Config(participant={})
        self.config = config
        self.simulation = bool(self.config.participant["scenario_args"].get("simulation", True))

        self.__node_socket = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.__node_socket.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.__node_socket.bind((self.host, int(self.port or 0)))
        self.__node_socket.listen(50)
        if not self.port:
            self.port = self.__node_socket.getsockname()[1]
        self.name = "node-" + self.get_name()

        if not self.simulation:
            self._shape_network()

        self.__neighbors = []
        self.__nei_lock = threading.RLock()
        self.heartbeater = None
        self.gossiper = None

        self.log_dir = os.path.join(self.config.participant["tracking_args"].get("log_dir") or "logs",
                                    self.experiment_name or "scenario")
        self.log_filename = None
        self._log_handlers = []
        if self.config.participant["tracking_args"].get("log_dir"):
            self.setup_logging()

    # ------------------------------------------------------------------ logging / shaping
    def setup_logging(self, log_dir=None):
        """console INFO + ``participant_<idx>.log`` + ``_debug.log`` + ``_error.log``
        (reference base_node.py:92-106,133-158)."""
        log_dir = log_dir or self.log_dir
        os.makedirs(log_dir, exist_ok=True)
        idx = self.config.participant["device_args"].get("idx", "")
        self.log_filename = os.path.join(log_dir, f"participant_{idx}")
        fmt = logging.Formatter(f"[%(asctime)s] [{self.get_name()}] [%(levelname)s] %(message)s")
        console = logging.StreamHandler(sys.stdout)
        console.setLevel(logging.INFO)
        console.setFormatter(logging.Formatter(f"\x1b[0;36m[{self.get_name()}]\x1b[0m %(message)s"))
        info = logging.FileHandler(self.log_filename + ".log", mode="w", encoding="utf-8")
        info.setLevel(logging.INFO)
        debug = logging.FileHandler(self.log_filename + "_debug.log", mode="w", encoding="utf-8")
        debug.setLevel(logging.DEBUG)
        debug.addFilter(_DebugOnly())
        error = logging.FileHandler(self.log_filename + "_error.log", mode="w", encoding="utf-8")
        error.setLevel(logging.WARNING)
        root = logging.getLogger()
        root.setLevel(logging.DEBUG if self.config.participant["scenario_args"].get("debug") else logging.INFO)
        for h in (info, debug, error):
            h.setFormatter(fmt)
        for h in (console, info, debug, error):
            root.addHandler(h)
            self._log_handlers.append(h)

    def _shape_network(self):
        """WAN emulation with ``tcset`` on real deployments (reference base_node.py:82-85)."""
        na = self.config.participant["network_args"]
        if shutil.which("tcset") is None:
            logging.info("[BASENODE] tcset not available, skipping network shaping")
            return
        cmd = ["tcset", "--device", str(na["interface"]), "--rate", str(na["rate"]), "--delay", str(na["delay"]),
               "--delay-distro", str(na["delay-distro"]), "--loss", str(na["loss"])]
        try:
            subprocess.run(cmd, check=False, timeout=10)
        except Exception as e:
            logging.info("[BASENODE] network shaping failed: %s", e)

    # ------------------------------------------------------------------ identity
    def get_addr(self):
        return self.host, self.port

    def get_name(self):
        return f"{self.host}:{self.port}"

    def get_name_demo(self):
        return f"{self.hostdemo}:{self.port}" if self.hostdemo else self.get_name()

    # ------------------------------------------------------------------ lifecycle
    def start(self):
        """Start the accept loop, the heartbeater and the gossiper."""
        super().start()
        self.heartbeater = Heartbeater(self.get_name(), self.get_neighbors, self.config)
        self.gossiper = Gossiper(self.get_name(), self.get_neighbors, self.config)
        self.heartbeater.add_observer(self)
        self.gossiper.add_observer(self)
        self.heartbeater.start()
        self.gossiper.start()

    def stop(self):
        """Stop the node, its helper threads and every peer link."""
        self._terminate_flag.set()
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        try:   # wake the accept loop
            s.settimeout(1)
            s.connect((self.host, self.port))
        except OSError:
            pass
        finally:
            s.close()

    def run(self):
        logging.info("[BASENODE] Node started")
        while not self._terminate_flag.is_set():
            try:
                ns, _ = self.__node_socket.accept()
                if self._terminate_flag.is_set():
                    ns.close()
                    break
                ns.settimeout(5)
                msg = _recv_line(ns, int(self.config.participant["BLOCK_SIZE"]))
                if msg:
                    callback = lambda h, p, fu, fc: self.__process_new_connection(ns, h, p, fu, fc)
                    if not CommunicationProtocol.process_connection(msg, callback):
                        ns.close()
                else:
                    ns.close()
            except Exception as e:
                if not self._terminate_flag.is_set():
                    logging.exception(e)
        if self.heartbeater is not None:
            self.heartbeater.stop()
        if self.gossiper is not None:
            self.gossiper.stop()
        logging.info("[BASENODE] Stopping node. Disconnecting from %d nodes", len(self.__neighbors))
        for n in self.get_neighbors():
            n.stop()
        self.__node_socket.close()
        root = logging.getLogger()
        for h in self._log_handlers:
            root.removeHandler(h)
            try:
                h.close()
            except Exception:
                pass

    # ------------------------------------------------------------------ connections
    def __process_new_connection(self, node_socket, h, p, full, force):
        nc = None
        try:
            with self.__nei_lock:
                if self.get_neighbor(h, p) is not None:
                    node_socket.close()
                    return
                # the address the peer announced must be reachable (reference base_node.py:241-244)
                probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                probe.settimeout(2)
                result = probe.connect_ex((h, p))
                probe.close()
                if result != 0:
                    node_socket.close()
                    return
                aes_cipher = None
                if self.encrypt:
                    rsa = RSACipher()
                    node_socket.sendall(rsa.get_key())
                    rsa.load_pair_public_key(_recv_exact(node_socket, _RSA_KEY_B64_LEN))
                    aes_cipher = AESCipher()
                    node_socket.sendall(rsa.encrypt(aes_cipher.get_key()))
                node_socket.settimeout(None)
                logging.info("%s Connection accepted with %s:%s", self.get_name(), h, p)
                nc = NodeConnection(self.get_name(), node_socket, (h, p), aes_cipher, config=self.config)
                nc.add_observer(self)
                self.__neighbors.append(nc)
                nc.start(force=force)
                if full and not nc.is_stopped():
                    self.broadcast(CommunicationProtocol.build_connect_to_msg(h, p), exc=[nc])
        except Exception as e:
            logging.info("[BASENODE] Connection refused with %s:%s (%s)", h, p, e)
            try:
                node_socket.close()
            except OSError:
                pass
            if nc is not None:
                self.rm_neighbor(nc)

    def connect_to(self, h, p, full=False, force=False):
        """Dial ``h:p``; returns the new :class:`NodeConnection` or None."""
        full_s, force_s = ("1" if full else "0"), ("1" if force else "0")
        s = nc = None
        try:
            h = socket.gethostbyname(h)
            with self.__nei_lock:
                if self.get_neighbor(h, p) is not None or (h, int(p)) == (self.host, int(self.port)):
                    logging.info("%s Already connected to %s:%s", self.get_name(), h, p)
                    return None
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(5)
                s.connect((h, int(p)))
                s.sendall(CommunicationProtocol.build_connect_msg(self.host, self.port, full_s, force_s))
                aes_cipher = None
                if self.encrypt:
                    rsa = RSACipher()
                    rsa.load_pair_public_key(_recv_exact(s, _RSA_KEY_B64_LEN))
                    s.sendall(rsa.get_key())
                    aes_cipher = AESCipher(key=rsa.decrypt(_recv_exact(s, _RSA_BLOCK_LEN)))
                s.settimeout(None)
                nc = NodeConnection(self.get_name(), s, (h, int(p)), aes_cipher, config=self.config)
                nc.add_observer(self)
                logging.info("[BASENODE_connect_to] Connected to %s:%s", h, p)
                self.__neighbors.append(nc)
                nc.start(force=force)
                return None if nc.is_stopped() else nc
        except Exception as e:
            logging.info("%s Can't connect to the node %s:%s (%s)", self.get_name(), h, p, e)
            if s is not None and nc is None:      # the dial or the handshake failed: nobody owns the socket yet
                try:
                    s.close()
                except OSError:
                    pass
            return None

    def disconnect_from(self, h, p):
        nc = self.get_neighbor(socket.gethostbyname(h), p)
        if nc is not None:
            nc.stop()

    def get_neighbor(self, h, p, thread_safe=True):
        with self.__nei_lock:
            for n in self.__neighbors:
                if n.get_addr() == (h, int(p)):
                    return n
        return None

    def get_neighbors(self, thread_safe=True):
        with self.__nei_lock:
            return list(self.__neighbors)

    def get_neighbors_names(self):
        return [n.get_name() for n in self.get_neighbors()]

    def rm_neighbor(self, n):
        with self.__nei_lock:
            if n in self.__neighbors:
                self.__neighbors.remove(n)
        n.stop(local=True)

    def get_network_nodes(self):
        """Every live node of the network known through heartbeats (multi-hop)."""
        return self.heartbeater.get_nodes() if self.heartbeater is not None else [self.get_name()]

    # ------------------------------------------------------------------ messaging
    def broadcast(self, msg, exc=None, thread_safe=True):
        """Send ``msg`` to every neighbour except those in ``exc``."""
        exc = exc or []
        for n in self.get_neighbors():
            if n not in exc:
                n.send(msg)

    # ------------------------------------------------------------------ events
    def update(self, event, obj):
        if event == Events.END_CONNECTION_EVENT:
            self.rm_neighbor(obj)
        elif event == Events.NODE_CONNECTED_EVENT:
            n, _ = obj
            n.send(CommunicationProtocol.build_beat_msg(self.get_name()))
        elif event == Events.CONN_TO_EVENT:
            self.connect_to(obj[0], obj[1], full=False)
        elif event == Events.SEND_BEAT_EVENT:
            self.broadcast(CommunicationProtocol.build_beat_msg(self.get_name()))
        elif event == Events.GOSSIP_BROADCAST_EVENT:
            self.broadcast(obj[0], exc=obj[1])
        elif event == Events.PROCESSED_MESSAGES_EVENT:
            node, msgs = obj
            # remember the ids on the other links so the flood does not bounce back, then re-flood
            for nc in self.get_neighbors():
                if nc is not node:
                    nc.comm_protocol.add_processed_messages(list(msgs.keys()))
            if self.gossiper is not None:
                self.gossiper.add_messages(list(msgs.values()), node)
        elif event == Events.BEAT_RECEIVED_EVENT:
            if self.heartbeater is not None:
                self.heartbeater.add_node(obj)
