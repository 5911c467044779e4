"""Wire protocol of the host control plane (reference: fedstellar/communication_protocol.py:32-777).

Same text protocol as the reference so that nodes of this build interoperate on the TCP path:

Gossiped (carry a trailing message id, re-flooded, de-duplicated):
    ``BEAT <node> <id>`` · ``ROLE <node> <role> <id>`` · ``START_LEARNING <rounds> <epochs> <id>`` ·
    ``STOP_LEARNING <id>`` · ``VOTE_TRAIN_SET <node> (<node> <votes>)* \\VOTE_TRAIN_SET <id>`` ·
    ``METRICS <node> <round> <loss> <metric> <id>``
Point to point:
    ``CONNECT <ip> <port> <full> <force>`` · ``CONNECT_TO <ip> <port>`` · ``STOP`` ·
    ``PARAMS<bytes>\\PARAMS`` · ``MODELS_READY <round>`` · ``MODELS_AGGREGATED <node>* \\MODELS_AGGREGATED`` ·
    ``MODEL_INITIALIZED`` · ``TRANSFER_LEADERSHIP``

Differences to the reference, all wire-compatible:
* the parser is table driven (one ``_Rule`` per header instead of a 250-line if/elif ladder);
* message ids are **unsigned 63-bit** random integers — the reference uses Python ``hash()``,
  which may be negative and then fails its own ``isdigit()`` check for ``STOP_LEARNING``
  (communication_protocol.py:297,547);
* the de-duplication window is an O(1) native ring (``csrc/host/framing.cpp``) instead of a list;
* PARAMS framing is done by one native pass over the payload.
"""
from __future__ import annotations

import ctypes
import logging
import secrets
from typing import Callable, Dict, List

from .utils import hostlib


class _IdRing:
    """Fixed-capacity set of recently seen message ids (native, O(1))."""

    def __init__(self, capacity: int):
        self._lib = hostlib.load()
        self._h = self._lib.p2pfl_idring_new(int(capacity))

    def check_and_add(self, key: str) -> bool:
        """True if ``key`` was already seen; otherwise remembers it."""
        return bool(self._lib.p2pfl_idring_check_and_add(self._h, _key64(key)))

    def __contains__(self, key: str) -> bool:
        return bool(self._lib.p2pfl_idring_contains(self._h, _key64(key)))

    def __len__(self) -> int:
        return int(self._lib.p2pfl_idring_size(self._h))

    def __del__(self):
        try:
            self._lib.p2pfl_idring_free(self._h)
        except Exception:
            pass


def _key64(key: str) -> int:
    if key.isdigit():
        return int(key) & 0xFFFFFFFFFFFFFFFF
    import zlib
    return (zlib.crc32(key.encode()) << 32 | zlib.adler32(key.encode())) & 0xFFFFFFFFFFFFFFFF


class CommunicationProtocol:
    """Parses incoming byte blocks into commands and builds outgoing messages.

    Args:
        command_dict: header -> object with ``execute(*args)`` (see :mod:`p2pfl_amd.command`).
        config: :class:`~p2pfl_amd.config.Config` (``AMOUNT_LAST_MESSAGES_SAVED`` is read from it).
    """

    TRANSFER_LEADERSHIP = "TRANSFER_LEADERSHIP"
    BEAT = "BEAT"
    ROLE = "ROLE"
    STOP = "STOP"
    CONN = "CONNECT"
    CONN_TO = "CONNECT_TO"
    START_LEARNING = "START_LEARNING"
    STOP_LEARNING = "STOP_LEARNING"
    PARAMS = "PARAMS"                      # binary
    PARAMS_CLOSE = "\\PARAMS"              # binary
    MODELS_READY = "MODELS_READY"
    METRICS = "METRICS"
    VOTE_TRAIN_SET = "VOTE_TRAIN_SET"
    VOTE_TRAIN_SET_CLOSE = "\\VOTE_TRAIN_SET"
    MODELS_AGGREGATED = "MODELS_AGGREGATED"
    MODELS_AGGREGATED_CLOSE = "\\MODELS_AGGREGATED"
    MODEL_INITIALIZED = "MODEL_INITIALIZED"

    _PARAMS_B = PARAMS.encode("utf-8")
    _PARAMS_CLOSE_B = PARAMS_CLOSE.encode("utf-8")

    def __init__(self, command_dict, config=None):
        self.command_dict = command_dict
        self.config = config
        cap = 100 if config is None else int(config.participant["AMOUNT_LAST_MESSAGES_SAVED"])
        self._seen = _IdRing(cap)
        self.tmp_exec_msgs: Dict[str, bytes] = {}

    # ------------------------------------------------------------------ de-duplication
    @property
    def last_messages(self):
        return self._seen

    def add_processed_messages(self, messages: List[str]) -> None:
        for m in messages:
            self._seen.check_and_add(str(m))

    # ------------------------------------------------------------------ parsing
    def process_message(self, msg: bytes):
        """Execute every command contained in ``msg``.

        Returns ``(executed, error)``: ``executed`` maps message id -> raw command text of the gossiped
        commands that were new to this connection (the caller re-floods them); ``error`` is True if
        the block could not be fully parsed (the reference then drops the link)."""
        self.tmp_exec_msgs = {}
        if msg[:len(self._PARAMS_B)] == self._PARAMS_B:
            end_pos = msg.find(self._PARAMS_CLOSE_B)
            if end_pos != -1:
                return [], not self._exec(self.PARAMS, None, None, msg[len(self._PARAMS_B):end_pos], True)
            return [], not self._exec(self.PARAMS, None, None, msg[len(self._PARAMS_B):], False)
        try:
            tokens = msg.decode("utf-8").split()
        except UnicodeDecodeError:
            return self.tmp_exec_msgs, True
        error = False
        while tokens:
            rule = _RULES.get(tokens[0])
            if rule is None:
                error = True
                break
            consumed = rule(self, tokens)
            if consumed <= 0:
                error = True
                break
            tokens = tokens[consumed:]
        return self.tmp_exec_msgs, error

    def _exec(self, action, hash_, cmd_text, *args) -> bool:
        try:
            if hash_ is None:
                self.command_dict[action].execute(*args)
                return True
            if not self._seen.check_and_add(hash_):
                self.command_dict[action].execute(*args)
                self.tmp_exec_msgs[hash_] = cmd_text
            return True
        except Exception as e:   # a failing callback must not kill the receive loop silently
            logging.info("Error executing callback: %s", e)
            logging.exception(e)
            return False

    # ---- per-header rules: return number of tokens consumed, 0 on a malformed command
    def _r_beat(self, t):
        if len(t) < 3:
            return 0
        return 3 if self._exec(self.BEAT, t[2], _text(t[:3]), t[1]) else 0

    def _r_role(self, t):
        if len(t) < 4:
            return 0
        return 4 if self._exec(self.ROLE, t[3], _text(t[:4]), t[1], t[2]) else 0

    def _r_stop(self, t):
        return 1 if self._exec(self.STOP, None, None) else 0

    def _r_conn_to(self, t):
        if len(t) < 3 or not t[2].isdigit():
            return 0
        return 3 if self._exec(self.CONN_TO, None, None, t[1], int(t[2])) else 0

    def _r_start(self, t):
        if len(t) < 4 or not (t[1].isdigit() and t[2].isdigit()):
            return 0
        return 4 if self._exec(self.START_LEARNING, t[3], _text(t[:4]), int(t[1]), int(t[2])) else 0

    def _r_stop_learning(self, t):
        if len(t) < 2 or not t[1].lstrip("-").isdigit():
            return 0
        return 2 if self._exec(self.STOP_LEARNING, t[1], _text(t[:2])) else 0

    def _r_models_ready(self, t):
        if len(t) < 2 or not t[1].isdigit():
            return 0
        return 2 if self._exec(self.MODELS_READY, None, None, int(t[1])) else 0

    def _r_metrics(self, t):
        if len(t) < 6:
            return 0
        try:
            args = (t[1], int(t[2]), float(t[3]), float(t[4]))
        except ValueError:
            return 0
        return 6 if self._exec(self.METRICS, t[5], _text(t[:6]), *args) else 0

    def _r_vote(self, t):
        try:
            close = t.index(self.VOTE_TRAIN_SET_CLOSE)
            node, body, hash_ = t[1], t[2:close], t[close + 1]
            if len(body) % 2:
                return 0
            votes = {body[i]: int(body[i + 1]) for i in range(0, len(body), 2)}
        except (ValueError, IndexError):
            return 0
        return close + 2 if self._exec(self.VOTE_TRAIN_SET, hash_, _text(t[:close + 2]), node, votes) else 0

    def _r_models_aggregated(self, t):
        try:
            close = t.index(self.MODELS_AGGREGATED_CLOSE)
        except ValueError:
            return 0
        return close + 1 if self._exec(self.MODELS_AGGREGATED, None, None, list(t[1:close])) else 0

    def _r_model_initialized(self, t):
        return 1 if self._exec(self.MODEL_INITIALIZED, None, None) else 0

    def _r_transfer(self, t):
        return 1 if self._exec(self.TRANSFER_LEADERSHIP, None, None) else 0

    # ------------------------------------------------------------------ handshake / framing helpers
    @staticmethod
    def process_connection(message, callback: Callable[[str, int, bool, bool], None]) -> bool:
        """Validate a ``CONNECT`` handshake line and run ``callback(host, port, full, force)``."""
        if isinstance(message, (bytes, bytearray)):
            try:
                message = message.decode("utf-8")
            except UnicodeDecodeError:
                return False
        t = message.split()
        if len(t) > 4 and t[0] == CommunicationProtocol.CONN:
            try:
                callback(t[1], int(t[2]), t[3] == "1", t[4] == "1")
                return True
            except Exception:
                return False
        return False

    @staticmethod
    def check_collapse(msg: bytes) -> int:
        """Bytes of text glued in front of a binary PARAMS fragment (0 if none)."""
        pos = msg.find(CommunicationProtocol._PARAMS_B)
        if pos > 0:
            return pos
        return 0

    @staticmethod
    def check_params_incomplete(msg: bytes, block_size: int) -> int:
        """Missing bytes of a short PARAMS fragment (fragments are always BLOCK_SIZE long)."""
        if msg[:len(CommunicationProtocol._PARAMS_B)] == CommunicationProtocol._PARAMS_B and len(msg) < block_size:
            return block_size - len(msg)
        return 0

    # ------------------------------------------------------------------ builders
    @staticmethod
    def generate_hased_message(msg: str) -> bytes:
        """Append a fresh unsigned message id (kept under the reference's spelling)."""
        return (msg + " " + str(secrets.randbits(63)) + "\n").encode("utf-8")

    generate_hashed_message = generate_hased_message

    @staticmethod
    def build_beat_msg(node):
        return CommunicationProtocol.generate_hased_message(CommunicationProtocol.BEAT + " " + node)

    @staticmethod
    def build_role_msg(node, role):
        return CommunicationProtocol.generate_hased_message(CommunicationProtocol.ROLE + " " + node + " " + role)

    @staticmethod
    def build_stop_msg():
        return (CommunicationProtocol.STOP + "\n").encode("utf-8")

    @staticmethod
    def build_connect_to_msg(ip, port):
        return (CommunicationProtocol.CONN_TO + " " + ip + " " + str(port) + "\n").encode("utf-8")

    @staticmethod
    def build_start_learning_msg(rounds, epochs):
        return CommunicationProtocol.generate_hased_message(
            CommunicationProtocol.START_LEARNING + " " + str(rounds) + " " + str(epochs))

    @staticmethod
    def build_stop_learning_msg():
        return CommunicationProtocol.generate_hased_message(CommunicationProtocol.STOP_LEARNING)

    @staticmethod
    def build_models_ready_msg(round):
        return (CommunicationProtocol.MODELS_READY + " " + str(round) + "\n").encode("utf-8")

    @staticmethod
    def build_metrics_msg(node, round, loss, metric):
        return CommunicationProtocol.generate_hased_message(
            " ".join([CommunicationProtocol.METRICS, node, str(round), str(loss), str(metric)]))

    @staticmethod
    def build_vote_train_set_msg(node, votes):
        body = "".join(" " + v[0] + " " + str(v[1]) for v in votes)
        return CommunicationProtocol.generate_hased_message(
            CommunicationProtocol.VOTE_TRAIN_SET + " " + node + body + " " + CommunicationProtocol.VOTE_TRAIN_SET_CLOSE)

    @staticmethod
    def build_models_aggregated_msg(nodes):
        body = "".join(" " + n for n in nodes)
        return (CommunicationProtocol.MODELS_AGGREGATED + body + " "
                + CommunicationProtocol.MODELS_AGGREGATED_CLOSE + "\n").encode("utf-8")

    @staticmethod
    def build_model_initialized_msg():
        return (CommunicationProtocol.MODEL_INITIALIZED + "\n").encode("utf-8")

    @staticmethod
    def build_connect_msg(ip, port, broadcast, force):
        return (CommunicationProtocol.CONN + " " + ip + " " + str(port) + " " + str(broadcast) + " "
                + str(force) + "\n").encode("utf-8")

    @staticmethod
    def build_transfer_leadership_msg():
        return (CommunicationProtocol.TRANSFER_LEADERSHIP + "\n").encode("utf-8")

    @staticmethod
    def build_params_msg(data: bytes, block_size: int) -> List[bytes]:
        """Fragment a serialized model into ``PARAMS``-prefixed frames of exactly ``block_size`` bytes;
        the last one carries the tail, ``\\PARAMS`` and zero padding (reference builder:
        communication_protocol.py:737-769). One native pass; unlike the reference no frame is ever
        shorter than ``block_size``, which its fixed-size receiver needs."""
        lib = hostlib.load()
        n_frames = lib.p2pfl_params_frame_count(len(data), block_size)
        if n_frames == 0:
            raise ValueError("BLOCK_SIZE too small for PARAMS framing")
        out = ctypes.create_string_buffer(n_frames * block_size)
        last_len = ctypes.c_size_t(0)
        total = lib.p2pfl_params_build(bytes(data), len(data), block_size, out, ctypes.byref(last_len))
        raw = out.raw[:total]
        return [raw[i:i + block_size] for i in range(0, total, block_size)]


def _text(tokens) -> bytes:
    return (" ".join(tokens) + "\n").encode("utf-8")


_P = CommunicationProtocol
_RULES = {
    _P.BEAT: _P._r_beat, _P.ROLE: _P._r_role, _P.STOP: _P._r_stop, _P.CONN_TO: _P._r_conn_to,
    _P.START_LEARNING: _P._r_start, _P.STOP_LEARNING: _P._r_stop_learning,
    _P.MODELS_READY: _P._r_models_ready, _P.METRICS: _P._r_metrics, _P.VOTE_TRAIN_SET: _P._r_vote,
    _P.MODELS_AGGREGATED: _P._r_models_aggregated, _P.MODEL_INITIALIZED: _P._r_model_initialized,
    _P.TRANSFER_LEADERSHIP: _P._r_transfer,
}
