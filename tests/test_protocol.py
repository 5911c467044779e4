"""Wire protocol: round trips of all 15 message kinds, multi-command buffers, framing, dedup."""
import os

import pytest

from p2pfl_amd.communication_protocol import CommunicationProtocol as CP
from p2pfl_amd.config import Config


class Rec:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def execute(self, *args):
        self.log.append((self.name, args))


def make():
    log = []
    heads = [CP.BEAT, CP.ROLE, CP.STOP, CP.CONN_TO, CP.START_LEARNING, CP.STOP_LEARNING, CP.PARAMS, CP.MODELS_READY,
             CP.METRICS, CP.VOTE_TRAIN_SET, CP.MODELS_AGGREGATED, CP.MODEL_INITIALIZED, CP.TRANSFER_LEADERSHIP]
    return CP({h: Rec(log, h) for h in heads}, Config(participant={"AMOUNT_LAST_MESSAGES_SAVED": 5})), log


def test_every_text_message_round_trips():
    cp, log = make()
    msgs = [
        (CP.build_beat_msg("1.2.3.4:5"), (CP.BEAT, ("1.2.3.4:5",))),
        (CP.build_role_msg("n:1", "trainer"), (CP.ROLE, ("n:1", "trainer"))),
        (CP.build_stop_msg(), (CP.STOP, ())),
        (CP.build_connect_to_msg("10.0.0.1", 4000), (CP.CONN_TO, ("10.0.0.1", 4000))),
        (CP.build_start_learning_msg(3, 2), (CP.START_LEARNING, (3, 2))),
        (CP.build_stop_learning_msg(), (CP.STOP_LEARNING, ())),
        (CP.build_models_ready_msg(7), (CP.MODELS_READY, (7,))),
        (CP.build_metrics_msg("n:1", 2, 0.5, 0.9), (CP.METRICS, ("n:1", 2, 0.5, 0.9))),
        (CP.build_vote_train_set_msg("n:1", [("a:1", 3), ("b:2", 1)]), (CP.VOTE_TRAIN_SET, ("n:1", {"a:1": 3, "b:2": 1}))),
        (CP.build_models_aggregated_msg(["a:1", "b:2"]), (CP.MODELS_AGGREGATED, (["a:1", "b:2"],))),
        (CP.build_model_initialized_msg(), (CP.MODEL_INITIALIZED, ())),
        (CP.build_transfer_leadership_msg(), (CP.TRANSFER_LEADERSHIP, ())),
    ]
    for raw, expect in msgs:
        executed, err = cp.process_message(raw)
        assert not err, raw
        assert log[-1] == expect
    # all in one buffer (commands glued together by TCP), fresh protocol => nothing de-duplicated
    cp2, log2 = make()
    executed, err = cp2.process_message(b"".join(m for m, _ in msgs))
    assert not err and [e for e in log2] == [e for _, e in msgs]
    assert len(executed) == 6          # the six gossiped kinds come back for re-flooding


def test_message_ids_are_unsigned_and_deduplicated():
    cp, log = make()
    m = CP.build_stop_learning_msg()
    assert m.split()[1].isdigit()       # the reference's hash() ids can be negative and fail isdigit()
    assert cp.process_message(m) == ({m.split()[1].decode(): m}, False)
    executed, err = cp.process_message(m)
    assert executed == {} and not err and len(log) == 1
    for i in range(6):                  # the window holds AMOUNT_LAST_MESSAGES_SAVED ids
        cp.process_message(CP.build_beat_msg(f"n:{i}"))
    cp.process_message(m)
    assert len([e for e in log if e[0] == CP.STOP_LEARNING]) == 2


@pytest.mark.parametrize("bad", [b"BEAT onlynode\n", b"START_LEARNING x 1 5\n", b"CONNECT_TO h notaport\n", b"WHAT\n",
                                 b"METRICS n 1 x y 4\n", b"MODELS_AGGREGATED a b\n", b"\xff\xfe\x00garbage"])
def test_malformed_messages_flag_an_error(bad):
    cp, _ = make()
    assert cp.process_message(bad)[1] is True


def test_connect_handshake():
    got = []
    ok = CP.process_connection(CP.build_connect_msg("127.0.0.1", 4555, "1", "0"), lambda h, p, fu, fo: got.append((h, p, fu, fo)))
    assert ok and got == [("127.0.0.1", 4555, True, False)]
    assert not CP.process_connection(b"CONNECT 1.2.3.4\n", lambda *a: None)
    assert not CP.process_connection(b"BEAT a b c d\n", lambda *a: None)


@pytest.mark.parametrize("n", [0, 1, 2029, 2030, 2042, 2043, 70000])
def test_params_framing_and_reassembly(n):
    cp, log = make()
    data = os.urandom(n).replace(b"\\PARAMS", b"_______").replace(b"PARAMS", b"______")
    frames = CP.build_params_msg(data, 2048)
    assert all(len(f) == 2048 and f.startswith(b"PARAMS") for f in frames)
    body = b""
    for f in frames:
        _, err = cp.process_message(f)
        assert not err
        name, (seg, done) = log[-1]
        body += seg
    assert done and body == data
    assert CP.check_collapse(b"BEAT a 1\nPARAMSxyz") == 9 and CP.check_collapse(b"PARAMSxyz") == 0
    assert CP.check_params_incomplete(b"PARAMS123", 2048) == 2048 - 9
