import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running integration test")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _stream_watchdog(request):
    """If a GPU test runs longer than P2PFL_STREAM_WATCHDOG seconds (default 120, 0 = off), report which HIP
    streams still hold work (the engines' auxiliary streams and the default stream) and leave - a stalled
    device wait cannot be interrupted from Python otherwise."""
    limit = float(os.environ.get("P2PFL_STREAM_WATCHDOG", "120") or 0)
    if limit <= 0 or "gpu" not in request.keywords:
        yield
        return
    import threading

    import torch

    from p2pfl_amd.runtime import engine as E
    done = threading.Event()

    def watch():
        if done.wait(limit):
            return
        print(f"\nSTALL in {request.node.nodeid}", flush=True)
        for role, st in E.live_aux_streams():
            print("  stream", role, "idle" if st.query() else "BUSY", flush=True)
        print("  default stream", "idle" if torch.cuda.default_stream().query() else "BUSY", flush=True)
        import faulthandler
        faulthandler.dump_traceback()
        os._exit(3)

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    yield
    done.set()
