import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running integration test")


def pytest_collection_modifyitems(config, items):
    # parameter combinations that do not exist are not tests: the co-resident (small-footprint) convolution variants
    # are fp32-MFMA kernels only
    items[:] = [it for it in items
                if not (getattr(it, "callspec", None) is not None and it.callspec.params.get("cores") is True
                        and it.callspec.params.get("prec", "fp32-mfma") != "fp32-mfma")]
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
def _seed_per_test(request):
    """Random test data must not depend on which tests ran before: a GPU test draws from torch's global generators, so
    adding or deselecting one case shifted the data of every later case (and a marginal tolerance elsewhere failed).
    Every GPU test starts from a seed derived from its own id."""
    if "gpu" in request.keywords:
        import zlib

        import torch
        torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7fffffff)
    yield


@pytest.fixture(autouse=True)
def _stream_watchdog(request):
    """If a GPU test runs longer than P2PFL_STREAM_WATCHDOG seconds (default 180, 0 = off), report which HIP
    streams still hold work (the engines' auxiliary streams and the default stream) and leave - a stalled
    device wait cannot be interrupted from Python otherwise."""
    limit = float(os.environ.get("P2PFL_STREAM_WATCHDOG", "180") or 0)
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
        lines = [f"STALL in {request.node.nodeid} (> {limit:.0f} s)"]
        try:
            for role, st in E.live_aux_streams():
                lines.append(f"  stream {role}: {'idle' if st.query() else 'BUSY'}")
            lines.append(f"  default stream: {'idle' if torch.cuda.default_stream().query() else 'BUSY'}")
        except Exception as e:      # the report must get out whatever the runtime says
            lines.append(f"  (stream query failed: {e!r})")
        text = "\n".join(lines) + "\n"
        # pytest captures fds 1 and 2: write the report to a file as well, and to the real terminal
        try:
            with open(os.environ.get("P2PFL_STALL_LOG", "gpu_stall_report.log"), "a") as f:
                f.write(text)
                import faulthandler
                faulthandler.dump_traceback(file=f)
        except OSError:
            pass
        capman = request.config.pluginmanager.getplugin("capturemanager")
        if capman is not None:
            try:
                capman.suspend_global_capture(in_=True)
            except Exception:
                pass
        sys.stderr.write("\n" + text)
        sys.stderr.flush()
        os._exit(3)

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    yield
    done.set()
