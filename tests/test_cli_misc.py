"""CLI flags, Mender client, environment report (reference: app/main.py:9-48, fedstellar/config/mender.py:10-221,
fedstellar/utils/env.py:7-64)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_flags_match_the_reference():
    sys.path.insert(0, ROOT)
    from app.main import argparser
    a = argparser.parse_args([])
    # reference defaults (app/main.py:11-40): DFL, fully, ports 5000/5100; --simulation and --tracking are store_false
    assert (a.federation, a.topology, int(a.webport), int(a.statsport)) == ("DFL", "fully", 5000, 5100)
    assert a.simulation is True and a.tracking is True and a.docker is False and a.webserver is False and a.cloud is False
    b = argparser.parse_args(["-f", "CFL", "-t", "star", "-s", "-tr", "-d", "-w", "-wp", "8080", "--engine", "batched", "--gpus", "2"])
    assert (b.federation, b.topology, b.simulation, b.tracking, b.docker, b.webserver) == ("CFL", "star", False, False, True, True)
    assert b.webport == "8080" and b.engine == "batched" and b.gpus == 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "app", "main.py"), "--version"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().split()[-1][0].isdigit()


def test_cli_generates_and_runs_a_batched_scenario(tmp_path):
    """`app/main.py --nodes N --engine batched --wait` = scenario generation + launch + wait, no web UI."""
    env = dict(os.environ, P2PFL_START_DELAY="0.2")
    cmd = [sys.executable, os.path.join(ROOT, "app", "main.py"), "--nodes", "4", "--engine", "batched", "--gpus", "1",
           "--rounds", "1", "--epochs", "1", "--accelerator", "cpu", "--topology", "ring", "--scenario-name", "cli_scn",
           "--config", str(tmp_path / "config"), "--logs", str(tmp_path / "logs"), "--wait"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    cdir = tmp_path / "config" / "cli_scn"
    parts = sorted(cdir.glob("participant_*.json"))
    assert len(parts) == 4
    p0 = json.loads(parts[0].read_text())
    # ring (+ the reference's random "increase convergence" chords, p = 0.1 per pair: 2 or 3 neighbours of 4 nodes)
    nbrs = p0["network_args"]["neighbors"].split()
    assert p0["scenario_args"]["n_nodes"] == 4 and {"127.0.0.1:45001", "127.0.0.1:45003"} <= set(nbrs) and len(nbrs) <= 3
    assert sum(json.loads(p.read_text())["device_args"]["start"] for p in parts) == 1
    assert (tmp_path / "logs" / "cli_scn").is_dir()


class _Resp:
    def __init__(self, payload=None, text="tok", status=200):
        self._p, self.text, self.status_code = payload, text, status

    def json(self):
        return self._p

    def raise_for_status(self):
        if self.status_code >= 400:
            raise RuntimeError(self.status_code)


def test_mender_client_requests(monkeypatch):
    from p2pfl_amd.config.mender import Mender
    calls = []

    class FakeRequests:
        @staticmethod
        def post(url, **kw):
            calls.append(("POST", url, kw))
            return _Resp(text="jwt-token", status=201 if "deployments/deployments" in url else 200)

        @staticmethod
        def get(url, **kw):
            calls.append(("GET", url, kw))
            return _Resp(payload=[{"id": "dev1"}])

    m = Mender(server="https://mender.example/", user="u", password="p")
    monkeypatch.setattr(m, "_session", lambda: FakeRequests)
    assert m.get_token() == "jwt-token" and calls[0][1].endswith("/api/management/v1/useradm/auth/login")
    assert calls[0][2]["auth"] == ("u", "p")
    assert m.get_devices() == [{"id": "dev1"}]
    assert calls[-1][2]["headers"]["Authorization"] == "Bearer jwt-token"
    assert m.get_info_device("dev1") == [{"id": "dev1"}] and calls[-1][1].endswith("/inventory/devices/dev1")
    assert m.deploy_artifact_device("art", "dev1") == 201
    body = json.loads(calls[-1][2]["data"])
    assert body == {"artifact_name": "art", "devices": ["dev1"], "name": "art-dev1"}
    assert m.deploy_artifact_list("art", ["a", "b"]) == 201 and len(calls) == 5     # token is cached
    assert m.get_my_info() == [{"id": "dev1"}] and calls[-1][1].endswith("/useradm/users/me")
    assert m.get_info_deployment("d7") == [{"id": "dev1"}] and calls[-1][1].endswith("/deployments/deployments/d7")
    n = len(calls)
    assert m.renew_token() == "jwt-token" and len(calls) == n + 1 and calls[-1][0] == "POST"     # logs in again
    monkeypatch.setattr("shutil.which", lambda name: None)
    with pytest.raises(FileNotFoundError):
        Mender.generate_artifact("single-file", "art", "raspberrypi4", "/tmp/x")
    with pytest.raises(RuntimeError):
        Mender(server="", token="t").get_artifacts()
    # credentials come from the environment, never from the repository
    monkeypatch.setenv("MENDER_SERVER", "https://env.example")
    monkeypatch.setenv("MENDER_TOKEN", "envtok")
    assert Mender().get_token() == "envtok" and Mender().server == "https://env.example"


def test_collect_env_reports_framework_and_devices():
    from p2pfl_amd.utils.env import collect_env, gpu_stats
    info = collect_env()
    assert info["os"] and info["python"].count(".") == 2 and info["torch"]
    assert isinstance(info["gpus"], list) and isinstance(gpu_stats(), list)
    assert "cpu_percent" in info and 0 <= info["ram_percent"] <= 100
