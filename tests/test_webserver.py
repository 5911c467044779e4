"""Frontend tests with Flask's test client (the reference ships no webserver tests; routes and
payloads follow fedstellar/webserver/app.py:60-723)."""
import datetime
import importlib
import io
import json
import os
import zipfile

import pytest

flask = pytest.importorskip("flask")


@pytest.fixture()
def web(tmp_path, monkeypatch):
    monkeypatch.setenv("FEDSTELLAR_DB_DIR", str(tmp_path / "db"))
    monkeypatch.setenv("FEDSTELLAR_CONFIG_DIR", str(tmp_path / "config"))
    monkeypatch.setenv("FEDSTELLAR_LOGS_DIR", str(tmp_path / "logs"))
    monkeypatch.setenv("FEDSTELLAR_ADMIN_PASSWORD", "s3cret")
    from p2pfl_amd.webserver import app as appmod
    appmod = importlib.reload(appmod)
    appmod.app.config.update(TESTING=True, config_dir=str(tmp_path / "config"), log_dir=str(tmp_path / "logs"))
    os.makedirs(tmp_path / "config", exist_ok=True)
    os.makedirs(tmp_path / "logs", exist_ok=True)
    appmod.db.init_db()
    return appmod, appmod.app.test_client(), tmp_path


def _login(client, user="admin", pwd="s3cret"):
    return client.post("/login", data={"user": user, "password": pwd})


def test_login_logout_and_guards(web):
    _, c, _ = web
    assert c.get("/").status_code == 200
    assert c.get("/scenario/").status_code == 401
    assert _login(c, pwd="wrong").status_code == 401
    assert _login(c).status_code == 302
    assert c.get("/scenario/").status_code == 200
    assert c.get("/scenario/deployment/").status_code == 200
    assert c.get("/logout/").status_code == 302
    assert c.get("/scenario/").status_code == 401
    assert c.get("/nope").status_code == 404


def test_admin_user_crud_and_notes(web):
    appmod, c, _ = web
    _login(c)
    assert c.get("/admin/").status_code == 200
    assert c.post("/add_user", data={"user": "alice", "password": "pw", "role": "user"}).status_code == 302
    assert "ALICE" in appmod.db.list_users()
    assert c.post("/add_user", data={"user": "alice", "password": "pw", "role": "user"}).status_code == 403
    assert c.post("/add_user", data={"user": "a; DROP TABLE users", "password": "pw"}).status_code == 403
    c.get("/logout/")
    assert _login(c, "alice", "pw").status_code == 302
    assert c.get("/admin/").status_code == 401
    c.post("/write_note", data={"text_note_to_take": "hello"})
    notes = appmod.db.read_note_from_db("alice")
    assert len(notes) == 1 and notes[0][2] == "hello"
    c.get("/logout/")
    _login(c)
    assert c.get(f"/delete_note/{notes[0][0]}").status_code == 401     # not the owner
    assert c.get("/delete_user/alice/").status_code == 302
    assert "ALICE" not in appmod.db.list_users() and appmod.db.read_note_from_db("alice") == []
    assert c.get("/delete_user/admin/").status_code == 403


def _participant(idx, scenario, port):
    from p2pfl_amd.config.config import default_participant
    p = default_participant()
    p["scenario_args"].update(name=scenario, federation="DFL")
    p["device_args"].update(idx=idx, uid=f"uid{idx}", role="aggregator")
    p["network_args"].update(ip="127.0.0.1", port=port, neighbors=" ".join(f"127.0.0.1:{q}" for q in (46000, 46001) if q != port))
    return p


def test_node_update_and_monitoring_online_rule(web):
    appmod, c, tmp = web
    name = "fedstellar_DFL_test"
    appmod.db.scenario_update_record(name, "now", "", "t", "d", "running", "")
    os.makedirs(tmp / "config" / name)
    # the ingest routes carry no session: without the scenario's token (controller -> .token + participant configs,
    # node -> X-Scenario-Token) nothing is written
    body = json.dumps(_participant(0, name, 46000))
    assert c.post(f"/scenario/{name}/node/update", data=body, content_type="application/json").status_code == 403
    (tmp / "config" / name / ".token").write_text("s3cret")
    assert c.post(f"/scenario/{name}/node/update", data=body, content_type="application/json",
                  headers={"X-Scenario-Token": "wrong"}).status_code == 403
    assert not (tmp / "config" / name / "participant_0.json").exists()
    tok = {"X-Scenario-Token": "s3cret"}
    for i in range(2):
        r = c.post(f"/scenario/{name}/node/update", data=json.dumps(_participant(i, name, 46000 + i)),
                   content_type="application/json", headers=tok)
        assert r.status_code == 200 and r.get_json()["node_uid"] == f"uid{i}"
    assert (tmp / "config" / name / "participant_1.json").exists()
    assert c.post(f"/scenario/{name}/node/update", data="{}", content_type="application/json", headers=tok).status_code == 400
    # names made of dots address the config directory itself / its parent: refused on every route
    for bad in (".", "..", "%2e", "%2e%2e"):
        assert c.post(f"/scenario/{bad}/node/update", data=body, content_type="application/json", headers=tok).status_code \
            in (400, 404, 308)
        assert c.post(f"/scenario/{bad}/node/x/update/logs", data="x", headers=tok).status_code in (400, 404, 308)
    d = c.get(f"/api/scenario/{name}/monitoring").get_json()
    assert d["scenario_status"] == "running" and [n["status"] for n in d["nodes_table"]] == [True, True]
    assert d["nodes_table"][0]["neighbors"] == "127.0.0.1:46001"
    # a node that reported > 20 s ago is offline
    old = (datetime.datetime.now() - datetime.timedelta(seconds=60)).strftime("%Y-%m-%d %H:%M:%S.%f")
    appmod.db.update_node_record("uid1", 1, "127.0.0.1", 46001, "aggregator", "127.0.0.1:46000", 0, 0, old, "DFL", name)
    d = c.get(f"/api/scenario/{name}/monitoring").get_json()
    assert [n["status"] for n in d["nodes_table"]] == [True, False]
    assert c.get("/api/scenario/unknown/monitoring").status_code == 404
    assert c.get("/api/scenario/..%2Fetc/monitoring").status_code in (400, 404)
    # HTML page (renders the topology image from the reported neighbours)
    assert c.get(f"/scenario/{name}/monitoring").status_code == 401
    _login(c)
    assert c.get(f"/scenario/{name}/monitoring").status_code == 200
    img = c.get(f"/scenario/{name}/topology/image/")
    assert img.status_code in (200, 404)
    rows = c.get("/api/scenario/").get_json()
    assert rows[0]["name"] == name and rows[0]["status"] == "running"


def test_logs_statistics_and_download(web):
    _, c, tmp = web
    name = "scen"
    ldir = tmp / "logs" / name
    os.makedirs(ldir / "metrics" / "participant_0")
    (ldir / "participant_0.log").write_text("\x1b[32mline <one>\x1b[0m\nline two\n")
    (ldir / "metrics" / "participant_0" / "metrics.csv").write_text(
        "time,step,name,value\n1.0,0,Test/Loss,2.3\n2.0,1,Test/Loss,1.1\n2.0,1,TestEpoch/Accuracy,0.8\n")
    assert c.get(f"/scenario/{name}/node/0/infolog").status_code == 401
    m = c.get(f"/api/scenario/{name}/statistics/").get_json()
    assert m["participant_0"]["Test/Loss"] == [[0, 2.3], [1, 1.1]]
    _login(c)
    r = c.get(f"/scenario/{name}/node/0/infolog")          # whole file, as a download
    assert r.status_code == 200 and r.data.endswith(b"line two\n") and "attachment" in r.headers["Content-Disposition"]
    r = c.get(f"/scenario/{name}/node/0/infolog/1")        # live tail: last n lines, ANSI stripped, escaped
    assert r.status_code == 200 and r.data == b"line two\n"
    r = c.get(f"/scenario/{name}/node/0/infolog/50")
    assert b"line &lt;one&gt;" in r.data and b"\x1b" not in r.data
    assert c.get(f"/scenario/{name}/node/7/infolog/5").data == b"No logs available"
    assert c.get(f"/scenario/{name}/node/0/errorlog").status_code == 404
    assert c.get(f"/scenario/{name}/node/0/other").status_code == 404
    # log push from a remote node + controller log view / erase
    assert c.post(f"/scenario/{name}/node/abc123/update/logs", data="remote line\n").status_code == 403
    os.makedirs(tmp / "config" / name, exist_ok=True)
    (tmp / "config" / name / ".token").write_text("tok")
    assert c.post(f"/scenario/{name}/node/abc123/update/logs", data="remote line\n",
                  headers={"X-Scenario-Token": "tok"}).status_code == 200
    assert (ldir / "remote_abc123.log").read_text() == "remote line\n"
    assert c.get("/logs").status_code == 404
    (tmp / "logs" / "server.log").write_text("controller says hi\n")
    assert c.get("/logs").data == b"controller says hi\n"
    assert c.get("/logs/erase").status_code == 302
    assert (tmp / "logs" / "server.log").read_text().startswith("Fedstellar Core Logs")
    assert b"TestEpoch/Accuracy" in c.get(f"/scenario/{name}/statistics/").data
    z = c.get(f"/scenario/{name}/statistics/download")
    assert z.status_code == 200
    assert zipfile.ZipFile(io.BytesIO(z.data)).namelist() == ["participant_0/metrics.csv"]


def test_deployment_run_builds_scenario_and_calls_launcher(web, monkeypatch):
    appmod, c, tmp = web
    started = {}
    monkeypatch.setattr(appmod.Controller, "start_nodes_cmd", lambda self, idx: started.update(mode="cmd", idx=idx, n=self.n_nodes))
    monkeypatch.setattr(appmod.Controller, "start_nodes_batched", lambda self: started.update(mode="batched", n=self.n_nodes))
    monkeypatch.setattr(appmod.Controller, "killdockers", staticmethod(lambda: None))
    body = {"scenario_title": "T", "scenario_description": "D", "n_nodes": 3, "federation": "SDFL", "topology": "ring",
            "matrix": None, "simulation": True, "docker": False, "rounds": 2, "epochs": 1, "dataset": "MNIST", "model": "MLP",
            "accelerator": "cpu", "engine": "process",
            "nodes": {str(i): {"id": i, "ip": "127.0.0.1", "ipdemo": "", "port": 47000 + i,
                               "role": "aggregator" if i == 0 else "trainer", "start": i == 0} for i in range(3)}}
    assert c.post("/scenario/deployment/run", data=json.dumps(body), content_type="application/json").status_code == 401
    _login(c)
    r = c.post("/scenario/deployment/run", data=json.dumps(body), content_type="application/json")
    assert r.status_code == 200, r.data
    name = r.get_json()["scenario_name"]
    assert name.startswith("fedstellar_SDFL_") and started == {"mode": "cmd", "idx": 0, "n": 3}
    cdir = tmp / "config" / name
    assert (cdir / "controller.json").exists()
    p1 = json.loads((cdir / "participant_1.json").read_text())
    assert p1["scenario_args"]["rounds"] == 2 and p1["device_args"]["role"] == "trainer"
    assert p1["network_args"]["port"] == 47001 and len(p1["network_args"]["neighbors"].split()) == 2      # ring
    assert p1["scenario_args"]["name"] == name and p1["device_args"]["uid"]
    assert appmod.db.get_running_scenario()[0] == name
    # the whole scenario as one document (config/scenario.py), without the ingest token or credentials
    from p2pfl_amd.config.scenario import expand_scenario
    doc = c.get(f"/scenario/{name}/scenario.json")
    assert doc.status_code == 200
    parts = expand_scenario(doc.get_json())
    assert len(parts) == 3 and [p["network_args"]["port"] for p in parts] == [47000, 47001, 47002]
    assert all("controller_token" not in p["scenario_args"] and p["device_args"]["password"] == "" for p in parts)
    assert parts[1]["network_args"]["neighbors"] == p1["network_args"]["neighbors"]
    assert (cdir / "scenario.json").exists()                     # written by the controller beside the participant files
    assert c.get("/scenario/nope/scenario.json").status_code == 404
    # stop, re-run from the stored controller.json, remove
    assert c.get(f"/scenario/{name}/stop").status_code == 302
    assert appmod.db.get_running_scenario() is None
    started.clear()
    assert c.get(f"/scenario/{name}/deployment/reload").status_code == 302
    assert started["n"] == 3 and appmod.db.get_running_scenario()[0] == name
    assert c.get(f"/scenario/{name}/remove").status_code == 302
    assert appmod.db.get_scenario_by_name(name) is None and not cdir.exists()


def test_image_upload_view_delete(web):
    appmod, c, tmp = web
    png = b"\x89PNG\r\n\x1a\n" + b"0" * 32
    assert c.post("/upload_image", data={"file": (io.BytesIO(png), "plot.png")}).status_code == 401
    _login(c)
    assert c.post("/upload_image", data={"file": (io.BytesIO(png), "plot.png")},
                  content_type="multipart/form-data").status_code == 302
    assert c.post("/upload_image", data={"file": (io.BytesIO(b"x"), "evil.sh")},
                  content_type="multipart/form-data").status_code == 302          # ignored: extension not allowed
    imgs = appmod.db.list_images_for_user("admin")
    assert len(imgs) == 1 and imgs[0][2] == "plot.png"
    uid = imgs[0][0]
    assert b"plot.png" in c.get("/scenario/x/private/").data
    assert c.get(f"/images/{uid}/").data == png
    # another user cannot see or delete it
    c.post("/add_user", data={"user": "bob", "password": "pw", "role": "user"})
    c.get("/logout/")
    _login(c, "bob", "pw")
    assert c.get(f"/images/{uid}/").status_code == 401 and c.get(f"/delete_image/{uid}").status_code == 401
    c.get("/logout/")
    _login(c)
    assert c.get(f"/delete_image/{uid}").status_code == 302
    assert appmod.db.list_images_for_user("admin") == [] and c.get(f"/images/{uid}/").status_code == 401


def test_pages_render_with_bundled_scripts(web):
    """The designer / monitoring / statistics pages use only scripts shipped in static/ (no CDN: the
    target machines have no network)."""
    appmod, c, tmp = web
    _login(c)
    page = c.get("/scenario/deployment/").data.decode()
    assert "js/graph.js" in page and "TopologyGraph.generate" in page and "http://" not in page.replace("http://www.w3.org", "")
    for path in ("/static/js/graph.js", "/static/js/charts.js", "/static/style.css"):
        r = c.get(path)
        assert r.status_code == 200 and len(r.data) > 200
    name = "scen2"
    appmod.db.scenario_update_record(name, "now", "", "t", "d", "running", "")
    appmod.db.update_node_record("u0", 0, "127.0.0.1", 46000, "aggregator", "127.0.0.1:46001", 40.4, -3.7,
                                 datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f"), "DFL", name)
    mon = c.get(f"/scenario/{name}/monitoring").data.decode()
    assert "geoMap" in mon and f"/scenario/{name}/node/" in mon and "infolog/40" in mon
    d = c.get(f"/api/scenario/{name}/monitoring").get_json()
    assert d["nodes_table"][0]["latitude"] == "40.4"
    assert b"lineChart" in c.get(f"/scenario/{name}/statistics/").data


def test_designer_scripts_parse_and_graph_editing_works(web, tmp_path):
    """The designer's scripts are syntactically valid JavaScript and the topology widget grows / shrinks / projects a
    federation correctly (run under node with a stub canvas; skipped where node is not installed)."""
    import re
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed")
    appmod, c, tmp = web
    _login(c)
    page = c.get("/scenario/deployment/").data.decode()
    for must in ("view3d", "mode_expert", "ne_add", "ne_del", "adj_table", "save_json", "load_json", 'id="confirm"'):
        assert must in page
    scripts = re.findall(r"<script>(.*?)</script>", page, flags=re.S)
    assert scripts
    inline = tmp_path / "inline.js"
    inline.write_text("\n".join(scripts))
    graph_js = os.path.join(os.path.dirname(appmod.__file__), "static", "js", "graph.js")
    for f in (str(inline), graph_js, os.path.join(os.path.dirname(graph_js), "charts.js")):
        r = subprocess.run([node, "--check", f], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    driver = tmp_path / "drive.js"
    driver.write_text("""
const ctx = new Proxy({}, {get: (t, k) => (k in t ? t[k] : () => {}), set: (t, k, v) => { t[k] = v; return true; }});
global.window = global;
const canvas = {width: 520, height: 520, getContext: () => ctx, addEventListener: () => {}, getBoundingClientRect: () => ({left: 0, top: 0, width: 520, height: 520})};
global.window.addEventListener = () => {};
require(process.argv[2]);
let changes = 0;
const g = new TopologyGraph(canvas, {editable: true, onChange: () => changes++});
g.setMatrix(TopologyGraph.generate('ring', 5, 7), ['aggregator', 'trainer', 'trainer', 'trainer', 'trainer']);
const j = g.addNode(2, 'trainer');
let m = g.matrix();
if (j !== 5 || m.length !== 6 || m[5][2] !== 1 || m[2][5] !== 1 || m[5].reduce((a, b) => a + b) !== 1) throw new Error('addNode');
if (!g.removeNode(0) || g.matrix().length !== 5 || g.roles.length !== 5 || g.roles[0] !== 'trainer') throw new Error('removeNode');
m = g.matrix();
for (let a = 0; a < 5; a++) for (let b = 0; b < 5; b++) if (m[a][b] !== m[b][a]) throw new Error('symmetry');
g.set3d(true, false);
const p = g._proj(1);
if (!(isFinite(p[0]) && isFinite(p[1]) && p[2] > 0.5 && p[2] < 2)) throw new Error('projection');
g.linkDistance = 0.5; g.draw(); g.set3d(false, false);
const q = g._proj(1);
if (Math.abs(q[0] - (260 + (g.pos[1][0] - 260) * 0.5)) > 1e-9) throw new Error('distance scaling');
while (g.n > 2) g.removeNode(g.n - 1);
if (g.removeNode(0)) throw new Error('a federation keeps two nodes');
if (changes < 4) throw new Error('onChange');
console.log('ok');
""")
    r = subprocess.run([node, str(driver), graph_js], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_remove_refuses_dot_names_and_keeps_the_roots(web):
    """/scenario/%2e/remove used to rmtree the whole config and log directories (".." their parents)."""
    appmod, c, tmp = web
    from p2pfl_amd.controller import Controller
    (tmp / "config" / "keep").mkdir(parents=True, exist_ok=True)
    (tmp / "logs" / "keep").mkdir(parents=True, exist_ok=True)
    _login(c)
    for bad in (".", "..", "%2e", "%2e%2e", "..%2f..", "a%2f..%2f.."):
        assert c.get(f"/scenario/{bad}/remove").status_code in (400, 404, 308)
    assert (tmp / "config" / "keep").is_dir() and (tmp / "logs" / "keep").is_dir()
    os.environ["FEDSTELLAR_CONFIG_DIR"], os.environ["FEDSTELLAR_LOGS_DIR"] = str(tmp / "config"), str(tmp / "logs")
    for bad in (".", "..", "", "a/..", "../x"):
        with pytest.raises(ValueError):
            Controller.remove_files_by_scenario(bad)
    assert (tmp / "config" / "keep").is_dir() and (tmp / "logs" / "keep").is_dir()
    Controller.remove_files_by_scenario("keep")
    assert not (tmp / "config" / "keep").exists() and (tmp / "config").is_dir()


def test_designer_node_fields_topology_json_and_event_file_statistics(web, monkeypatch):
    """Per-node network shaping / coordinates from the designer land in the participant files (validated: they end
    up on a tcset command line); the live topology comes back in the reference's topology.json shape with
    coordinates; the statistics page reads TensorBoard event files without TensorBoard."""
    appmod, c, tmp = web
    started = {}
    monkeypatch.setattr(appmod.Controller, "start_nodes_cmd", lambda self, idx: started.update(mode="cmd"))
    monkeypatch.setattr(appmod.Controller, "start_nodes_batched",
                        lambda self: started.update(mode="batched", cp=self.control_plane, gw=self.gateways))
    monkeypatch.setattr(appmod.Controller, "killdockers", staticmethod(lambda: None))
    nodes = {str(i): {"id": i, "ip": "127.0.0.1", "ipdemo": "", "port": 47100 + i, "role": "aggregator", "start": i == 0,
                      "interface": "eth1", "rate": "10Mbps", "loss": "0.5%", "delay": "20ms", "delay_distro": "5",
                      "latitude": "46.5", "longitude": str(7.0 + i)} for i in range(3)}
    body = {"scenario_title": "T", "scenario_description": "D", "n_nodes": 3, "federation": "DFL", "topology": "fully",
            "matrix": None, "simulation": True, "docker": False, "rounds": 1, "epochs": 1, "dataset": "MNIST", "model": "MLP",
            "accelerator": "cpu", "engine": "batched", "control_plane": True, "gateways": ["0:47500", "bad;rm"],
            "nodes": nodes}
    _login(c)
    bad = json.loads(json.dumps(body))
    bad["nodes"]["1"]["rate"] = "10Mbps; reboot"
    r = c.post("/scenario/deployment/run", data=json.dumps(bad), content_type="application/json")
    assert r.status_code == 400 and "node 1" in r.get_json()["error"]
    r = c.post("/scenario/deployment/run", data=json.dumps(body), content_type="application/json")
    assert r.status_code == 200, r.data
    name = r.get_json()["scenario_name"]
    assert started == {"mode": "batched", "cp": True, "gw": ["0:47500"]}
    p2 = json.loads((tmp / "config" / name / "participant_2.json").read_text())
    na = p2["network_args"]
    assert (na["interface"], na["rate"], na["loss"], na["delay"], na["delay-distro"]) == ("eth1", "10Mbps", "0.5%", "20ms", 5.0)
    assert float(p2["geo_args"]["latitude"]) == 46.5 and float(p2["geo_args"]["longitude"]) == 9.0
    # nodes report in -> topology.json
    tok = {"X-Scenario-Token": (tmp / "config" / name / ".token").read_text()}
    for i in range(3):
        cfg = json.loads((tmp / "config" / name / f"participant_{i}.json").read_text())
        assert c.post(f"/scenario/{name}/node/update", data=json.dumps(cfg), content_type="application/json",
                      headers=tok).status_code == 200
    t = c.get(f"/api/scenario/{name}/topology").get_json()
    assert [n["ipport"] for n in t["nodes"]] == [f"127.0.0.1:{47100 + i}" for i in range(3)]
    assert t["nodes"][0]["start"] is True and t["nodes"][1]["role"] == "aggregator" and t["nodes"][2]["status"] is True
    assert float(t["nodes"][2]["longitude"]) == 9.0
    assert len(t["links"]) == 6 and {"source": "127.0.0.1:47100", "target": "127.0.0.1:47101", "value": 1} in t["links"]
    assert not (tmp / "config" / name / "topology.json").exists()          # a GET writes nothing
    assert c.get("/api/scenario/../topology").status_code in (400, 404)
    # a reference-protocol node sends no X-Scenario-Token header: the token inside the posted config is accepted too
    cfg = json.loads((tmp / "config" / name / "participant_1.json").read_text())
    assert cfg["scenario_args"]["controller_token"] == tok["X-Scenario-Token"]
    assert c.post(f"/scenario/{name}/node/update", data=json.dumps(cfg), content_type="application/json").status_code == 200
    cfg["scenario_args"]["controller_token"] = "wrong"
    assert c.post(f"/scenario/{name}/node/update", data=json.dumps(cfg), content_type="application/json").status_code == 403
    # the topology document names addresses and roles: no session and no token -> 401
    with appmod.app.test_client() as anon:
        assert anon.get(f"/api/scenario/{name}/topology").status_code == 401
        assert anon.get(f"/api/scenario/{name}/topology", headers=tok).status_code == 200
    # event files only (no metrics.csv): the statistics API still has the scalars
    from p2pfl_amd.learning.pytorch.statisticslogger import FedstellarLogger
    lg = FedstellarLogger(str(tmp / "logs" / name), name="metrics", version="participant_1")
    lg.log_metrics_direct({"Train/Loss": 0.5, "TestEpoch/Accuracy": 0.75}, 0)
    lg.log_metrics_direct({"Train/Loss": 0.25}, 1)
    lg.finalize()
    m = c.get(f"/api/scenario/{name}/statistics/").get_json()
    assert m["participant_1"]["Train/Loss"] == [[0, 0.5], [1, 0.25]] and m["participant_1"]["TestEpoch/Accuracy"] == [[0, 0.75]]
    page = c.get(f"/scenario/{name}/statistics/")
    assert page.status_code == 200 and b"Train/Loss" in page.data
    assert c.get(f"/scenario/{name}/monitoring").status_code == 200
    assert b"Role presets" in c.get("/scenario/deployment/").data


def test_event_file_reader_rejects_corruption(tmp_path):
    from p2pfl_amd.learning.pytorch.statisticslogger import FedstellarLogger, read_event_file
    import glob
    lg = FedstellarLogger(str(tmp_path), name="metrics", version="participant_0")
    for s in range(5):
        lg.log_metrics_direct({"a/b": float(s)}, s)
    lg.finalize()
    path = glob.glob(str(tmp_path / "metrics" / "participant_0" / "events.out.tfevents.*"))[0]
    assert read_event_file(path) == {"a/b": [[s, float(s)] for s in range(5)]}
    raw = bytearray(open(path, "rb").read())
    open(path, "wb").write(raw[:-7])                       # a run still writing: the torn tail is ignored
    assert len(read_event_file(path)["a/b"]) == 4
    raw[40] ^= 0xFF
    open(path, "wb").write(raw)
    with pytest.raises(ValueError):
        read_event_file(path)
