"""Frontend: Flask web UI + REST API (reference: fedstellar/webserver/app.py:25-723).

Routes kept (same paths and payloads): login/logout, admin user CRUD, notes, scenario list
(``/scenario/``, ``/api/scenario/``), deployment designer + ``POST /scenario/deployment/run`` (same JSON
body the reference UI sends, deployment.html:538-555), monitoring page + JSON API with the
"online = reported < 20 s ago" rule, node status ingest ``POST /scenario/<n>/node/update``, per-node
log views, stop / remove / reload, topology image, metrics zip download, statistics.
The statistics page reads the nodes' ``metrics.csv`` files itself (TensorBoard is optional, see
``Controller.run_statistics``).
"""
from __future__ import annotations

import argparse
import datetime
import glob
import hashlib
import io
import json
import os
import re
import sys
import zipfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from flask import (Flask, abort, jsonify, redirect, render_template, request, send_file, session,  # noqa: E402
                   url_for)

from p2pfl_amd.controller import Controller  # noqa: E402
from p2pfl_amd.role import default_roles  # noqa: E402
from p2pfl_amd.utils.topologymanager import TopologyManager  # noqa: E402
from p2pfl_amd.webserver import config as webconfig  # noqa: E402
from p2pfl_amd.webserver import database as db  # noqa: E402

app = Flask(__name__)
app.config.from_object(webconfig)
app.config["config_dir"] = os.environ.get("FEDSTELLAR_CONFIG_DIR", os.path.join(os.getcwd(), "app", "config"))
app.config["log_dir"] = os.environ.get("FEDSTELLAR_LOGS_DIR", os.path.join(os.getcwd(), "app", "logs"))
app.config["python_path"] = os.environ.get("FEDSTELLAR_PYTHON_PATH", sys.executable)
app.config["statistics_port"] = os.environ.get("FEDSTELLAR_STATISTICS_PORT", 5100)
_RUNNING = {}          # scenario name -> Controller
_SAFE = re.compile(r"^[A-Za-z0-9_.\-]+$")


def _safe(name):
    """A single path component: letters, digits, ``_ . -`` - and never a name made of dots only ("." / ".." would
    address the config / log directory itself or its parent)."""
    if not name or not _SAFE.match(name) or set(name) <= {"."}:
        abort(400)
    return name


def _under(root, *parts):
    """``root/parts...`` with every part checked and the result verified to stay inside ``root`` (symlinks resolved)."""
    path = os.path.join(root, *[_safe(str(p)) for p in parts])
    real_root = os.path.realpath(root)
    if os.path.commonpath([real_root, os.path.realpath(path)]) != real_root or os.path.realpath(path) == real_root:
        abort(400)
    return path


def _ingest_allowed(scenario_name):
    """Node-ingest routes carry no session: they are authenticated by the scenario's token, which the controller
    writes into ``<config_dir>/<scenario>/.token`` and into every participant config
    (``scenario_args.controller_token``); nodes send it back as ``X-Scenario-Token``."""
    tok_path = os.path.join(_under(app.config["config_dir"], scenario_name), ".token")
    if not os.path.exists(tok_path):
        return False
    with open(tok_path) as f:
        want = f.read().strip()
    # header (this repo's nodes), or - for reference-protocol nodes, whose POST carries no such header - the same
    # token inside the posted participant config (``scenario_args.controller_token``, which the controller wrote
    # into the file the node was started from and which the node echoes back) or as ``?token=`` of the URL
    got = request.headers.get("X-Scenario-Token", "") or request.args.get("token", "")
    if not got:
        body = request.get_json(force=True, silent=True)
        if isinstance(body, dict):
            got = str((body.get("scenario_args") or {}).get("controller_token") or "")
    import hmac
    return bool(want) and hmac.compare_digest(want, got)


def _dirs():
    return app.config["config_dir"], app.config["log_dir"]


for code in (401, 403, 404, 405, 413):
    app.register_error_handler(code, (lambda c: (lambda e: (render_template("error.html", code=c), c)))(code))


@app.before_request
def _init():
    if not app.config.get("_db_ready"):
        db.init_db()
        app.config["_db_ready"] = True


# ------------------------------------------------------------------ session / users
@app.route("/")
def fedstellar_home():
    return render_template("index.html")


@app.route("/login", methods=["POST"])
def fedstellar_login():
    user, password = request.form.get("user", ""), request.form.get("password", "")
    if db.verify(user, password):
        info = db.get_user_info(user)
        session["user"], session["role"] = info[0], info[2]
        return redirect(url_for("fedstellar_scenario"))
    return abort(401)


@app.route("/logout/")
def fedstellar_logout():
    session.pop("user", None)
    session.pop("role", None)
    return redirect(url_for("fedstellar_home"))


@app.route("/admin/")
def fedstellar_admin():
    if session.get("role") != "admin":
        return abort(401)
    return render_template("admin.html", users=db.list_users(all_info=True))


@app.route("/add_user", methods=["POST"])
def fedstellar_add_user():
    if session.get("role") != "admin":
        return abort(401)
    user, pwd, role = request.form.get("user", ""), request.form.get("password", ""), request.form.get("role", "user")
    if not user.isalnum() or not pwd or user.upper() in db.list_users():
        return abort(403)
    db.add_user(user, pwd, role)
    return redirect(url_for("fedstellar_admin"))


@app.route("/delete_user/<user>/", methods=["GET", "POST"])
def fedstellar_delete_user(user):
    if session.get("role") != "admin" or user.upper() == "ADMIN":
        return abort(401 if session.get("role") != "admin" else 403)
    db.delete_user_from_db(user)
    return redirect(url_for("fedstellar_admin"))


@app.route("/scenario/<scenario_name>/private/")
def fedstellar_scenario_private(scenario_name):
    if "user" not in session:
        return abort(401)
    return render_template("private.html", scenario_name=_safe(scenario_name), notes=db.read_note_from_db(session["user"]),
                           images=db.list_images_for_user(session["user"]))


@app.route("/write_note", methods=["POST"])
def fedstellar_write_note():
    if "user" not in session:
        return abort(401)
    db.write_note_into_db(session["user"], request.form.get("text_note_to_take", "")[:10000])
    return redirect(request.referrer or url_for("fedstellar_scenario"))


@app.route("/delete_note/<note_id>", methods=["GET"])
def fedstellar_delete_note(note_id):
    if session.get("user") != db.match_user_id_with_note_id(note_id):
        return abort(401)
    db.delete_note_from_db(note_id)
    return redirect(request.referrer or url_for("fedstellar_scenario"))


# ------------------------------------------------------------------ images (reference app.py:131-192)
def _upload_dir():
    d = app.config["UPLOAD_FOLDER"]
    if not os.path.isabs(d):
        d = os.path.join(os.environ.get("FEDSTELLAR_DB_DIR") or os.path.dirname(os.path.abspath(__file__)), d)
    os.makedirs(d, exist_ok=True)
    return d


def allowed_file(filename: str) -> bool:
    return "." in filename and filename.rsplit(".", 1)[1].lower() in app.config["ALLOWED_EXTENSIONS"]


@app.route("/upload_image", methods=["POST"])
def fedstellar_upload_image():
    if "user" not in session:
        return abort(401)
    back = request.referrer or url_for("fedstellar_scenario")
    file = request.files.get("file")
    if file is None or file.filename == "" or not allowed_file(file.filename):
        return redirect(back)
    from werkzeug.utils import secure_filename
    filename = secure_filename(file.filename)
    upload_time = str(datetime.datetime.now())
    image_uid = hashlib.sha1((upload_time + filename).encode()).hexdigest()
    file.save(os.path.join(_upload_dir(), image_uid + "-" + filename))
    db.image_upload_record(image_uid, session["user"], filename, upload_time)
    return redirect(back)


@app.route("/images/<image_uid>/", methods=["GET"])
def fedstellar_get_image(image_uid):
    if session.get("user") is None or session.get("user") != db.match_user_id_with_image_uid(_safe(image_uid)):
        return abort(401)
    path = os.path.join(_upload_dir(), db.get_image_file_name(image_uid))
    return send_file(path) if os.path.exists(path) else abort(404)


@app.route("/delete_image/<image_uid>", methods=["GET"])
def fedstellar_delete_image(image_uid):
    if session.get("user") is None or session.get("user") != db.match_user_id_with_image_uid(_safe(image_uid)):
        return abort(401)
    name = db.get_image_file_name(image_uid)
    db.delete_image_from_db(image_uid)
    path = os.path.join(_upload_dir(), name)
    if os.path.exists(path):
        os.remove(path)
    return redirect(request.referrer or url_for("fedstellar_scenario"))


# ------------------------------------------------------------------ scenarios
def _scenario_rows():
    keys = ["name", "start_time", "end_time", "title", "description", "status", "network_subnet"]
    return [dict(zip(keys, r)) for r in db.get_all_scenarios()]


@app.route("/scenario/")
@app.route("/api/scenario/", endpoint="api_scenario")
def fedstellar_scenario():
    if "user" not in session and not request.path.startswith("/api"):
        return abort(401)
    rows = _scenario_rows()
    if request.path.startswith("/api"):
        return jsonify(rows)
    return render_template("scenario.html", scenarios=rows, running=db.get_running_scenario())


def _node_rows(scenario_name):
    keys = ["uid", "idx", "ip", "port", "role", "neighbors", "latitude", "longitude", "timestamp", "federation", "scenario"]
    out = []
    now = datetime.datetime.now()
    for r in db.list_nodes_by_scenario_name(scenario_name):
        d = dict(zip(keys, r))
        try:
            ts = datetime.datetime.strptime(d["timestamp"], "%Y-%m-%d %H:%M:%S.%f")
            d["status"] = (now - ts) <= datetime.timedelta(seconds=webconfig.NODE_ONLINE_SECONDS)
        except ValueError:
            d["status"] = False
        out.append(d)
    return out


@app.route("/scenario/<scenario_name>/monitoring")
@app.route("/api/scenario/<scenario_name>/monitoring", endpoint="api_monitoring")
def fedstellar_scenario_monitoring(scenario_name):
    _safe(scenario_name)
    scenario = db.get_scenario_by_name(scenario_name)
    nodes = _node_rows(scenario_name)
    if request.path.startswith("/api"):
        if scenario is None:
            return jsonify({"scenario_status": "not found", "nodes_table": []}), 404
        return jsonify({"scenario_status": scenario[5], "nodes_table": nodes, "scenario_name": scenario[0],
                        "title": scenario[3], "description": scenario[4]})
    if "user" not in session:
        return abort(401)
    if scenario is None:
        return abort(404)
    update_topology(scenario_name, nodes)
    return render_template("monitoring.html", scenario_name=scenario_name, scenario=scenario, nodes=nodes)


def update_topology(scenario_name, nodes):
    """Re-render topology.png from the neighbours the nodes last reported (reference app.py:367-381)."""
    if not nodes:
        return
    try:
        import numpy as np
        names = [f"{n['ip']}:{n['port']}" for n in nodes]
        m = np.zeros((len(nodes), len(nodes)))
        for i, n in enumerate(nodes):
            for nb in n["neighbors"].split():
                if nb in names:
                    m[i, names.index(nb)] = 1
        tm = TopologyManager(n_nodes=len(nodes), topology=m, scenario_name=scenario_name,
                             undirected_neighbor_num=max(len(nodes) - 1, 2))
        tm.add_nodes([(n["ip"], n["port"], n["role"], None) for n in nodes])
        os.makedirs(os.path.join(app.config["log_dir"], scenario_name), exist_ok=True)
        tm.draw_graph(path=os.path.join(app.config["log_dir"], scenario_name, "topology.png"))
    except Exception as e:
        app.logger.debug("topology update failed: %s", e)


@app.route("/api/scenario/<scenario_name>/topology")
def fedstellar_api_topology(scenario_name):
    """Live topology as the reference's 3-D graph JSON (``TopologyManager.update_topology_3d_json``,
    utils/topologymanager.py:321-354: nodes {uid,id,ipport,role,ip,port,start} + links {source,target,value}) built
    from the nodes' last status reports, plus each node's coordinates and liveness for the map overlay. Needs a
    session (or the scenario token): the document names every node's address and role. Read-only - nothing is
    written from this GET."""
    cdir = _under(app.config["config_dir"], scenario_name)
    if "user" not in session and not _ingest_allowed(scenario_name):
        return abort(401)
    rows = _node_rows(scenario_name)
    starts = {}
    for path in glob.glob(os.path.join(cdir, "participant_*.json")):
        try:
            with open(path) as f:
                pc = json.load(f)
            starts[int(pc["device_args"]["idx"])] = bool(pc["device_args"].get("start"))
        except (OSError, ValueError, KeyError, TypeError):
            continue
    participants = [{"device_args": {"uid": r["uid"], "idx": r["idx"], "role": r["role"], "start": starts.get(int(r["idx"]), False)},
                     "network_args": {"ip": r["ip"], "port": r["port"], "neighbors": r["neighbors"] or ""}} for r in rows]
    data = TopologyManager.update_topology_3d_json(participants) if os.path.isdir(cdir) else {"nodes": [], "links": []}
    known = {n["ipport"] for n in data["nodes"]}
    data["links"] = [l for l in data["links"] if l["target"] in known]
    for n, r in zip(data["nodes"], rows):
        n.update(latitude=r["latitude"], longitude=r["longitude"], status=r["status"], timestamp=r["timestamp"])
    return jsonify(data)


@app.route("/scenario/<scenario_name>/node/update", methods=["POST"])
def fedstellar_update_node(scenario_name):
    """Status ingest from the nodes: the body is the node's live participant config."""
    cdir = _under(app.config["config_dir"], scenario_name)
    if not _ingest_allowed(scenario_name):
        return abort(403)
    cfg = request.get_json(force=True, silent=True)
    if not cfg or "device_args" not in cfg or "network_args" not in cfg:
        return abort(400)
    ts = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f")
    if os.path.isdir(cdir):
        with open(os.path.join(cdir, f"participant_{int(cfg['device_args']['idx'])}.json"), "w") as f:
            json.dump(cfg, f, sort_keys=False, indent=2)
    db.update_node_record(cfg["device_args"]["uid"], cfg["device_args"]["idx"], cfg["network_args"]["ip"],
                          cfg["network_args"]["port"], cfg["device_args"]["role"], cfg["network_args"]["neighbors"],
                          cfg["geo_args"]["latitude"], cfg["geo_args"]["longitude"], ts,
                          cfg["scenario_args"]["federation"], scenario_name)
    return jsonify({"node_uid": cfg["device_args"]["uid"], "timestamp": ts})


def _tail(path, lines=200):
    if not os.path.exists(path):
        return None
    with open(path, errors="replace") as f:
        data = f.readlines()[-lines:]
    text = "".join(data)
    text = re.sub(r"\x1B\[[0-?]*[ -/]*[@-~]", "", text)
    return text.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")


_LOG_SUFFIX = {"infolog": ".log", "debuglog": "_debug.log", "errorlog": "_error.log"}


@app.route("/scenario/<scenario_name>/node/<int:idx>/<kind>")
def fedstellar_monitoring_log(scenario_name, idx, kind):
    """Whole log file of a node as a download: .../infolog, .../debuglog, .../errorlog
    (reference app.py:443-453,479-500)."""
    if "user" not in session:
        return abort(401)
    suffix = _LOG_SUFFIX.get(kind)
    if suffix is None:
        return abort(404)
    path = os.path.join(_under(app.config["log_dir"], scenario_name), f"participant_{idx}{suffix}")
    if not os.path.exists(path):
        return abort(404)
    return send_file(path, mimetype="text/plain", as_attachment=True)


@app.route("/scenario/<scenario_name>/node/<int:idx>/infolog/<int:number>")
def fedstellar_monitoring_log_x(scenario_name, idx, number):
    """Last ``number`` lines of a node's log, ANSI colours stripped and HTML-escaped (the live tail of
    the monitoring page, reference app.py:455-476)."""
    if "user" not in session:
        return abort(401)
    text = _tail(os.path.join(_under(app.config["log_dir"], scenario_name), f"participant_{idx}.log"),
                 max(1, min(int(number), 10000)))
    return app.response_class("No logs available" if text is None else text, mimetype="text/plain")


@app.route("/scenario/<scenario_name>/node/<uid>/update/logs", methods=["POST"])
def fedstellar_update_node_logs(scenario_name, uid):
    """Log push from a node that cannot share the log directory (commented out in the reference,
    app.py:407-416): the body is appended to ``<log_dir>/<scenario>/remote_<uid>.log``."""
    _safe(uid)
    d = _under(app.config["log_dir"], scenario_name)
    if not _ingest_allowed(scenario_name):
        return abort(403)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"remote_{uid}.log"), "a") as f:
        f.write(request.get_data(as_text=True)[:1 << 20])
    return "Logs received successfully", 200


@app.route("/logs", methods=["GET"])
def fedstellar_logs():
    """Controller / server log (reference app.py:418-427)."""
    if "user" not in session:
        return abort(401)
    path = os.path.join(app.config["log_dir"], "server.log")
    return send_file(path, mimetype="text/plain") if os.path.exists(path) else abort(404)


@app.route("/logs/erase", methods=["GET"])
def fedstellar_logs_erase():
    if "user" not in session:
        return abort(401)
    path = os.path.join(app.config["log_dir"], "server.log")
    if not os.path.exists(path):
        return abort(404)
    with open(path, "w") as f:
        f.write("Fedstellar Core Logs\n")
    return redirect(url_for("fedstellar_logs"))


@app.route("/scenario/<scenario_name>/topology/image/")
def fedstellar_monitoring_image(scenario_name):
    path = os.path.join(_under(app.config["log_dir"], scenario_name), "topology.png")
    return send_file(path, mimetype="image/png") if os.path.exists(path) else abort(404)


def stop_scenario(scenario_name):
    c = _RUNNING.pop(scenario_name, None)
    if c is not None:
        for p in c.processes:
            if p.poll() is None:
                p.terminate()
    Controller.killdockers()
    db.scenario_set_status_to_finished(scenario_name)


def stop_all_scenarios():
    for name in list(_RUNNING):
        stop_scenario(name)
    db.scenario_set_all_status_to_finished()


@app.route("/scenario/<scenario_name>/stop")
def fedstellar_stop_scenario(scenario_name):
    if "user" not in session:
        return abort(401)
    stop_scenario(_safe(scenario_name))
    return redirect(url_for("fedstellar_scenario"))


@app.route("/scenario/<scenario_name>/remove")
def fedstellar_remove_scenario(scenario_name):
    if "user" not in session:
        return abort(401)
    _under(app.config["config_dir"], scenario_name), _under(app.config["log_dir"], scenario_name)
    stop_scenario(scenario_name)
    db.remove_nodes_by_scenario_name(scenario_name)
    db.remove_scenario_by_name(scenario_name)
    os.environ["FEDSTELLAR_CONFIG_DIR"], os.environ["FEDSTELLAR_LOGS_DIR"] = _dirs()
    Controller.remove_files_by_scenario(scenario_name)
    return redirect(url_for("fedstellar_scenario"))


# ------------------------------------------------------------------ statistics
def _read_metrics(scenario_name):
    """Per participant ``{metric: [[step, value], ...]}`` from what the nodes' loggers left under
    ``<log_dir>/<scenario>/metrics/participant_<i>/``: the ``metrics.csv`` of the CSV logger and the scalars of
    every TensorBoard event file (the reference's only format, statisticslogger.py:13-60) - the page needs no
    TensorBoard process. Where both carry a metric the event file wins (it is what TensorBoard would show)."""
    from ..learning.pytorch.statisticslogger import read_event_file
    out = {}
    root = os.path.join(app.config["log_dir"], scenario_name, "metrics")
    for pdir in sorted(glob.glob(os.path.join(root, "participant_*"))):
        if not os.path.isdir(pdir):
            continue
        series = {}
        path = os.path.join(pdir, "metrics.csv")
        if os.path.exists(path):
            with open(path) as f:
                next(f, None)
                for line in f:
                    parts = line.rstrip("\n").split(",")
                    if len(parts) == 4:
                        try:
                            series.setdefault(parts[2], []).append([int(parts[1]), float(parts[3])])
                        except ValueError:
                            continue
        events = {}
        for ev in sorted(glob.glob(os.path.join(pdir, "**", "events.out.tfevents.*"), recursive=True)):
            try:
                for tag, pts in read_event_file(ev).items():
                    events.setdefault(tag, []).extend(pts)
            except (OSError, ValueError, IndexError) as e:
                app.logger.debug("skipping %s: %s", ev, e)
        for tag, pts in events.items():
            series[tag] = sorted(pts)
        out[os.path.basename(pdir)] = series
    return out


@app.route("/scenario/statistics/")
@app.route("/scenario/<scenario_name>/statistics/")
def fedstellar_scenario_statistics(scenario_name=None):
    if "user" not in session:
        return abort(401)
    if scenario_name is None:
        running = db.get_running_scenario()
        scenario_name = running[0] if running else None
    metrics = _read_metrics(_safe(scenario_name)) if scenario_name else {}
    tb_url = f"http://{request.host.split(':')[0]}:{app.config['statistics_port']}"
    return render_template("statistics.html", scenario_name=scenario_name, metrics=metrics, tensorboard_url=tb_url)


@app.route("/api/scenario/<scenario_name>/statistics/")
def fedstellar_api_statistics(scenario_name):
    return jsonify(_read_metrics(_safe(scenario_name)))


@app.route("/scenario/<scenario_name>/statistics/download")
def fedstellar_statistics_download(scenario_name):
    if "user" not in session:
        return abort(401)
    root = os.path.join(_under(app.config["log_dir"], scenario_name), "metrics")
    if not os.path.isdir(root):
        return abort(404)
    mem = io.BytesIO()
    with zipfile.ZipFile(mem, "w", zipfile.ZIP_DEFLATED) as z:
        for dirpath, _, files in os.walk(root):
            for fn in files:
                full = os.path.join(dirpath, fn)
                z.write(full, os.path.relpath(full, root))
    mem.seek(0)
    return send_file(mem, mimetype="application/zip", as_attachment=True, download_name=f"{scenario_name}_metrics.zip")


@app.route("/scenario/<scenario_name>/scenario.json")
def fedstellar_scenario_file(scenario_name):
    """The whole scenario as ONE document (config/scenario.py: defaults + per-node deltas), packed from the
    participant files the controller keeps - what the batched launcher takes instead of N files. Tokens and
    credentials never leave: ``controller_token``, device passwords and W&B keys are blanked."""
    if "user" not in session:
        return abort(401)
    from ..config.scenario import pack_scenario, read_participant_files
    parts = read_participant_files(_under(app.config["config_dir"], scenario_name))
    if not parts:
        return abort(404)
    for p in parts:
        p.get("scenario_args", {}).pop("controller_token", None)
        if "password" in p.get("device_args", {}):
            p["device_args"]["password"] = ""
        if "wandb_apikey" in p.get("tracking_args", {}):
            p["tracking_args"]["wandb_apikey"] = ""
    return jsonify(pack_scenario(parts))


# ------------------------------------------------------------------ deployment
@app.route("/scenario/deployment/")
def fedstellar_scenario_deployment():
    if "user" not in session:
        return abort(401)
    return render_template("deployment.html", running=db.get_running_scenario())


def _controller_args(data, scenario_name):
    return argparse.Namespace(
        scenario_name=scenario_name, federation=data["federation"], topology=data["topology"], webserver=False,
        webport=request.host.split(":")[1] if ":" in request.host else 80, statsport=app.config["statistics_port"],
        simulation=bool(data.get("simulation", True)), docker=bool(data.get("docker", False)),
        config=app.config["config_dir"], logs=app.config["log_dir"], env=None, python=app.config["python_path"],
        matrix=data.get("matrix"), engine=data.get("engine", "auto"), gpus=data.get("gpus"),
        encrypt=bool(data.get("encrypt", False)), network_subnet=data.get("network_subnet"),
        network_gateway=data.get("network_gateway"), control_plane=bool(data.get("control_plane", False)),
        gateway=[str(g) for g in (data.get("gateways") or []) if re.fullmatch(r"\d+(:\d+)?", str(g))])


_SHAPING = {"interface": r"[A-Za-z0-9_.:\-]{1,15}", "rate": r"\d+(\.\d+)?[KMG]?bps", "loss": r"\d+(\.\d+)?%",
            "delay": r"\d+(\.\d+)?(us|ms|s)", "delay-distro": r"\d+(\.\d+)?"}


def _shaping_fields(nd: dict) -> dict:
    """The network-shaping keys of one designer node (they end up on a ``tcset`` command line: strict formats)."""
    out = {}
    for key, pattern in _SHAPING.items():
        v = nd.get(key, nd.get(key.replace("-", "_")))
        if v in (None, ""):
            continue
        if not re.fullmatch(pattern, str(v)):
            raise ValueError(f"bad {key} {v!r}")
        out[key] = float(v) if key == "delay-distro" else str(v)
    return out


@app.route("/scenario/deployment/run", methods=["POST"])
def fedstellar_scenario_deployment_run():
    """Body (reference deployment.html:538-555): scenario_title, scenario_description, nodes{i:{id,ip,
    ipdemo,port,role,start}}, n_nodes, matrix, federation, simulation, docker, topology, rounds, dataset,
    model, epochs, accelerator, network_subnet, network_gateway (+ engine, gpus, encrypt)."""
    if "user" not in session:
        return abort(401)
    data = request.get_json(force=True, silent=True)
    if not data:
        return abort(400)
    stop_all_scenarios()
    n = int(data["n_nodes"])
    federation = data.get("federation", "DFL")
    if federation == "CFL":
        data["topology"] = "star"
        data["matrix"] = None
    scenario_name = f'fedstellar_{federation}_{datetime.datetime.now().strftime("%d_%m_%Y_%H_%M_%S")}'
    cdir = os.path.join(app.config["config_dir"], scenario_name)
    os.makedirs(cdir, exist_ok=True)
    os.makedirs(os.path.join(app.config["log_dir"], scenario_name), exist_ok=True)
    with open(os.path.join(cdir, "controller.json"), "w") as f:
        json.dump(data, f, indent=2)
    tpl_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "participant.json.example")
    nodes = data.get("nodes") or {}
    roles = default_roles(federation, n)
    have_start = any(bool(v.get("start")) for v in nodes.values())
    for i in range(n):
        with open(tpl_path) as f:
            p = json.load(f)
        nd = nodes.get(str(i), {})
        p["scenario_args"].update(name=scenario_name, federation=federation, rounds=int(data.get("rounds", 10)),
                                  n_nodes=n, controller=request.host, simulation=bool(data.get("simulation", True)))
        p["device_args"].update(idx=i, role=nd.get("role", roles[i]),
                                start=bool(nd.get("start")) if have_start else i == 0,
                                accelerator=data.get("accelerator", "auto"))
        p["network_args"].update(ip=nd.get("ip", "127.0.0.1"), ipdemo=nd.get("ipdemo", ""), port=int(nd.get("port", 45000 + i)))
        # WAN emulation of a physical deployment (tcset in BaseNode, reference base_node.py:82-85) and the map position
        try:
            p["network_args"].update(_shaping_fields(nd))
            if nd.get("latitude") not in (None, "") and nd.get("longitude") not in (None, ""):
                lat, lon = float(nd["latitude"]), float(nd["longitude"])
                if not (-90 <= lat <= 90 and -180 <= lon <= 180):
                    raise ValueError("coordinates out of range")
                p["geo_args"].update(latitude=str(lat), longitude=str(lon), fixed=True)
        except ValueError as e:
            return jsonify({"error": f"node {i}: {e}"}), 400
        p["data_args"]["dataset"] = data.get("dataset", "MNIST")
        p["model_args"]["model"] = data.get("model", "MLP")
        p["training_args"]["epochs"] = int(data.get("epochs", 3))
        with open(os.path.join(cdir, f"participant_{i}.json"), "w") as f:
            json.dump(p, f, indent=2)
    controller = Controller(_controller_args(data, scenario_name))
    try:
        controller.load_configurations_and_start_nodes()
    except Exception as e:
        return jsonify({"error": str(e)}), 400
    _RUNNING[scenario_name] = controller
    db.scenario_update_record(scenario_name, controller.start_date_scenario, "", data.get("scenario_title", ""),
                              data.get("scenario_description", ""), "running", data.get("network_subnet", ""))
    return jsonify({"scenario_name": scenario_name, "redirect": url_for("fedstellar_scenario")})


@app.route("/scenario/<scenario_name>/deployment/reload")
def fedstellar_scenario_deployment_reload(scenario_name):
    if "user" not in session:
        return abort(401)
    _safe(scenario_name)
    path = os.path.join(app.config["config_dir"], scenario_name, "controller.json")
    if not os.path.exists(path):
        return abort(404)
    with open(path) as f:
        data = json.load(f)
    stop_all_scenarios()
    controller = Controller(_controller_args(data, scenario_name))
    controller.load_configurations_and_start_nodes()
    _RUNNING[scenario_name] = controller
    old = db.get_scenario_by_name(scenario_name)
    db.scenario_update_record(scenario_name, controller.start_date_scenario, "", old[3] if old else "",
                              old[4] if old else "", "running", old[6] if old else "")
    return redirect(url_for("fedstellar_scenario"))


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--port", type=int, default=5000)
    parser.add_argument("--host", default="0.0.0.0")
    a = parser.parse_args()
    db.init_db()
    app.run(debug=False, host=a.host, port=a.port, threaded=True)
