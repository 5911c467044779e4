"""One file for a whole scenario.

The reference's configuration surface is one ``participant_<idx>.json`` per node (controller.py:277-295,
webserver/app.py:640-673): a 1000-client federation is 1000 files that differ in a handful of fields. The client-batched
runtime hosts all of them in one process per GPU, so it also reads (and the controller also writes) ``scenario.json``::

    {"version": 1, "n_nodes": N,
     "defaults": { ...one full participant document: what all nodes share... },
     "nodes": [ { ...only what node i changes: device_args.idx / role / start, network_args.port / neighbors, ... }, ... ]}

``expand_scenario`` gives back exactly the per-participant documents (so either form can be produced from the other:
``emit_participants`` / ``write_scenario``); the per-participant files stay the wire format towards per-process nodes
and the frontend.
"""
from __future__ import annotations

import copy
import glob
import json
import os
from typing import Dict, List, Optional

SCENARIO_FILE = "scenario.json"
_MISSING = object()


def _diff(base, doc):
    """What ``doc`` changes against ``base`` (dicts recursively; anything else by value). ``_MISSING``: nothing."""
    if isinstance(base, dict) and isinstance(doc, dict):
        out = {}
        for k, v in doc.items():
            d = _diff(base[k], v) if k in base else copy.deepcopy(v)
            if d is not _MISSING:
                out[k] = d
        gone = [k for k in base if k not in doc]
        if gone:
            out["__removed__"] = gone
        return out if out else _MISSING
    return _MISSING if base == doc else copy.deepcopy(doc)


def _apply(base: dict, delta: dict) -> dict:
    out = copy.deepcopy(base)
    for k, v in delta.items():
        if k == "__removed__":
            for g in v:
                out.pop(g, None)
        elif isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _apply(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def pack_scenario(participants: List[dict]) -> dict:
    """Participant documents (index order) -> scenario document. ``defaults`` is node 0's document; every node lists
    only what it changes."""
    if not participants:
        raise ValueError("a scenario needs at least one participant")
    defaults = copy.deepcopy(participants[0])
    nodes = []
    for p in participants:
        d = _diff(defaults, p)
        nodes.append({} if d is _MISSING else d)
    return {"version": 1, "n_nodes": len(participants), "defaults": defaults, "nodes": nodes}


def expand_scenario(doc: dict) -> List[dict]:
    """Scenario document -> participant documents. ``nodes`` may be shorter than ``n_nodes`` (missing entries = the
    defaults with ``device_args.idx`` and the port advanced) so hand-written scenarios stay short."""
    if int(doc.get("version", 1)) != 1:
        raise ValueError(f"unknown scenario.json version {doc.get('version')!r}")
    defaults = doc["defaults"]
    nodes = list(doc.get("nodes", []))
    n = int(doc.get("n_nodes", len(nodes)))
    if n < 1 or len(nodes) > n:
        raise ValueError("scenario.json: n_nodes must be >= 1 and cover every entry of 'nodes'")
    out = []
    base_port = int(defaults.get("network_args", {}).get("port", 45000))
    for i in range(n):
        if i < len(nodes):
            p = _apply(defaults, nodes[i])
        else:
            p = copy.deepcopy(defaults)
            p.setdefault("device_args", {}).update(idx=i, start=False)
            p.setdefault("network_args", {})["port"] = base_port + i
        out.append(p)
    return out


def read_participant_files(config_dir: str) -> List[dict]:
    files = sorted(glob.glob(os.path.join(config_dir, "participant_*.json")),
                   key=lambda p: int(os.path.basename(p).split("_")[1].split(".")[0]))
    out = []
    for f in files:
        with open(f) as fh:
            out.append(json.load(fh))
    return out


def write_scenario(config_dir: str, participants: Optional[List[dict]] = None) -> str:
    """Write ``scenario.json`` for the participants given (default: the participant files of ``config_dir``)."""
    parts = participants if participants is not None else read_participant_files(config_dir)
    path = os.path.join(config_dir, SCENARIO_FILE)
    with open(path, "w") as f:
        json.dump(pack_scenario(parts), f, indent=1)
    return path


def read_scenario(path: str) -> List[dict]:
    if os.path.isdir(path):
        path = os.path.join(path, SCENARIO_FILE)
    with open(path) as f:
        return expand_scenario(json.load(f))


def emit_participants(scenario_path: str, out_dir: Optional[str] = None) -> List[str]:
    """``scenario.json`` -> ``participant_<i>.json`` files (for per-process nodes, the frontend, the reference's tools)."""
    parts = read_scenario(scenario_path)
    out_dir = out_dir or (scenario_path if os.path.isdir(scenario_path) else os.path.dirname(scenario_path))
    os.makedirs(out_dir, exist_ok=True)
    files = []
    for i, p in enumerate(parts):
        path = os.path.join(out_dir, f"participant_{i}.json")
        with open(path, "w") as f:
            json.dump(p, f, indent=2)
        files.append(path)
    return files


def main(argv=None) -> int:
    """``python -m p2pfl_amd.config.scenario pack|expand <config dir>``"""
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("what", choices=["pack", "expand"])
    ap.add_argument("config_dir")
    a = ap.parse_args(argv)
    if a.what == "pack":
        print(write_scenario(a.config_dir))
    else:
        print("\n".join(emit_participants(a.config_dir)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
