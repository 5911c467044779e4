"""sqlite data-access layer of the frontend (reference: fedstellar/webserver/database.py:1-398).

Same tables and row shapes — ``users(user, password, role)``, ``notes``, ``nodes(uid, idx, ip, port,
role, neighbors, latitude, longitude, timestamp, federation, scenario)``, ``scenarios(name,
start_time, end_time, title, description, status, network_subnet)`` — in one database file, every
statement parameterised (the reference concatenates SQL strings, database.py:34,48,260,270,330).
Passwords are salted PBKDF2 (the reference stores unsalted sha256, database.py:44-53).
"""
from __future__ import annotations

import datetime
import hashlib
import os
import secrets
import sqlite3
import threading

_DB_DIR = os.environ.get("FEDSTELLAR_DB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "database_file")
_lock = threading.Lock()


def _path():
    d = os.environ.get("FEDSTELLAR_DB_DIR") or _DB_DIR
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "fedstellar.db")


def _conn():
    c = sqlite3.connect(_path(), timeout=10)
    c.execute("PRAGMA journal_mode=WAL")
    return c


def init_db():
    with _lock, _conn() as c:
        c.execute("CREATE TABLE IF NOT EXISTS users (user TEXT PRIMARY KEY, password TEXT NOT NULL, role TEXT NOT NULL)")
        c.execute("CREATE TABLE IF NOT EXISTS notes (id TEXT PRIMARY KEY, user TEXT, timestamp TEXT, note TEXT)")
        c.execute("CREATE TABLE IF NOT EXISTS images (uid TEXT PRIMARY KEY, owner TEXT, name TEXT, timestamp TEXT)")
        c.execute("CREATE TABLE IF NOT EXISTS nodes (uid TEXT PRIMARY KEY, idx TEXT, ip TEXT, port TEXT, role TEXT, "
                  "neighbors TEXT, latitude TEXT, longitude TEXT, timestamp TEXT, federation TEXT, scenario TEXT)")
        c.execute("CREATE TABLE IF NOT EXISTS scenarios (name TEXT PRIMARY KEY, start_time TEXT, end_time TEXT, "
                  "title TEXT, description TEXT, status TEXT, network_subnet TEXT)")
        if c.execute("SELECT COUNT(*) FROM users").fetchone()[0] == 0:
            pwd = os.environ.get("FEDSTELLAR_ADMIN_PASSWORD", "admin")
            c.execute("INSERT INTO users VALUES (?, ?, ?)", ("ADMIN", _hash(pwd), "admin"))


# ---------------------------------------------------------------- users
def _hash(password: str, salt: str = None) -> str:
    salt = salt or secrets.token_hex(8)
    return salt + "$" + hashlib.pbkdf2_hmac("sha256", password.encode(), salt.encode(), 100_000).hex()


def list_users(all_info=False):
    with _conn() as c:
        rows = c.execute("SELECT user, password, role FROM users").fetchall()
    return rows if all_info else [r[0] for r in rows]


def get_user_info(user):
    with _conn() as c:
        return c.execute("SELECT user, password, role FROM users WHERE user = ?", (user.upper(),)).fetchone()


def verify(user, password) -> bool:
    row = get_user_info(user)
    if row is None:
        return False
    salt = row[1].split("$", 1)[0]
    return secrets.compare_digest(row[1], _hash(password, salt))


def add_user(user, password, role):
    with _lock, _conn() as c:
        c.execute("INSERT INTO users VALUES (?, ?, ?)", (user.upper(), _hash(password), role))


def delete_user_from_db(user):
    with _lock, _conn() as c:
        c.execute("DELETE FROM users WHERE user = ?", (user.upper(),))
        c.execute("DELETE FROM notes WHERE user = ?", (user.upper(),))
        c.execute("DELETE FROM images WHERE owner = ?", (user.upper(),))


# ---------------------------------------------------------------- notes
def read_note_from_db(user):
    with _conn() as c:
        return c.execute("SELECT id, timestamp, note FROM notes WHERE user = ?", (user.upper(),)).fetchall()


def match_user_id_with_note_id(note_id):
    with _conn() as c:
        row = c.execute("SELECT user FROM notes WHERE id = ?", (note_id,)).fetchone()
    return row[0] if row else None


def write_note_into_db(user, note):
    ts = str(datetime.datetime.now())
    nid = hashlib.sha1((user.upper() + ts).encode()).hexdigest()
    with _lock, _conn() as c:
        c.execute("INSERT INTO notes VALUES (?, ?, ?, ?)", (nid, user.upper(), ts, note))
    return nid


def delete_note_from_db(note_id):
    with _lock, _conn() as c:
        c.execute("DELETE FROM notes WHERE id = ?", (note_id,))


# ---------------------------------------------------------------- images (reference database.py:176-236)
def image_upload_record(uid, owner, image_name, timestamp):
    with _lock, _conn() as c:
        c.execute("INSERT INTO images VALUES (?, ?, ?, ?)", (uid, owner.upper(), image_name, timestamp))


def list_images_for_user(owner):
    with _conn() as c:
        return c.execute("SELECT uid, timestamp, name FROM images WHERE owner = ?", (owner.upper(),)).fetchall()


def match_user_id_with_image_uid(image_uid):
    with _conn() as c:
        row = c.execute("SELECT owner FROM images WHERE uid = ?", (image_uid,)).fetchone()
    return row[0] if row else None


def get_image_file_name(image_uid):
    with _conn() as c:
        row = c.execute("SELECT name FROM images WHERE uid = ?", (image_uid,)).fetchone()
    return None if row is None else f"{image_uid}-{row[0]}"


def delete_image_from_db(image_uid):
    with _lock, _conn() as c:
        c.execute("DELETE FROM images WHERE uid = ?", (image_uid,))


# ---------------------------------------------------------------- nodes
def list_nodes(scenario_name=None, sort_by="idx"):
    col = sort_by if sort_by in ("idx", "uid", "ip", "port", "role", "timestamp") else "idx"
    with _conn() as c:
        if scenario_name:
            rows = c.execute(f"SELECT * FROM nodes WHERE scenario = ? ORDER BY CAST({col} AS INTEGER)", (scenario_name,))
        else:
            rows = c.execute(f"SELECT * FROM nodes ORDER BY CAST({col} AS INTEGER)")
        return rows.fetchall()


def list_nodes_by_scenario_name(scenario_name):
    return list_nodes(scenario_name)


def update_node_record(node_uid, idx, ip, port, role, neighbors, latitude, longitude, timestamp, federation, scenario):
    with _lock, _conn() as c:
        c.execute("INSERT OR REPLACE INTO nodes VALUES (?, ?, ?, ?, ?, ?, ?, ?, ?, ?, ?)",
                  (str(node_uid), str(idx), str(ip), str(port), str(role), str(neighbors), str(latitude), str(longitude),
                   str(timestamp), str(federation), str(scenario)))


def remove_all_nodes():
    with _lock, _conn() as c:
        c.execute("DELETE FROM nodes")


def remove_nodes_by_scenario_name(scenario_name):
    with _lock, _conn() as c:
        c.execute("DELETE FROM nodes WHERE scenario = ?", (scenario_name,))


# ---------------------------------------------------------------- scenarios
def get_all_scenarios(sort_by="start_time"):
    col = sort_by if sort_by in ("name", "start_time", "end_time", "status") else "start_time"
    with _conn() as c:
        return c.execute(f"SELECT * FROM scenarios ORDER BY {col} DESC").fetchall()


def scenario_update_record(scenario_name, start_time, end_time, title, description, status, network_subnet):
    with _lock, _conn() as c:
        c.execute("INSERT OR REPLACE INTO scenarios VALUES (?, ?, ?, ?, ?, ?, ?)",
                  (scenario_name, start_time, end_time, title, description, status, network_subnet))


def scenario_set_all_status_to_finished():
    with _lock, _conn() as c:
        c.execute("UPDATE scenarios SET status = 'finished', end_time = ? WHERE status = 'running'",
                  (datetime.datetime.now().strftime("%d/%m/%Y %H:%M:%S"),))


def scenario_set_status_to_finished(scenario_name):
    with _lock, _conn() as c:
        c.execute("UPDATE scenarios SET status = 'finished', end_time = ? WHERE name = ?",
                  (datetime.datetime.now().strftime("%d/%m/%Y %H:%M:%S"), scenario_name))


def get_running_scenario():
    with _conn() as c:
        return c.execute("SELECT * FROM scenarios WHERE status = 'running'").fetchone()


def get_scenario_by_name(scenario_name):
    with _conn() as c:
        return c.execute("SELECT * FROM scenarios WHERE name = ?", (scenario_name,)).fetchone()


def remove_scenario_by_name(scenario_name):
    with _lock, _conn() as c:
        c.execute("DELETE FROM scenarios WHERE name = ?", (scenario_name,))
