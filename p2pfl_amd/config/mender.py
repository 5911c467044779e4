"""Mender OTA REST client (reference: fedstellar/config/mender.py:10-221).

Used only for deployments on physical devices (``--simulation`` off); irrelevant on a single MI355X
box but kept for capability parity: login, artifact listing/upload, deployments, device inventory.
Credentials come from the environment (``MENDER_SERVER``, ``MENDER_USER``, ``MENDER_PASSWORD`` or a
ready ``MENDER_TOKEN``), never from the repository.
"""
from __future__ import annotations

import json
import logging
import os


class Mender:
    def __init__(self, server=None, user=None, password=None, token=None, timeout=10):
        self.server = (server or os.environ.get("MENDER_SERVER", "")).rstrip("/")
        self.user = user or os.environ.get("MENDER_USER")
        self.password = password or os.environ.get("MENDER_PASSWORD")
        self.token = token or os.environ.get("MENDER_TOKEN")
        self.timeout = timeout

    # ------------------------------------------------------------------ plumbing
    def _session(self):
        import requests
        return requests

    def _headers(self):
        return {"Accept": "application/json", "Authorization": f"Bearer {self.get_token()}",
                "Content-Type": "application/json"}

    def _url(self, path):
        if not self.server:
            raise RuntimeError("MENDER_SERVER is not configured")
        return f"{self.server}{path}"

    def get_token(self):
        if self.token:
            return self.token
        r = self._session().post(self._url("/api/management/v1/useradm/auth/login"), auth=(self.user, self.password),
                                 timeout=self.timeout)
        r.raise_for_status()
        self.token = r.text
        return self.token

    def renew_token(self):
        """Log in again whatever token is held (reference mender.py:22-32)."""
        self.token = None
        return self.get_token()

    @staticmethod
    def generate_artifact(type_artifact, artifact_name, device_type, file_path):
        """Write ``<artifact_name>.mender`` with the ``mender-artifact`` tool (reference mender.py:34-36; the tool is
        not part of this image: a missing executable is reported, not ignored)."""
        import shutil
        import subprocess
        tool = shutil.which("mender-artifact")
        if tool is None:
            raise FileNotFoundError("mender-artifact is not installed")
        out = f"{artifact_name}.mender"
        subprocess.run([tool, "write", "module-image", "-T", str(type_artifact), "-n", str(artifact_name),
                        "-t", str(device_type), "-o", out, "-f", str(file_path)], check=True)
        return out

    def get_my_info(self):
        r = self._session().get(self._url("/api/management/v1/useradm/users/me"), headers=self._headers(),
                                timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    def get_info_deployment(self, deployment_id):
        r = self._session().get(self._url(f"/api/management/v1/deployments/deployments/{deployment_id}"),
                                headers=self._headers(), timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    # ------------------------------------------------------------------ API
    def get_artifacts(self):
        r = self._session().get(self._url("/api/management/v1/deployments/artifacts"), headers=self._headers(),
                                timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    def upload_artifact(self, artifact_path, description=""):
        with open(artifact_path, "rb") as f:
            r = self._session().post(self._url("/api/management/v1/deployments/artifacts"),
                                     headers={"Authorization": f"Bearer {self.get_token()}"},
                                     files={"artifact": f}, data={"description": description}, timeout=600)
        r.raise_for_status()
        return r.status_code

    def get_devices(self):
        try:
            r = self._session().get(self._url("/api/management/v2/devauth/devices"), headers=self._headers(),
                                    params={"per_page": 300}, timeout=self.timeout)
            r.raise_for_status()
            return r.json()
        except Exception as e:
            logging.info("[Mender] cannot list devices: %s", e)
            return []

    def get_devices_by_group(self, group):
        r = self._session().get(self._url(f"/api/management/v1/inventory/groups/{group}/devices"),
                                headers=self._headers(), params={"per_page": 300}, timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    def get_info_device(self, device_id):
        r = self._session().get(self._url(f"/api/management/v1/inventory/devices/{device_id}"), headers=self._headers(),
                                timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    def get_connected_device(self, device_id):
        r = self._session().get(self._url(f"/api/management/v1/deviceconnect/devices/{device_id}"),
                                headers=self._headers(), timeout=self.timeout)
        r.raise_for_status()
        return r.json()

    def deploy_artifact_device(self, artifact_name, device_id, name=None):
        body = {"artifact_name": artifact_name, "devices": [device_id], "name": name or f"{artifact_name}-{device_id}"}
        r = self._session().post(self._url("/api/management/v1/deployments/deployments"), headers=self._headers(),
                                 data=json.dumps(body), timeout=self.timeout)
        r.raise_for_status()
        return r.status_code

    def deploy_artifact_list(self, artifact_name, devices, name=None):
        body = {"artifact_name": artifact_name, "devices": list(devices), "name": name or artifact_name}
        r = self._session().post(self._url("/api/management/v1/deployments/deployments"), headers=self._headers(),
                                 data=json.dumps(body), timeout=self.timeout)
        r.raise_for_status()
        return r.status_code
