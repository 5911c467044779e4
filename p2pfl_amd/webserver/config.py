"""Flask constants (reference: fedstellar/webserver/config.py:1-6)."""
import os

SECRET_KEY = os.environ.get("FEDSTELLAR_SECRET_KEY") or os.urandom(24).hex()
MAX_CONTENT_LENGTH = 64 * 1024 * 1024
UPLOAD_FOLDER = "image_pool"
ALLOWED_EXTENSIONS = {"png", "jpg", "jpeg", "gif"}
NODE_ONLINE_SECONDS = 20      # a node is "online" if it reported within this window (app.py:307-311)
