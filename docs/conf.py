# Sphinx configuration (reference: docs/conf.py:22-52 - autoapi over the package, furo theme).
import os
import sys

sys.path.insert(0, os.path.abspath(".."))

project = "p2pfl-amd"
author = "p2pfl-amd developers"
release = "0.1"

extensions = ["sphinx.ext.napoleon", "sphinx.ext.viewcode", "sphinx.ext.todo"]
try:
    import autoapi  # noqa: F401
    extensions.append("autoapi.extension")
    autoapi_type = "python"
    autoapi_dirs = ["../p2pfl_amd"]
    autoapi_ignore = ["*/_lib/*", "*/csrc/*"]
    autoapi_keep_files = False
except ImportError:
    extensions.append("sphinx.ext.autodoc")

templates_path = ["_templates"]
exclude_patterns = ["_build"]
try:
    import furo  # noqa: F401
    html_theme = "furo"
except ImportError:
    html_theme = "alabaster"
html_title = "p2pfl-amd"
todo_include_todos = True
