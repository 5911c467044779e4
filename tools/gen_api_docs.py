#!/usr/bin/env python
"""Write ``docs/api/<module>.rst`` (one ``automodule`` page per module of the package, like the reference's checked-in
``docs/fedstellar.*.rst`` pages) and ``docs/api/index.rst``. ``docs/conf.py`` prefers sphinx-autoapi when it is
installed; these pages are what plain ``sphinx.ext.autodoc`` builds from. ``--check`` fails when a page is missing or
stale (tests/test_public_api.py runs it)."""
import argparse
import os
import pkgutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "docs", "api")


def modules():
    import p2pfl_amd
    names = ["p2pfl_amd"]
    for m in pkgutil.walk_packages(p2pfl_amd.__path__, "p2pfl_amd."):
        if "._lib" in m.name or ".csrc" in m.name:
            continue
        names.append(m.name)
    return sorted(names)


def page(name: str) -> str:
    title = name.replace("_", r"\_")
    return (f"{title}\n{'=' * len(title)}\n\n.. automodule:: {name}\n   :members:\n   :undoc-members:\n"
            f"   :show-inheritance:\n")


def index(names) -> str:
    head = "API reference\n=============\n\nOne page per module (``tools/gen_api_docs.py``).\n\n.. toctree::\n   :maxdepth: 1\n\n"
    return head + "".join(f"   {n}\n" for n in names)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args(argv)
    names = modules()
    want = {f"{n}.rst": page(n) for n in names}
    want["index.rst"] = index(names)
    if a.check:
        bad = []
        for fn, text in want.items():
            path = os.path.join(OUT, fn)
            if not os.path.exists(path):
                bad.append(f"missing {fn}")
                continue
            with open(path) as f:
                if f.read() != text:
                    bad.append(f"stale {fn}")
        have = set(os.listdir(OUT)) if os.path.isdir(OUT) else set()
        bad += [f"orphan {fn}" for fn in sorted(have - set(want))]
        if bad:
            print("\n".join(bad))
            return 1
        return 0
    os.makedirs(OUT, exist_ok=True)
    for fn in set(os.listdir(OUT)) - set(want):
        os.remove(os.path.join(OUT, fn))
    for fn, text in want.items():
        with open(os.path.join(OUT, fn), "w") as f:
            f.write(text)
    print(f"{len(names)} module pages -> {OUT}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
