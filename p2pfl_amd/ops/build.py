"""In-tree native builds.

* ``libp2pfl_hip.so``  – every gfx950 kernel under ``p2pfl_amd/csrc/*.hip`` (hipcc,
  ``--offload-arch=gfx950``; cross-compiles on a CPU-only box).
* ``libp2pfl_host.so`` – host-side C++ runtime pieces under ``p2pfl_amd/csrc/host/*.cpp``
  (AES-128 link cipher, PARAMS fragmenter / reassembler, message-id ring) built with g++.

Both land in ``p2pfl_amd/ops/_lib/`` so they travel with the source tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "csrc"
LIB_DIR = Path(__file__).resolve().parent / "_lib"
HIP_LIB = LIB_DIR / "libp2pfl_hip.so"
HOST_LIB = LIB_DIR / "libp2pfl_host.so"
GPU_ARCH = "gfx950"


def _digest(files) -> str:
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def _needs_build(lib: Path, files) -> bool:
    stamp = lib.with_suffix(".sha256")
    return not (lib.exists() and stamp.exists() and stamp.read_text().strip() == _digest(files))


def _write_stamp(lib: Path, files) -> None:
    lib.with_suffix(".sha256").write_text(_digest(files))


def hipcc_path() -> str | None:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    return None


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile all HIP kernels for gfx950 into one shared library."""
    srcs = sorted(CSRC.glob("*.hip"))
    deps = srcs + sorted(CSRC.glob("*.h"))
    if not force and not _needs_build(HIP_LIB, deps):
        return HIP_LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    objs = []
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    procs = []
    for s in srcs:
        o = obj_dir / (s.stem + ".o")
        cmd = [hipcc, f"--offload-arch={GPU_ARCH}", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
               "-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, f"--offload-arch={GPU_ARCH}", "-shared", "-fPIC", "-o", str(HIP_LIB)] + [str(o) for o in objs]
    subprocess.run(cmd, check=True)
    shutil.rmtree(obj_dir, ignore_errors=True)
    _write_stamp(HIP_LIB, deps)
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """Compile the host C++ runtime (no GPU code) with g++."""
    srcs = sorted((CSRC / "host").glob("*.cpp"))
    deps = srcs + sorted((CSRC / "host").glob("*.h"))
    if not force and not _needs_build(HOST_LIB, deps):
        return HOST_LIB
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("no C++ compiler found for the host runtime")
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-maes", "-msse4.1", "-pthread",
           "-fvisibility=hidden", "-o", str(HOST_LIB)] + [str(s) for s in srcs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    _write_stamp(HOST_LIB, deps)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose=True))
