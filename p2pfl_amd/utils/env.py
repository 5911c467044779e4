"""Environment / device report (reference: fedstellar/utils/env.py:7-64; the reference reads
``torch.cuda`` names and, elsewhere, pynvml — on MI355X the device facts come from
``torch.cuda`` (HIP) and amdgpu sysfs)."""
from __future__ import annotations

import glob
import logging
import os
import platform
import sys


def gpu_stats():
    """Per-GPU dicts: busy_percent (sysfs), temp_c (hwmon), mem_percent (torch.cuda.mem_get_info)."""
    out = []
    try:
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        n = 0
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/gpu_busy_percent"))
    for i in range(n):
        g = {}
        try:
            import torch
            free, total = torch.cuda.mem_get_info(i)
            g["mem_percent"] = 100.0 * (total - free) / max(total, 1)
        except Exception:
            pass
        if i < len(cards):
            try:
                with open(cards[i]) as f:
                    g["busy_percent"] = float(f.read().strip())
                hw = glob.glob(os.path.join(os.path.dirname(cards[i]), "hwmon/hwmon*/temp1_input"))
                if hw:
                    with open(hw[0]) as f:
                        g["temp_c"] = float(f.read().strip()) / 1000.0
            except Exception:
                pass
        out.append(g)
    return out


def collect_env():
    """Log framework / OS / python / torch / CPU / GPU facts; returns them as a dict."""
    from .. import __version__
    info = {"p2pfl_amd": __version__, "os": platform.system(), "os_release": platform.release(),
            "python": sys.version.split()[0]}
    try:
        import torch
        info["torch"] = torch.__version__
        info["hip"] = getattr(torch.version, "hip", None)
        info["gpus"] = ([torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]
                        if torch.cuda.is_available() else [])
    except Exception:
        info["torch"] = None
    try:
        import psutil
        info["cpu_count"] = os.cpu_count()
        info["cpu_percent"] = psutil.cpu_percent()
        info["ram_percent"] = psutil.virtual_memory().percent
    except Exception:
        pass
    for k, v in info.items():
        logging.info("======== %s: %s", k, v)
    return info
