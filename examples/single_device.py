"""Two reference-style nodes in one process: ``python examples/single_device.py [--rounds 2] [--epochs 1]``.
The implementation lives in the package (``p2pfl_amd/single_device.py``, reference path fedstellar/single_device.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from p2pfl_amd.single_device import main  # noqa: E402

if __name__ == "__main__":
    main()
