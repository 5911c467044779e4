"""p2pfl_amd — MI355X-native decentralized federated learning runtime.

Same capabilities and public surface as Fedstellar/p2pfl (reference:
``fedstellar/__init__.py:1``), rebuilt around one process per GPU, ``[clients, P]``
parameter slabs in HBM, hand-written gfx950 HIP kernels and RCCL collectives.
"""
__version__ = "0.1.0"
