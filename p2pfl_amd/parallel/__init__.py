"""Multi-GPU layer: topology -> collective planning (planner) and the RCCL / gloo exchange primitives
(collectives). One process per GPU; ``torch.distributed`` backend "nccl" is RCCL over xGMI."""
from .planner import (ExchangePlan, estimate_exchange_seconds, link_key, owner_of, partition_nodes,  # noqa: F401
                      plan_exchange)
from .collectives import DeviceExchange  # noqa: F401,E402
