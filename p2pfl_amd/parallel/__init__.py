"""Multi-GPU layer: topology -> collective planning (planner) and the RCCL / gloo exchange primitives
(collectives). One process per GPU; ``torch.distributed`` backend "nccl" is RCCL over xGMI."""
from .planner import (ExchangePlan, cut_links, estimate_exchange_seconds, link_key, locality_placement,  # noqa: F401
                      owner_of, partition_nodes, placement_order, plan_exchange)
from .collectives import DeviceExchange  # noqa: F401,E402
