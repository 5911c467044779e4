"""Node roles (reference: fedstellar/role.py:12-36)."""


class Role:
    """String constants; the wire protocol and the participant JSON carry these literally."""
    TRAINER = "trainer"
    AGGREGATOR = "aggregator"
    PROXY = "proxy"
    IDLE = "idle"
    SERVER = "server"

    ALL = (TRAINER, AGGREGATOR, PROXY, IDLE, SERVER)

    @staticmethod
    def trains(role: str) -> bool:
        """Roles whose round includes local training (node.py:427-490: SERVER never trains)."""
        return role in (Role.TRAINER, Role.AGGREGATOR)

    @staticmethod
    def aggregates(role: str) -> bool:
        return role in (Role.AGGREGATOR, Role.SERVER)

    @staticmethod
    def waits(role: str) -> bool:
        """Roles that adopt the first (aggregated) model they receive (aggregator.py:119-123)."""
        return role in (Role.TRAINER, Role.PROXY, Role.IDLE)


def default_roles(federation: str, n_nodes: int):
    """Role assignment used by the deployment UI (webserver/templates/deployment.html:608-617)."""
    fed = federation.upper()
    if fed == "DFL":
        return [Role.AGGREGATOR] * n_nodes
    if fed == "SDFL":
        return [Role.AGGREGATOR] + [Role.TRAINER] * (n_nodes - 1)
    if fed == "CFL":
        return [Role.SERVER] + [Role.TRAINER] * (n_nodes - 1)
    raise ValueError(f"unknown federation {federation!r}")
