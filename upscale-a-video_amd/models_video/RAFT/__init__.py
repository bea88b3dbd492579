from .raft_bi import RAFT_bi  # noqa: F401
