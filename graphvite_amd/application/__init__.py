"""graphvite_amd.application — drop-in for graphvite.application (python/graphvite/application/)."""
from .application import Application, GraphApplication

__all__ = ["Application", "GraphApplication"]
