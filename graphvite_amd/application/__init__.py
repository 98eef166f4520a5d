"""graphvite_amd.application — drop-in for graphvite.application (python/graphvite/application/)."""
from .application import Application, GraphApplication, WordGraphApplication

__all__ = ["Application", "GraphApplication", "WordGraphApplication"]
