"""graphvite_amd — MI355X-native node-embedding training (LINE / DeepWalk / node2vec), a drop-in for the
GraphSolver / GraphApplication.train path of GraphVite.  See DESIGN.md and INTEGRATION.md."""

__version__ = "0.1.0"
