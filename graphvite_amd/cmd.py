"""`python -m graphvite_amd.cmd run config.yaml` — the `graphvite run` entry point of the reference
(python/graphvite/cmd.py:82-163) for the node-embedding application: a YAML file with the sections
application / resource / format / graph / build / train / evaluate / save / load is turned into the same
GraphApplication calls.  "auto" values become `auto` (0), as cmd.py:98 does.  Dataset placeholders
(<blogcatalog.train>) need the reference's downloader and are not resolved here: give real file names."""
import argparse
import logging
import sys

import yaml

from . import application, optimizer
from .base import auto, init_logging


def load_config(config_file):
    with open(config_file, "r") as fin:
        cfg = yaml.safe_load(fin)

    def convert(node):
        if isinstance(node, dict):
            return {k: convert(v) for k, v in node.items()}
        if isinstance(node, list):
            return [convert(v) for v in node]
        if node == "auto":
            return auto
        if isinstance(node, str) and node.startswith("<") and node.endswith(">"):
            raise ValueError("dataset placeholder %s cannot be resolved offline; use a file name" % node)
        return node

    return convert(cfg)


def run_main(args):
    cfg = load_config(args.config)
    init_logging(logging.INFO)
    kind = cfg.get("application", "graph")
    if kind not in application.Application.application:
        raise ValueError("only the `graph` and `word graph` (node embedding) applications are implemented, not `%s`" % kind)
    resource = dict(cfg.get("resource", {}))
    if args.gpu is not None:
        resource["gpus"] = [args.gpu]
    if args.cpu is not None:
        resource["cpu_per_gpu"] = args.cpu
    app = application.Application(kind, **resource)
    if "format" in cfg:
        app.set_format(**cfg["format"])
    app.load(**cfg["graph"])
    build = dict(cfg.get("build", {}))
    if isinstance(build.get("optimizer"), dict):
        build["optimizer"] = optimizer.Optimizer(**build["optimizer"])
    app.build(**build)
    if "load" in cfg:
        app.load_model(**cfg["load"])
    app.train(**cfg.get("train", {}))
    if args.eval and "evaluate" in cfg:
        evaluate = cfg["evaluate"]
        for task in (evaluate if isinstance(evaluate, list) else [evaluate]):
            app.evaluate(**task)
    if "save" in cfg:
        app.save_model(**cfg["save"])
    return app


def main(argv=None):
    parser = argparse.ArgumentParser(prog="graphvite_amd", description="GraphVite node embedding on MI355X")
    sub = parser.add_subparsers(dest="command")
    run = sub.add_parser("run", help="run from a configuration file")
    run.add_argument("config", help="yaml configuration file")
    run.add_argument("--no-eval", dest="eval", action="store_false", help="turn off evaluation")
    run.add_argument("--gpu", type=int, default=None, help="override the gpu id")
    run.add_argument("--cpu", type=int, default=None, help="override the number of CPU threads per GPU")
    args = parser.parse_args(argv)
    if args.command != "run":
        parser.print_help()
        return 1
    run_main(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
