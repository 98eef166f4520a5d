"""`python -m graphvite_amd.cmd run config.yaml | baseline keyword ... | list` — the `graphvite run`, `graphvite baseline` and
`graphvite list` entry points of the reference (python/graphvite/cmd.py:82-163, 193-260) for the node-embedding
applications: a YAML file with the sections application / resource / format / graph / build / train / evaluate / save / load
is turned into the same GraphApplication calls.  "auto" values become `auto` (0), as cmd.py:98 does.  A dataset placeholder
(<blogcatalog.train>, cmd.py:100-104) names the file the reference's downloader would leave at
<dataset path>/blogcatalog/blogcatalog_train.txt (python/graphvite/dataset.py:121,183): it is resolved to that file when it
is there (--dataset-path, $GRAPHVITE_DATASET_PATH, default ~/.graphvite/dataset — the reference's default); nothing is
downloaded.  Baselines are looked up as the reference looks them up (every keyword must match a path component of ONE
configuration file) in --config-path / $GRAPHVITE_CONFIG_PATH / <package>/config."""
import argparse
import logging
import os
import re
import sys

import yaml

from . import application, optimizer
from .base import auto, init_logging


def dataset_path(given=None):
    return given or os.environ.get("GRAPHVITE_DATASET_PATH") or os.path.join(os.path.expanduser("~"), ".graphvite", "dataset")


def resolve_placeholder(node, path):
    """<name.split> -> <path>/name/name_split.txt, the file the reference's dataset module serves (dataset.py:183)."""
    name, _, split = node[1:-1].partition(".")
    file_name = os.path.join(path, name, "%s_%s.txt" % (name, split))
    if not name or not split or not os.path.exists(file_name):
        raise ValueError("dataset placeholder %s: %s is not there, and nothing is downloaded here; give a file name or put the "
                         "dataset under --dataset-path" % (node, file_name))
    return file_name


def load_config(config_file, datasets=None):
    with open(config_file, "r") as fin:
        cfg = yaml.safe_load(fin)

    def convert(node):
        if isinstance(node, dict):
            return {k: convert(v) for k, v in node.items()}
        if isinstance(node, list):
            return [convert(v) for v in node]
        if node == "auto":
            return auto
        if isinstance(node, str) and re.fullmatch(r"<[\w.]+>", node):
            return resolve_placeholder(node, dataset_path(datasets))
        return node

    return convert(cfg)


def config_path(given=None):
    """The directory of baseline configurations (cmd.py:35-43): given, $GRAPHVITE_CONFIG_PATH, or <package>/config."""
    candidates = [given, os.environ.get("GRAPHVITE_CONFIG_PATH"),
                  os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")]
    for path in candidates:
        if path and os.path.isdir(path):
            return os.path.realpath(path)
    raise IOError("Can't find a configuration directory: pass --config-path (the reference ships one, config/)")


def find_baseline(keywords, path):
    """cmd.py:196-213: the one configuration file whose path contains every keyword as a component."""
    configs = []
    for directory, _, files in os.walk(path):
        for name in files:
            file_name = os.path.join(directory, name)
            if all(re.search(r"[/\\_.]%s[/\\_.]" % re.escape(keyword), file_name) for keyword in keywords):
                configs.append(file_name)
    if not configs:
        raise ValueError("Can't find a baseline with keywords: %s" % ", ".join(keywords))
    if len(configs) > 1:
        raise ValueError("Ambiguous keywords. Candidates are:%s" % "\n    ".join([""] + [os.path.relpath(c, path) for c in sorted(configs)]))
    return configs[0]


def run_main(args):
    cfg = load_config(args.config, getattr(args, "dataset_path", None))
    if getattr(args, "epoch", None) is not None:  # `baseline --epoch` (cmd.py:224-225)
        cfg.setdefault("train", {})["num_epoch"] = args.epoch
    init_logging(logging.INFO)
    kind = cfg.get("application", "graph")
    if kind not in application.Application.application:
        raise ValueError("only the `graph` and `word graph` (node embedding) applications are implemented, not `%s`" % kind)
    resource = dict(cfg.get("resource", {}))
    if args.gpu is not None:  # run: the gpu id; baseline: the number of gpus (cmd.py:93-94, 220-221)
        resource["gpus"] = list(range(args.gpu)) if getattr(args, "count_gpus", False) else [args.gpu]
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


def baseline_main(args):
    path = config_path(args.config_path)
    args.config = find_baseline(args.keywords, path)
    print("running baseline: %s" % os.path.relpath(args.config, path))
    args.count_gpus = True
    return run_main(args)


def list_main(args):
    """cmd.py:240-260: the tree of baseline configurations."""
    path = config_path(args.config_path)
    print("list of baselines\n")
    indent, count = " " * 4, 0
    for directory, _, files in sorted(os.walk(path)):
        relative = os.path.relpath(directory, path)
        if relative == "template":
            continue
        depth = relative.count(os.sep)
        if relative != ".":
            depth += 1
            print("%s%s" % (indent * depth, os.path.basename(directory)))
        for name in sorted(files):
            print("%s%s" % (indent * (depth + 1), name))
        count += len(files)
        print()
    print("total: %d baselines" % count)
    return count


def main(argv=None):
    parser = argparse.ArgumentParser(prog="graphvite_amd", description="GraphVite node embedding on MI355X")
    sub = parser.add_subparsers(dest="command")
    run = sub.add_parser("run", help="run from a configuration file")
    run.add_argument("config", help="yaml configuration file")
    run.add_argument("--gpu", type=int, default=None, help="override the gpu id")
    baseline = sub.add_parser("baseline", help="reproduce a baseline benchmark")
    baseline.add_argument("keywords", nargs="+", help="keywords of the baseline (e.g. model, dataset)")
    baseline.add_argument("--gpu", type=int, default=None, help="override the number of GPUs")
    baseline.add_argument("--epoch", type=int, default=None, help="override the number of epochs")
    baseline.add_argument("--config-path", default=None, help="directory of baseline configurations")
    for command in (run, baseline):
        command.add_argument("--no-eval", dest="eval", action="store_false", help="turn off evaluation")
        command.add_argument("--cpu", type=int, default=None, help="override the number of CPU threads per GPU")
        command.add_argument("--dataset-path", default=None, help="where <dataset.split> placeholders are looked up")
    lister = sub.add_parser("list", help="list available baselines")
    lister.add_argument("--config-path", default=None, help="directory of baseline configurations")
    args = parser.parse_args(argv)
    if args.command == "run":
        run_main(args)
    elif args.command == "baseline":
        baseline_main(args)
    elif args.command == "list":
        list_main(args)
    else:
        parser.print_help()
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
