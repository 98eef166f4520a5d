"""GraphApplication: load -> build -> train -> evaluate -> save, the user-facing pipeline of the node-embedding
path (python/graphvite/application/application.py:38-241, 244-453 in the reference).  Keyword arguments are
forwarded verbatim to Graph.load / GraphSolver.build / GraphSolver.train, as the reference does."""
import logging
import pickle
import re
import time

import numpy as np

from .. import graph as graph_module
from .. import solver as solver_module
from ..base import auto, dtype, io

logger = logging.getLogger("graphvite_amd")


def easy_dict_class():
    """The class the reference pickles its models as (easydict.EasyDict: a dict whose keys are attributes as well).
    When the `easydict` package is not installed, a minimal stand-in is registered under that module name, so that
    pickles written here name `easydict.EasyDict` (and load in the reference) and the reference's pickles load here."""
    try:
        from easydict import EasyDict
        return EasyDict
    except ImportError:
        import sys
        import types

        class EasyDict(dict):
            def __init__(self, *args, **kwargs):
                super(EasyDict, self).__init__()
                for key, value in dict(*args, **kwargs).items():
                    self[key] = value

            def __setitem__(self, key, value):
                if isinstance(value, dict) and not isinstance(value, EasyDict):
                    value = EasyDict(value)
                super(EasyDict, self).__setitem__(key, value)

            def __getattr__(self, name):
                try:
                    return self[name]
                except KeyError:
                    raise AttributeError(name)

            def __setattr__(self, name, value):
                self[name] = value

            def __setstate__(self, state):  # the real class keeps a copy of its items in __dict__ and pickles it
                for key, value in (state or {}).items():
                    self[key] = value

            def update(self, *args, **kwargs):
                for key, value in dict(*args, **kwargs).items():
                    self[key] = value

        EasyDict.__module__, EasyDict.__qualname__ = "easydict", "EasyDict"
        module = types.ModuleType("easydict")
        module.EasyDict = EasyDict
        sys.modules["easydict"] = module
        return EasyDict


def _field(record, name):
    """record.name or record[name]: models arrive as EasyDicts, plain dicts (older files of this package) or objects."""
    if isinstance(record, dict):
        return record[name]
    return getattr(record, name)


def _timed(func):
    """@monitor.time of the reference (python/graphvite/util.py:148-167): logs the wall time of each stage."""
    def wrapper(self, *args, **kwargs):
        start = time.time()
        result = func(self, *args, **kwargs)
        logger.info("[time] %s.%s: %g s", type(self).__name__, func.__name__, time.time() - start)
        return result
    wrapper.__name__, wrapper.__doc__ = func.__name__, func.__doc__
    return wrapper


class ApplicationMixin(object):
    """
    General interface of graph applications.

    Parameters:
        dim (int): dimension of embeddings
        gpus (list of int, optional): GPU ids, default is all GPUs (one process per GPU, see GraphSolver)
        cpu_per_gpu (int, optional): number of CPU threads per GPU, default is all CPUs
        gpu_memory_limit (int, optional): memory limit per GPU in bytes, default is all memory
        float_type (dtype, optional): type of parameters
        index_type (dtype, optional): type of graph indexes
    """

    def __init__(self, dim, gpus=[], cpu_per_gpu=auto, gpu_memory_limit=auto, float_type=dtype.float32,
                 index_type=dtype.uint32):
        self.dim, self.gpus, self.cpu_per_gpu = dim, list(gpus), cpu_per_gpu
        self.gpu_memory_limit, self.float_type, self.index_type = gpu_memory_limit, float_type, index_type
        self.set_format()

    def get_graph(self, **kwargs):
        raise NotImplementedError

    def get_solver(self, **kwargs):
        raise NotImplementedError

    def set_format(self, delimiters=" \t\r\n", comment="#"):
        """Set the format for parsing input data."""
        self.delimiters, self.comment = delimiters, comment
        self.pattern = re.compile("[%s]" % self.delimiters)

    @_timed
    def load(self, **kwargs):
        """load(**kwargs): load a graph from file or Python object (arguments of Graph.load)."""
        self.graph = self.get_graph(**kwargs)
        if "file_name" in kwargs:
            self.graph.load(delimiters=self.delimiters, comment=self.comment, **kwargs)
        else:
            self.graph.load(**kwargs)

    @_timed
    def build(self, **kwargs):
        """build(**kwargs): build the solver from the graph (arguments of GraphSolver.build)."""
        self.solver = self.get_solver(**kwargs)
        self.solver.build(self.graph, **kwargs)

    @_timed
    def train(self, **kwargs):
        """train(**kwargs): train embeddings with the solver (arguments of GraphSolver.train)."""
        self.solver.train(**kwargs)

    @_timed
    def evaluate(self, task, **kwargs):
        """evaluate(task, **kwargs): evaluate the learned embeddings on a downstream task; returns a dict."""
        func_name = task.replace(" ", "_")
        if not hasattr(self, func_name):
            raise ValueError("Unknown task `%s`" % task)
        logger.info(io.header(task))
        result = getattr(self, func_name)(**kwargs)
        if isinstance(result, dict):
            for metric, value in sorted(result.items()):
                logger.warning("%s: %g", metric, value)
        return result

    # what the reference's generic attribute filters pick up from its pybind objects (application.py:155-183 over
    # bind.h:137-143, 415-436 and 824-989): name maps, embeddings, and the int / float / str read-only members
    _GRAPH_HYPERPARAMETERS = ("num_vertex", "num_edge", "as_undirected", "normalization")
    _SOLVER_HYPERPARAMETERS = ("num_partition", "num_negative", "negative_sample_exponent", "negative_weight", "model",
                               "num_epoch", "resume", "episode_size", "batch_size", "augmentation_step",
                               "random_walk_length", "random_walk_batch_size", "shuffle_base", "p", "q", "positive_reuse",
                               "log_frequency", "num_worker", "num_sampler", "gpu_memory_limit", "gpu_memory_cost")

    @_timed
    def save_model(self, file_name, save_hyperparameter=False):
        """Save the graph's name maps and the solver's embeddings with pickle, in the reference's own layout
        (application.py:145-187): an EasyDict {graph: {name2id, id2name}, solver: {vertex_embeddings,
        context_embeddings}}, plus — with save_hyperparameter — the graph's and solver's scalar members and
        solver.optimizer (its scalars, schedule as the schedule's type name).  A file written here loads in the reference
        (`model.graph.name2id`, `model.solver.vertex_embeddings`) and the other way round."""
        logger.warning("save model to `%s`", file_name)
        EasyDict = easy_dict_class()
        model = EasyDict()
        model.graph = EasyDict(name2id=dict(self.graph.name2id), id2name=list(self.graph.id2name))
        model.solver = EasyDict(vertex_embeddings=np.array(self.solver.vertex_embeddings),
                                context_embeddings=np.array(self.solver.context_embeddings))
        if save_hyperparameter:
            for name in self._GRAPH_HYPERPARAMETERS:
                model.graph[name] = getattr(self.graph, name)
            for name in self._SOLVER_HYPERPARAMETERS:
                model.solver[name] = getattr(self.solver, name)
            optimizer = self.solver.optimizer
            extra = {"Momentum": ("momentum",), "AdaGrad": ("epsilon",), "RMSprop": ("alpha", "epsilon"),
                     "Adam": ("beta1", "beta2", "epsilon")}.get(optimizer.type, ())
            model.solver.optimizer = EasyDict(type=optimizer.type, lr=optimizer.init_lr,
                                              weight_decay=optimizer.weight_decay)
            for name in extra:
                model.solver.optimizer[name] = getattr(optimizer, name)
            model.solver.optimizer.schedule = optimizer.schedule.type
        with open(file_name, "wb") as fout:
            pickle.dump(model, fout, protocol=pickle.HIGHEST_PROTOCOL)

    @_timed
    def load_model(self, file_name):
        """Load embeddings saved by save_model — here or by the reference — into the (already built) solver, matching
        nodes by name (application.py:131-142)."""
        logger.warning("load model from `%s`", file_name)
        easy_dict_class()  # a reference pickle names easydict.EasyDict; make sure something answers to that name
        with open(file_name, "rb") as fin:
            model = pickle.load(fin)
        graph, solver = _field(model, "graph"), _field(model, "solver")
        mapping = self.get_mapping(self.graph.id2name, _field(graph, "name2id"))
        self.solver.vertex_embeddings[:] = np.asarray(_field(solver, "vertex_embeddings"))[mapping]
        self.solver.context_embeddings[:] = np.asarray(_field(solver, "context_embeddings"))[mapping]

    def get_mapping(self, id2name, name2id):
        mapping = []
        for name in id2name:
            if name not in name2id:
                raise ValueError("Can't find the embedding for node `%s`" % name)
            mapping.append(name2id[name])
        return np.asarray(mapping, np.int64)

    def tokenize(self, line):
        comment_start = line.find(self.comment)
        if comment_start != -1:
            line = line[:comment_start]
        return [t for t in self.pattern.split(line) if t]

    def read_columns(self, file_name, num_column, what):
        """The columns of a delimiter-separated text file (comments and blank lines skipped) as `num_column` lists of
        strings — the evaluation files of every task are of this form (application.py:318-327, 379-404)."""
        columns = tuple([] for _ in range(num_column))
        with open(file_name, "r") as fin:
            for number, line in enumerate(fin, 1):
                tokens = self.tokenize(line)
                if not tokens:
                    continue
                if len(tokens) != num_column:
                    raise ValueError("%s `%s`, line %d: expected %d fields, found %d" % (what, file_name, number,
                                                                                       num_column, len(tokens)))
                for column, token in zip(columns, tokens):
                    column.append(token)
        return columns

    @staticmethod
    def one_source(data, file_name, what):
        """Evaluation input comes either as Python lists or as a file, never both, never neither."""
        given = [d is not None for d in data]
        if file_name and any(given):
            raise ValueError("%s data and file should not be provided at the same time" % what)
        if not file_name and not all(given):
            raise ValueError("Either %s data or a file name should be provided" % what.lower())
        return bool(file_name)

    def name_map(self, dicts, names):
        """Map columns of names to ids, dropping the rows with an unknown name (application.py:216-236)."""
        keep = [all(name in d for d, name in zip(dicts, row)) for row in zip(*names)]
        return tuple([d[name] for name, k in zip(column, keep) if k] for d, column in zip(dicts, names))


class GraphApplication(ApplicationMixin):
    """
    Node embedding application (DeepWalk, LINE, node2vec).

    Parameters:
        dim (int): dimension of embeddings
        gpus (list of int, optional): GPU ids, default is all GPUs
        cpu_per_gpu (int, optional): number of CPU threads per GPU, default is all CPUs
        float_type (dtype, optional): type of parameters
        index_type (dtype, optional): type of graph indexes
    """

    def get_graph(self, **kwargs):
        return graph_module.Graph(self.index_type)

    def get_solver(self, **kwargs):
        num_sampler_per_worker = auto if self.cpu_per_gpu == auto else self.cpu_per_gpu - 1
        return solver_module.GraphSolver(self.dim, self.float_type, self.index_type, self.gpus,
                                         num_sampler_per_worker, self.gpu_memory_limit)

    def set_parameters(self, model):
        """Copy embeddings from another application, matching nodes by name (application.py:288-291)."""
        mapping = self.get_mapping(self.graph.id2name, model.graph.name2id)
        self.solver.vertex_embeddings[:] = model.solver.vertex_embeddings[mapping]
        self.solver.context_embeddings[:] = model.solver.context_embeddings[mapping]

    def node_classification(self, X=None, Y=None, file_name=None, portions=(0.02,), normalization=False, times=1,
                            patience=100):
        """
        Evaluate node embeddings on node classification task: one-vs-rest logistic regression on the vertex
        embeddings for each training portion, macro / micro F1 with the top-k-labels rule
        (application.py:293-351, 456-533).  Downstream evaluation, so plain torch like the reference's.

        Returns:
            dict: macro-F1 & micro-F1 averaged over all trials
        """
        if self.one_source((X, Y), file_name, "Evaluation"):
            X, Y = self.read_columns(file_name, 2, "label file")
        name2id = self.graph.name2id
        class2id = {c: i for i, c in enumerate(np.unique(Y))}
        new_X, new_Y = self.name_map((name2id, class2id), ([str(x) for x in X], list(Y)))
        logger.info("effective labels: %d / %d", len(new_X), len(X))
        # a (node, label) line that occurs n times counts n times, as in the reference's coo_matrix -> dense sum
        # (application.py:336-338): it enters the node's label count for the top-k rule and the F1 totals
        labels = np.zeros((self.graph.num_vertex, len(class2id)), np.int64)
        np.add.at(labels, (np.asarray(new_X, np.int64), np.asarray(new_Y, np.int64)), 1)
        indexes = np.nonzero(labels.sum(1) > 0)[0]  # discard non-labeled nodes
        labels = labels[indexes]
        embeddings = np.array(self.solver.vertex_embeddings[indexes])
        metrics = {}
        for portion in portions:
            metrics.update(linear_classification(embeddings, labels, portion, normalization, times, patience))
        return metrics

    def link_prediction(self, H=None, T=None, Y=None, file_name=None, filter_H=None, filter_T=None, filter_file=None):
        """
        Evaluate node embeddings on link prediction task: AUC of score = <vertex[h], context[t]>
        (application.py:353-453; scores are computed with the solver's predict kernel).
        """
        if self.one_source((H, T, Y), file_name, "Evaluation"):
            H, T, Y = self.read_columns(file_name, 3, "edge file")
        if filter_file or filter_H is not None or filter_T is not None:
            if self.one_source((filter_H, filter_T), filter_file, "Filter"):
                filter_H, filter_T = self.read_columns(filter_file, 2, "filter file")
        else:
            filter_H, filter_T = [], []

        name2id = self.graph.name2id
        Y = [int(y) for y in Y]
        new_H, new_T, new_Y = self.name_map((name2id, name2id, {0: 0, 1: 1}), ([str(h) for h in H],
                                                                                [str(t) for t in T], Y))
        logger.info("effective edges: %d / %d", len(new_H), len(H))
        H, T, Y = new_H, new_T, new_Y
        fH, fT = self.name_map((name2id, name2id), ([str(h) for h in filter_H], [str(t) for t in filter_T]))
        logger.info("effective filter edges: %d / %d", len(fH), len(filter_H))
        filters = set(zip(fH, fT))
        keep = [(h, t, y) for h, t, y in zip(H, T, Y) if (h, t) not in filters]
        logger.info("remaining edges: %d / %d", len(keep), len(H))
        H = np.asarray([k[0] for k in keep], np.int64)
        T = np.asarray([k[1] for k in keep], np.int64)
        Y = np.asarray([k[2] for k in keep], np.int64)

        score = self.solver.predict(np.stack([H, T], 1)) if len(H) else np.zeros(0, np.float32)
        order = np.argsort(-score, kind="stable")
        Y = Y[order]
        hit = np.cumsum(Y)
        total = int((Y == 0).sum()) * int((Y == 1).sum())
        auc = float(hit[Y == 0].sum()) / total if total else float("nan")
        return {"AUC": auc}


def linear_classification(embeddings, labels, portion, normalization=False, times=1, patience=100):
    """One-vs-rest logistic regression on frozen embeddings, as the reference's linear_classification
    (application.py:456-533): SGD(lr 1, weight decay 2e-5, momentum 0.9) on the full training set until the loss
    has not improved for `patience` epochs; a test node with n true labels is assigned its n top-scoring classes."""
    import torch
    from torch.nn import functional as F

    device = "cuda" if torch.cuda.is_available() else "cpu"
    num_sample, num_class = labels.shape
    num_train = int(num_sample * portion)
    if normalization:
        embeddings = embeddings / np.linalg.norm(embeddings, axis=1, keepdims=True)
    features = torch.as_tensor(embeddings, dtype=torch.float32, device=device)
    all_labels = torch.as_tensor(labels, device=device)
    macro_f1s, micro_f1s = [], []
    for _ in range(times):
        samples = np.random.permutation(num_sample)
        train = labels[samples[:num_train]]
        rows, classes = np.nonzero(train)  # one training example per (node, label) pair
        train_x = features[torch.as_tensor(samples[:num_train][rows], device=device)]
        train_y = torch.zeros((len(rows), num_class), device=device)
        train_y[torch.arange(len(rows), device=device), torch.as_tensor(classes, device=device)] = 1
        test_index = torch.as_tensor(samples[num_train:], device=device)
        test_x, test_y = features[test_index], all_labels[test_index]

        linear = torch.nn.Linear(features.shape[1], num_class, bias=True).to(device)
        optimizer = torch.optim.SGD(linear.parameters(), lr=1, weight_decay=2e-5, momentum=0.9)
        best_loss, best_epoch = float("inf"), -1
        for epoch in range(100000):
            optimizer.zero_grad()
            loss = F.binary_cross_entropy_with_logits(linear(train_x), train_y)
            loss.backward()
            optimizer.step()
            loss = loss.item()
            if loss < best_loss:
                best_epoch, best_loss = epoch, loss
            if epoch == best_epoch + patience:
                break

        with torch.no_grad():
            logits = linear(test_x)
            num_labels = test_y.sum(dim=1, keepdim=True)
            ordered, _ = logits.sort(dim=1, descending=True)
            thresholds = ordered.gather(dim=1, index=num_labels - 1)
            predictions = (logits >= thresholds).long()
            tp = (predictions & test_y).sum(dim=0).float()
            t, p = test_y.sum(dim=0).float(), predictions.sum(dim=0).float()
            # a class with neither true nor predicted test labels scores 0, not 0 / 0 (the reference divides blindly)
            macro_f1s.append((2 * tp / (t + p).clamp(min=1)).mean().item())
            micro_f1s.append((2 * tp.sum() / (t.sum() + p.sum()).clamp(min=1)).item())
    return {"macro-F1@%g%%" % (portion * 100): float(np.mean(macro_f1s)),
            "micro-F1@%g%%" % (portion * 100): float(np.mean(micro_f1s))}


class WordGraphApplication(GraphApplication):
    """
    Word node embedding application: the graph of word co-occurrences of a corpus (WordGraph), embedded with the same
    solver and the same training path as GraphApplication (application.py:536-573).

    Parameters: as GraphApplication.
    """

    def get_graph(self, **kwargs):
        return graph_module.WordGraph(self.index_type)


class Application(object):
    """
    Application(type, *args, **kwargs)
    Create an application instance of any type (application.py:1371-1393).

    Parameters:
        type (str): application type: 'graph' or 'word graph' ('knowledge graph' and 'visualization' belong to other
            solvers of the reference and are not part of this package)
    """

    application = {"graph": GraphApplication, "word graph": WordGraphApplication}

    def __new__(cls, type, *args, **kwargs):
        if type in cls.application:
            return cls.application[type](*args, **kwargs)
        raise ValueError("Unknown application `%s`" % type)
