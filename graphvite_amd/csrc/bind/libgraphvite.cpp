// libgraphvite — the pybind11 module the reference's Python package loads (python/graphvite/__init__.py:27-36 finds
// "libgraphvite" and python/graphvite/helper.py:83-105 dispatches `GraphSolver(dim, float_type, index_type)` to
// `solver.GraphSolver_<dim>_<float>_<index>`), rebuilt over the MI355X runtime: every class here is a thin binding of
// the C ABI in include/gvs.h (graph store), include/gvx.h (native solver engine) and include/gvk.h (kernels), all in
// libgvk.so.  Counterpart of src/graphvite.cu:28-106 + include/bind.h for the node-embedding path: optimizer.*,
// graph.Graph / graph.WordGraph, solver.GraphSolver x {32, 64, 96, 128, 256, 512}, dtype, auto, init_logging, io.*,
// KiB / MiB / GiB.  The knowledge-graph and visualization classes belong to other solvers and are not bound.
// Every bound function releases the GIL (bind.h:44); nothing here holds Python state while the GPUs work.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <functional>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "gvx.h"

namespace py = pybind11;
typedef py::call_guard<py::gil_scoped_release> no_gil;

namespace {

const char *const kVersion = "0.2.2";  // the API level this module is a drop-in for (src/graphvite.cu:26)
constexpr int kAuto = GVX_AUTO;

enum DType { uint32 = 0, uint64, float32, float64 };  // bind.h:53-58

[[noreturn]] void raise(int code, const char *what) {
    const std::string message = std::string(what) + ": " + gvk_last_error();
    if (code == GVK_ENOMEM) throw std::bad_alloc();
    if (code == GVK_EINVAL || code == GVK_EDIM) throw py::value_error(message);
    throw std::runtime_error(message);
}

void check(int code, const char *what) {
    if (code != GVK_OK) raise(code, what);
}

// ---- io (include/util/io.h) -------------------------------------------------------------------------------------------

std::string size_string(size_t size) {
    char buf[64];
    const double s = (double)size;
    if (s >= (double)((size_t)1 << 40)) snprintf(buf, sizeof(buf), "%.3g TiB", s / (double)((size_t)1 << 40));
    else if (s >= (1 << 30)) snprintf(buf, sizeof(buf), "%.3g GiB", s / (1 << 30));
    else if (s >= (1 << 20)) snprintf(buf, sizeof(buf), "%.3g MiB", s / (1 << 20));
    else if (s >= (1 << 10)) snprintf(buf, sizeof(buf), "%.3g KiB", s / (1 << 10));
    else snprintf(buf, sizeof(buf), "%zu B", size);
    return buf;
}
std::string yes_no(bool x) { return x ? "yes" : "no"; }
std::string block(const std::string &content) {
    return "\n" + std::string(40, '<') + "\n" + content + "\n" + std::string(40, '>');
}
std::string header(const std::string &content) {
    const int pad = std::max(40 - (int)content.size() - 2, 0);
    return std::string(pad / 2, '-') + " " + content + " " + std::string(pad - pad / 2, '-');
}

// ---- optimizers (include/core/optimizer.h:36-300, bind.h:757-999) ------------------------------------------------------

struct LRSchedule {
    typedef std::function<float(int, int)> ScheduleFunction;
    std::string type;
    ScheduleFunction schedule_function;
    LRSchedule(const std::string &t = "constant") : type(t) {
        if (t != "constant" && t != "linear") throw py::value_error("Invalid schedule type `" + t + "`");
    }
    LRSchedule(const ScheduleFunction &f) : type("custom"), schedule_function(f) {}
    std::string info() const { return "lr schedule: " + type; }
};

struct Optimizer {
    std::string type = "Default";
    float lr = 0, weight_decay = 0;
    LRSchedule schedule{"linear"};
    float momentum = 0, alpha = 0, beta1 = 0, beta2 = 0, epsilon = 0;
    Optimizer(int t = kAuto) {
        if (t != kAuto) throw py::value_error("Optimizer(type): only `auto` is a valid integer type");
    }
    Optimizer(float learning_rate) : lr(learning_rate) {}
    Optimizer(const std::string &t, float learning_rate, float wd, const LRSchedule &s)
        : type(t), lr(learning_rate), weight_decay(wd), schedule(s) {}
    std::string info() const {
        std::stringstream ss;
        ss << "optimizer: " << type << "\nlearning rate: " << lr << ", " << schedule.info() << "\nweight decay: " << weight_decay;
        if (type == "Momentum") ss << "\nmomentum: " << momentum;
        if (type == "AdaGrad") ss << "\nepsilon: " << epsilon;
        if (type == "RMSprop") ss << "\nalpha: " << alpha << ", epsilon: " << epsilon;
        if (type == "Adam") ss << "\nbeta1: " << beta1 << ", beta2: " << beta2 << ", epsilon: " << epsilon;
        return ss.str();
    }
};
struct SGD : Optimizer {
    SGD(float lr, float wd, const LRSchedule &s) : Optimizer("SGD", lr, wd, s) {}
};
struct Momentum : Optimizer {
    Momentum(float lr, float wd, float m, const LRSchedule &s) : Optimizer("Momentum", lr, wd, s) { momentum = m; }
};
struct AdaGrad : Optimizer {
    AdaGrad(float lr, float wd, float eps, const LRSchedule &s) : Optimizer("AdaGrad", lr, wd, s) { epsilon = eps; }
};
struct RMSprop : Optimizer {
    RMSprop(float lr, float wd, float a, float eps, const LRSchedule &s) : Optimizer("RMSprop", lr, wd, s) {
        alpha = a, epsilon = eps;
    }
};
struct Adam : Optimizer {
    Adam(float lr, float wd, float b1, float b2, float eps, const LRSchedule &s) : Optimizer("Adam", lr, wd, s) {
        beta1 = b1, beta2 = b2, epsilon = eps;
    }
};

float call_schedule(int batch_id, int num_batch, void *user) {  // the engine runs without the GIL
    py::gil_scoped_acquire gil;
    return (*static_cast<LRSchedule::ScheduleFunction *>(user))(batch_id, num_batch);
}

// ---- graphs (bind.h:109-234 over include/gvs.h) ----------------------------------------------------------------------------

class Graph {
public:
    gvs_graph *handle;
    size_t num_vertex = 0, num_edge = 0;
    bool as_undirected = true, normalization = false;
    std::unordered_map<std::string, unsigned int> name2id;
    std::vector<std::string> id2name;

    Graph() : handle(gvs_graph_create()) {
        if (!handle) throw std::bad_alloc();
    }
    virtual ~Graph() { gvs_graph_destroy(handle); }
    Graph(const Graph &) = delete;

    void refresh() {
        num_vertex = gvs_graph_num_vertex(handle), num_edge = gvs_graph_num_edge(handle);
        as_undirected = gvs_graph_as_undirected(handle) != 0, normalization = gvs_graph_normalization(handle) != 0;
        id2name.assign(num_vertex, std::string());
        name2id.clear();
        name2id.reserve(num_vertex);
        std::vector<char> buf(256);
        for (size_t v = 0; v < num_vertex; v++) {
            int64_t n = gvs_graph_id2name(handle, (uint32_t)v, buf.data(), buf.size());
            if (n >= (int64_t)buf.size()) {
                buf.resize(n + 1);
                gvs_graph_id2name(handle, (uint32_t)v, buf.data(), buf.size());
            }
            id2name[v] = buf.data();
            name2id[id2name[v]] = (unsigned int)v;
        }
    }
    void load_file(const char *file_name, bool undirected, bool norm, const char *delimiters, const char *comment) {
        check(gvs_graph_load_file(handle, file_name, undirected, norm, delimiters, comment), "Graph.load");
        refresh();
    }
    void load_edge_list(const std::vector<std::tuple<std::string, std::string>> &edges, bool undirected, bool norm) {
        std::vector<const char *> u(edges.size()), v(edges.size());
        for (size_t i = 0; i < edges.size(); i++) u[i] = std::get<0>(edges[i]).c_str(), v[i] = std::get<1>(edges[i]).c_str();
        check(gvs_graph_load_names(handle, u.data(), v.data(), nullptr, edges.size(), undirected, norm), "Graph.load");
        refresh();
    }
    void load_weighted_edge_list(const std::vector<std::tuple<std::string, std::string, float>> &edges, bool undirected,
                                 bool norm) {
        std::vector<const char *> u(edges.size()), v(edges.size());
        std::vector<float> w(edges.size());
        for (size_t i = 0; i < edges.size(); i++)
            u[i] = std::get<0>(edges[i]).c_str(), v[i] = std::get<1>(edges[i]).c_str(), w[i] = std::get<2>(edges[i]);
        check(gvs_graph_load_names(handle, u.data(), v.data(), w.data(), edges.size(), undirected, norm), "Graph.load");
        refresh();
    }
    void save(const char *file_name, bool weighted, bool anonymous) {
        check(gvs_graph_save(handle, file_name, weighted, anonymous), "Graph.save");
    }
    virtual std::string name() const { return "Graph"; }
    std::string info() const {
        std::stringstream ss;
        ss << name() << "<uint32>\n" << header("Graph") << "\n#vertex: " << num_vertex << ", #edge: " << num_edge
           << "\nas undirected: " << yes_no(as_undirected) << ", normalization: " << yes_no(normalization);
        return ss.str();
    }
};

class WordGraph : public Graph {
public:
    void load_file_compact(const char *file_name, int window, int min_count, bool norm, const char *delimiters,
                           const char *comment) {
        check(gvs_graph_load_corpus(handle, file_name, window, min_count, norm, delimiters, comment), "WordGraph.load");
        refresh();
    }
    std::string name() const override { return "WordGraph"; }
};

// ---- GraphSolver (bind.h:383-513 over include/gvx.h) ------------------------------------------------------------------------

class GraphSolverBase {
public:
    gvx_solver *handle = nullptr;
    Optimizer optimizer;  // the Python-visible copy (bind.h:411); owns the schedule function the engine calls back
    py::object graph_keepalive;

    GraphSolverBase(int dim, const std::vector<int> &device_ids, int num_sampler_per_worker, size_t gpu_memory_limit,
                    bool device_sampling, int64_t seed, const std::string &fidelity) {
        if (fidelity != "auto" && fidelity != "throughput" && fidelity != "reference")
            throw py::value_error("fidelity must be 'auto', 'throughput' or 'reference', not '" + fidelity + "'");
        handle = gvx_solver_create(dim, device_ids.data(), (int)device_ids.size(), num_sampler_per_worker, gpu_memory_limit);
        if (!handle) raise(GVK_EINVAL, "GraphSolver");
        check(gvx_solver_set(handle, GVX_FIDELITY, fidelity == "auto" ? -1 : (fidelity == "reference" ? 1 : 0)), "GraphSolver");
        check(gvx_solver_set(handle, GVX_DEVICE_SAMPLING, device_sampling), "GraphSolver");
        check(gvx_solver_set(handle, GVX_SEED, seed), "GraphSolver");
    }
    ~GraphSolverBase() { gvx_solver_destroy(handle); }
    GraphSolverBase(const GraphSolverBase &) = delete;

    gvx_solver_members members() const {
        gvx_solver_members m;
        check(gvx_solver_get(handle, &m), "GraphSolver");
        return m;
    }
    void build(const Graph &graph, const Optimizer &opt, int num_partition, int num_negative, int batch_size, int episode_size) {
        optimizer = opt;
        gvx_optimizer o{};
        static const std::unordered_map<std::string, int> types = {{"Default", -1}, {"SGD", GVK_SGD}, {"Momentum", GVK_MOMENTUM},
                                                                   {"AdaGrad", GVK_ADAGRAD}, {"RMSprop", GVK_RMSPROP}, {"Adam", GVK_ADAM}};
        o.type = types.at(optimizer.type);
        o.lr = optimizer.lr, o.weight_decay = optimizer.weight_decay, o.epsilon = optimizer.epsilon;
        o.hp0 = optimizer.type == "Momentum" ? optimizer.momentum : (optimizer.type == "RMSprop" ? optimizer.alpha : optimizer.beta1);
        o.hp1 = optimizer.beta2;
        o.schedule = optimizer.schedule.type == "constant" ? 0 : (optimizer.schedule.type == "linear" ? 1 : 2);
        if (o.schedule == 2) o.schedule_function = call_schedule, o.user = &optimizer.schedule.schedule_function;
        check(gvx_solver_build(handle, graph.handle, &o, num_partition, num_negative, batch_size, episode_size), "GraphSolver.build");
        if (optimizer.type == "Default") {  // what build() resolved `auto` to (solver.h:290-296)
            const gvx_solver_members m = members();
            optimizer.type = "SGD", optimizer.lr = m.optimizer.lr, optimizer.weight_decay = m.optimizer.weight_decay;
            optimizer.schedule = LRSchedule("linear");
        }
    }
    void train(const std::string &model, int num_epoch, bool resume, int augmentation_step, int random_walk_length,
               int random_walk_batch_size, int shuffle_base, float p, float q, int positive_reuse,
               float negative_sample_exponent, float negative_weight, int log_frequency) {
        gvx_train_config c{};
        c.model = model.c_str(), c.num_epoch = num_epoch, c.resume = resume, c.augmentation_step = augmentation_step;
        c.random_walk_length = random_walk_length, c.random_walk_batch_size = random_walk_batch_size;
        c.shuffle_base = shuffle_base, c.p = p, c.q = q, c.positive_reuse = positive_reuse;
        c.negative_sample_exponent = negative_sample_exponent, c.negative_weight = negative_weight, c.log_frequency = log_frequency;
        check(gvx_solver_train(handle, &c), "GraphSolver.train");
    }
    void clear() { check(gvx_solver_clear(handle), "GraphSolver.clear"); }
    std::string info() {
        std::string text(gvx_solver_info(handle, nullptr, 0) + 1, '\0');
        gvx_solver_info(handle, &text[0], text.size());
        text.resize(text.size() - 1);
        return text;
    }
    py::array_t<float> view(int which) {  // mutable but unassignable numpy views of the host tables (bind.h:90-106)
        uint64_t rows = 0;
        float *data = gvx_solver_embeddings(handle, which, &rows);
        const size_t dim = (size_t)members().dim;
        if (!data) rows = 0;
        py::capsule keep(data ? (void *)data : (void *)this, [](void *) {});
        return py::array_t<float>({(size_t)rows, dim}, {sizeof(float) * dim, sizeof(float)}, data, keep);
    }
    py::array_t<float> predict(py::array_t<int64_t, py::array::c_style | py::array::forcecast> samples) {
        if (samples.ndim() != 2 || samples.shape(1) != 2) {
            std::stringstream ss;
            ss << "Expect an array with shape (?, 2), but shape (";
            for (py::ssize_t i = 0; i < samples.ndim(); i++) ss << (i ? ", " : "") << samples.shape(i);
            throw py::value_error(ss.str() + ") is found");
        }
        const size_t n = (size_t)samples.shape(0);
        py::array_t<float> logits(n);
        int rc;
        {
            py::gil_scoped_release release;
            rc = gvx_solver_predict(handle, samples.data(), n, logits.mutable_data());
        }
        check(rc, "GraphSolver.predict");
        return logits;
    }
};

template <int dim>
class GraphSolver : public GraphSolverBase {
public:
    GraphSolver(const std::vector<int> &device_ids, int num_sampler_per_worker, size_t gpu_memory_limit, bool device_sampling,
                int64_t seed, const std::string &fidelity)
        : GraphSolverBase(dim, device_ids, num_sampler_per_worker, gpu_memory_limit, device_sampling, seed, fidelity) {}
};

template <class T, class... Extra>
py::class_<T, Extra...> template_class(py::handle scope, const char *name, const std::string &full_name, const char *doc) {
    py::class_<T, Extra...> cls(scope, full_name.c_str());
    cls.attr("__doc__") = doc;
    cls.attr("__name__") = py::str(name);  // override instance name with template name (bind.h:127-128)
    cls.attr("__qualname__") = py::str(name);
    return cls;
}

template <int dim>
void bind_solver(py::module &solver) {
    typedef GraphSolver<dim> Solver;
    auto cls = template_class<Solver, GraphSolverBase>(
        solver, "GraphSolver", "GraphSolver_" + std::to_string(dim) + "_f_j",
        "GraphSolver(dim, float_type=dtype.float32, index_type=dtype.uint32, device_ids=[], num_sampler_per_worker=auto, "
        "gpu_memory_limit=auto)\n"
        "        Graph embedding solver.\n\n"
        "        Parameters:\n"
        "            dim (int): dimension of embeddings\n"
        "            float_type (dtype): type of parameters\n"
        "            index_type (dtype): type of node indexes\n"
        "            device_ids (list of int, optional): GPU ids, [] for auto\n"
        "            num_sampler_per_worker (int, optional): number of sampler thread per GPU\n"
        "            gpu_memory_limit (int, optional): memory limit for each GPU in bytes\n"
        "            device_sampling (bool, optional): beyond the reference — draw the positive samples on the GPUs\n"
        "                instead of the CPU sampler threads\n"
        "            seed (int, optional): beyond the reference — seeds the initial embeddings, the samplers and the\n"
        "                negative draws (the reference seeds them from one process-wide generator)\n"
        "            fidelity (str, optional): beyond the reference — 'auto' (default): the hub rows of a hub-heavy graph are\n"
        "                trained by chains wherever chains exist (SGD; the reference's sequential learning quality);\n"
        "                'reference': the same, an error where they do not; 'throughput': every row pair by pair\n"
        "                (gvx.h GVX_FIDELITY)\n        ");
    cls.def(py::init<std::vector<int>, int, size_t, bool, int64_t, std::string>(), no_gil(),
            py::arg("device_ids") = std::vector<int>(), py::arg("num_sampler_per_worker") = kAuto,
            py::arg("gpu_memory_limit") = kAuto, py::arg("device_sampling") = false, py::arg("seed") = 0,
            py::arg("fidelity") = "auto");
}

}  // namespace

PYBIND11_MODULE(libgraphvite, module) {
    py::options options;
    options.disable_function_signatures();

    // optimizers
    auto optimizer = module.def_submodule("optimizer");
    py::class_<LRSchedule>(optimizer, "LRSchedule",
                           "LRSchedule(*args, **kwargs)\n        Learning Rate Schedule.\n\n"
                           "        .. function:: LRSchedule(type='constant')\n        .. function:: LRSchedule(schedule_function)\n")
        .def_readonly("type", &LRSchedule::type)
        .def_readonly("schedule_function", &LRSchedule::schedule_function)
        .def(py::init<std::string>(), no_gil(), py::arg("type") = "constant")
        .def(py::init<LRSchedule::ScheduleFunction>(), py::arg("schedule_function"))
        .def("__repr__", &LRSchedule::info, no_gil());
    py::implicitly_convertible<std::string, LRSchedule>();
    py::implicitly_convertible<LRSchedule::ScheduleFunction, LRSchedule>();
    py::class_<Optimizer>(optimizer, "Optimizer",
                          "Optimizer(*args, **kwargs)\n        General interface of first-order optimizers.\n\n"
                          "        .. function:: Optimizer(type)\n        .. function:: Optimizer(lr=1e-4)\n")
        .def_readonly("type", &Optimizer::type)
        .def_readonly("lr", &Optimizer::lr)
        .def_readonly("weight_decay", &Optimizer::weight_decay)
        .def_readonly("schedule", &Optimizer::schedule)
        .def(py::init<int>(), no_gil(), py::arg("type") = kAuto)
        .def(py::init<float>(), no_gil(), py::arg("lr") = 1e-4)
        .def("__repr__", &Optimizer::info, no_gil());
    py::implicitly_convertible<int, Optimizer>();
    py::implicitly_convertible<float, Optimizer>();
    py::class_<SGD, Optimizer>(optimizer, "SGD", "SGD(lr=1e-4, weight_decay=0, schedule='linear')\n        Stochastic gradient descent optimizer.\n")
        .def(py::init<float, float, LRSchedule>(), py::arg("lr") = 1e-4, py::arg("weight_decay") = 0, py::arg("schedule") = "linear");
    py::class_<Momentum, Optimizer>(optimizer, "Momentum",
                                    "Momentum(lr=1e-4, weight_decay=0, momentum=0.999, schedule='linear')\n        Momentum optimizer.\n")
        .def_readonly("momentum", &Momentum::momentum)
        .def(py::init<float, float, float, LRSchedule>(), py::arg("lr") = 1e-4, py::arg("weight_decay") = 0,
             py::arg("momentum") = 0.999, py::arg("schedule") = "linear");
    py::class_<AdaGrad, Optimizer>(optimizer, "AdaGrad",
                                   "AdaGrad(lr=1e-4, weight_decay=0, epsilon=1e-10, schedule='linear')\n        AdaGrad optimizer.\n")
        .def_readonly("epsilon", &AdaGrad::epsilon)
        .def(py::init<float, float, float, LRSchedule>(), py::arg("lr") = 1e-4, py::arg("weight_decay") = 0,
             py::arg("epsilon") = 1e-10, py::arg("schedule") = "linear");
    py::class_<RMSprop, Optimizer>(optimizer, "RMSprop",
                                   "RMSprop(lr=1e-4, weight_decay=0, alpha=0.999, epsilon=1e-8, schedule='linear')\n        RMSprop optimizer.\n")
        .def_readonly("alpha", &RMSprop::alpha)
        .def_readonly("epsilon", &RMSprop::epsilon)
        .def(py::init<float, float, float, float, LRSchedule>(), py::arg("lr") = 1e-4, py::arg("weight_decay") = 0,
             py::arg("alpha") = 0.999, py::arg("epsilon") = 1e-8, py::arg("schedule") = "linear");
    py::class_<Adam, Optimizer>(optimizer, "Adam",
                                "Adam(lr=1e-4, weight_decay=0, beta1=0.999, beta2=0.99999, epsilon=1e-8, schedule='linear')\n        Adam optimizer.\n")
        .def_readonly("beta1", &Adam::beta1)
        .def_readonly("beta2", &Adam::beta2)
        .def_readonly("epsilon", &Adam::epsilon)
        .def(py::init<float, float, float, float, float, LRSchedule>(), py::arg("lr") = 1e-4, py::arg("weight_decay") = 0,
             py::arg("beta1") = 0.999, py::arg("beta2") = 0.99999, py::arg("epsilon") = 1e-8, py::arg("schedule") = "linear");

    // graphs
    auto graph = module.def_submodule("graph");
    auto py_graph = template_class<Graph>(graph, "Graph", "Graph_j",
                                          "Graph(index_type=dtype.uint32)\n        Normal graphs without attributes.\n\n"
                                          "        Parameters:\n            index_type (dtype): type of node indexes\n        ");
    py_graph.def_readonly("num_vertex", &Graph::num_vertex)
        .def_readonly("num_edge", &Graph::num_edge)
        .def_readonly("as_undirected", &Graph::as_undirected)
        .def_readonly("normalization", &Graph::normalization)
        .def_readonly("name2id", &Graph::name2id, "Map of node name to index.")
        .def_readonly("id2name", &Graph::id2name, "Map of node index to name.")
        .def(py::init<>(), no_gil())
        .def("load", &Graph::load_file, no_gil(), py::arg("file_name"), py::arg("as_undirected") = true,
             py::arg("normalization") = false, py::arg("delimiters") = " \t\r\n", py::arg("comment") = "#",
             "load(*args, **kwargs)\n            Load a graph from an edge-list file or an edge list.\n\n"
             "            .. function:: load(file_name, as_undirected=True, normalization=False, delimiters=' \\\\t\\\\r\\\\n', comment='#')\n"
             "            .. function:: load(edge_list, as_undirected=True, normalization=False)\n"
             "            .. function:: load(weighted_edge_list, as_undirected=True, normalization=False)\n")
        .def("load", &Graph::load_edge_list, no_gil(), py::arg("edge_list"), py::arg("as_undirected") = true,
             py::arg("normalization") = false)
        .def("load", &Graph::load_weighted_edge_list, no_gil(), py::arg("weighted_edge_list"), py::arg("as_undirected") = true,
             py::arg("normalization") = false)
        .def("save", &Graph::save, no_gil(), py::arg("file_name"), py::arg("weighted") = true, py::arg("anonymous") = false,
             "save(file_name, weighted=True, anonymous=False)\n            Save the graph in edge-list format.\n")
        .def("__repr__", &Graph::info, no_gil());
    template_class<WordGraph, Graph>(graph, "WordGraph", "WordGraph_j",
                                     "WordGraph(index_type=dtype.uint32)\n        Normal graphs of word co-occurrences.\n\n"
                                     "        Parameters:\n            index_type (dtype): type of node indexes\n        ")
        .def(py::init<>(), no_gil())
        .def("load", &WordGraph::load_file_compact, no_gil(), py::arg("file_name"), py::arg("window") = 5, py::arg("min_count") = 5,
             py::arg("normalization") = false, py::arg("delimiters") = " \t\r\n", py::arg("comment") = "#",
             "load(file_name, window=5, min_count=5, normalization=False, delimiters=' \\\\t\\\\r\\\\n', comment='#')\n"
             "            Load a word graph from a corpus file.\n")
        .def("__repr__", &WordGraph::info, no_gil());

    // solvers
    auto solver = module.def_submodule("solver");
    py::class_<GraphSolverBase> base(solver, "_GraphSolver");
#define MEMBER(name) base.def_property_readonly(#name, [](GraphSolverBase &s) { return s.members().name; })
    MEMBER(num_partition); MEMBER(num_negative); MEMBER(negative_sample_exponent); MEMBER(negative_weight);
    MEMBER(num_epoch); MEMBER(episode_size); MEMBER(batch_size); MEMBER(augmentation_step); MEMBER(random_walk_length);
    MEMBER(random_walk_batch_size); MEMBER(shuffle_base); MEMBER(p); MEMBER(q); MEMBER(positive_reuse); MEMBER(log_frequency);
    MEMBER(num_worker); MEMBER(num_sampler); MEMBER(gpu_memory_limit); MEMBER(gpu_memory_cost);
#undef MEMBER
    base.def_property_readonly("resume", [](GraphSolverBase &s) { return s.members().resume != 0; })
        .def_property_readonly("model", [](GraphSolverBase &s) { return std::string(s.members().model); })
        .def_readonly("optimizer", &GraphSolverBase::optimizer)
        // beyond the reference: what its log prints as "[time] train: ... s" for the episode loop (util/time.h:28-60)
        .def_property_readonly("train_seconds", [](GraphSolverBase &s) { return s.members().train_seconds; })
        .def_property_readonly("batch_id", [](GraphSolverBase &s) { return s.members().batch_id; })
        .def_property_readonly("vertex_embeddings", [](GraphSolverBase &s) { return s.view(0); },
                               "Vertex node embeddings (2D numpy view).")
        .def_property_readonly("context_embeddings", [](GraphSolverBase &s) { return s.view(1); },
                               "Context node embeddings (2D numpy view).")
        .def("build",
             [](GraphSolverBase &s, py::object graph, const Optimizer &optimizer, int num_partition, int num_negative,
                int batch_size, int episode_size) {
                 const Graph &g = graph.cast<const Graph &>();
                 s.graph_keepalive = graph;  // the engine borrows the graph store until the next build (solver.h:289)
                 py::gil_scoped_release release;
                 s.build(g, optimizer, num_partition, num_negative, batch_size, episode_size);
             },
             py::arg("graph"), py::arg("optimizer") = Optimizer(kAuto), py::arg("num_partition") = kAuto,
             py::arg("num_negative") = 1, py::arg("batch_size") = 100000, py::arg("episode_size") = kAuto,
             "build(graph, optimizer=auto, num_partition=auto, num_negative=1, batch_size=100000, episode_size=auto)\n"
             "            Determine and allocate all resources for the solver.\n")
        .def("train", &GraphSolverBase::train, no_gil(), py::arg("model") = "LINE", py::arg("num_epoch") = 2000,
             py::arg("resume") = false, py::arg("augmentation_step") = kAuto, py::arg("random_walk_length") = 40,
             py::arg("random_walk_batch_size") = 100, py::arg("shuffle_base") = kAuto, py::arg("p") = 1, py::arg("q") = 1,
             py::arg("positive_reuse") = 1, py::arg("negative_sample_exponent") = 0.75, py::arg("negative_weight") = 5,
             py::arg("log_frequency") = 1000,
             "train(model='LINE', num_epoch=2000, resume=False, augmentation_step=auto, random_walk_length=40, "
             "random_walk_batch_size=100, shuffle_base=auto, p=1, q=1, positive_reuse=1, negative_sample_exponent=0.75, "
             "negative_weight=5, log_frequency=1000)\n            Train node embeddings.\n")
        .def("predict", &GraphSolverBase::predict, py::arg("samples"),
             "predict(samples)\n            Predict logits for samples.\n")
        .def("clear", &GraphSolverBase::clear, no_gil(), "clear()\n            Free CPU and GPU memory, except the embeddings on CPU.\n")
        .def("save_embeddings",
             [](GraphSolverBase &s, const std::string &file_name) {
                 check(gvx_solver_save_embeddings(s.handle, file_name.c_str()), "GraphSolver.save_embeddings");
             },
             no_gil(), py::arg("file_name"))
        .def("__repr__", &GraphSolverBase::info, no_gil());
    bind_solver<128>(solver);
    bind_solver<32>(solver);
    bind_solver<64>(solver);
    bind_solver<96>(solver);
    bind_solver<256>(solver);
    bind_solver<512>(solver);

    // interface
    py::enum_<DType> dtype(module, "dtype");
    dtype.value("uint32", DType::uint32).value("uint64", DType::uint64).value("float32", DType::float32).value("float64", DType::float64);
    py::dict dtype2name;  // typeid(T).name() under the Itanium ABI: the suffixes of the template instantiations
    dtype2name[py::cast(DType::uint32)] = "j", dtype2name[py::cast(DType::uint64)] = "m";
    dtype2name[py::cast(DType::float32)] = "f", dtype2name[py::cast(DType::float64)] = "d";
    module.attr("dtype2name") = dtype2name;

    // logging (glog severities, src/graphvite.cu:81-88)
    module.def("init_logging",
               [](int threshold, const std::string &dir, bool verbose) {
                   (void)dir, (void)verbose;  // messages go to stderr; files and source locations are glog features
                   gvx_set_logging(threshold, nullptr, nullptr);
               },
               no_gil(), py::arg("threshhold") = 0, py::arg("dir") = "", py::arg("verbose") = false);
    module.attr("INFO") = 0, module.attr("WARNING") = 1, module.attr("ERROR") = 2, module.attr("FATAL") = 3;

    auto io = module.def_submodule("io");
    io.def("size_string", size_string, no_gil(), py::arg("size"));
    io.def("yes_no", yes_no, no_gil(), py::arg("x"));
    io.def("block", block, no_gil(), py::arg("content"));
    io.def("header", header, no_gil(), py::arg("content"));

    module.attr("auto") = kAuto;
    module.def("KiB", [](size_t size) { return size << 10; }, no_gil(), py::arg("size"));
    module.def("MiB", [](size_t size) { return size << 20; }, no_gil(), py::arg("size"));
    module.def("GiB", [](size_t size) { return size << 30; }, no_gil(), py::arg("size"));
    module.attr("__version__") = kVersion;
}
