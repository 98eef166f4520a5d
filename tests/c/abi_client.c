/* A plain C host of the drop-in boundary (include/gvs.h + include/gvx.h; no Python, no C++, no torch): loads an edge
 * list, builds a GraphSolver, trains LINE, predicts, and checks that edges score above non-edges.
 *   gcc -std=c11 -O2 -I include tests/c/abi_client.c -L graphvite_amd -lgvk -Wl,-rpath,$PWD/graphvite_amd -lm -o abi_client
 *   ./abi_client <num_vertex> <num_edge>        exit 0: trained and verified; 3: no GPU (the library said so); 1: failure */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gvx.h"

static uint64_t state = 88172645463325252ull;
static uint32_t next_random(void) { /* xorshift64 */
    state ^= state << 13, state ^= state >> 7, state ^= state << 17;
    return (uint32_t)(state >> 32);
}

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != GVK_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gvk_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 5000, communities = 25;
    const size_t m = argc > 2 ? (size_t)atol(argv[2]) : 100000;
    uint32_t *u = malloc(m * 4), *v = malloc(m * 4);
    for (size_t e = 0; e < m; e++) { /* planted partition: 90 % of the edges stay inside a community */
        u[e] = next_random() % n;
        const uint32_t size = n / communities, base = u[e] / size * size;
        v[e] = next_random() % 10 ? base + next_random() % size : next_random() % n;
        if (v[e] >= n) v[e] = n - 1;
    }
    gvs_graph *graph = gvs_graph_create();
    CHECK(gvs_graph_load_labels(graph, u, v, NULL, m, /*as_undirected*/ 1, /*normalization*/ 0));
    printf("graph: %u vertices, %llu edges\n", gvs_graph_num_vertex(graph), (unsigned long long)gvs_graph_num_edge(graph));

    gvx_set_logging(/*WARNING*/ 2, NULL, NULL);
    gvx_solver *solver = gvx_solver_create(128, NULL, 0, /*samplers*/ 2, GVX_AUTO);
    if (!solver) {
        fprintf(stderr, "gvx_solver_create: %s\n", gvk_last_error());
        return strstr(gvk_last_error(), "No GPU") ? 3 : 1;
    }
    gvx_optimizer sgd = {GVK_SGD, 0.025f, 0.005f, 0, 0, 0, /*linear*/ 1, NULL, NULL};
    CHECK(gvx_solver_build(solver, graph, &sgd, GVX_AUTO, 1, 10000, 10));
    gvx_train_config config = {"LINE", 300, 0, 1, 40, 100, GVX_AUTO, 1, 1, 1, 0.75f, 5, 1 << 30};
    CHECK(gvx_solver_train(solver, &config));
    gvx_solver_members members;
    CHECK(gvx_solver_get(solver, &members));
    printf("trained %llu batches of %d in %.2f s (model %s, %d partition(s), episode %d)\n",
           (unsigned long long)members.batch_id, members.batch_size, members.train_seconds, members.model,
           members.num_partition, members.episode_size);

    /* held-in edges against random pairs: mean logit of edges must be clearly larger */
    enum { K = 2000 };
    int64_t *samples = malloc(sizeof(int64_t) * 4 * K);
    float *logits = malloc(sizeof(float) * 2 * K);
    const uint32_t *edges = gvs_graph_edges(graph);
    for (int i = 0; i < K; i++) {
        const size_t e = next_random() % gvs_graph_num_directed_edge(graph);
        samples[2 * i] = edges[2 * e], samples[2 * i + 1] = edges[2 * e + 1];
        samples[2 * (K + i)] = next_random() % n, samples[2 * (K + i) + 1] = next_random() % n;
    }
    CHECK(gvx_solver_predict(solver, samples, 2 * K, logits));
    double edge = 0, random = 0;
    for (int i = 0; i < K; i++) edge += logits[i], random += logits[K + i];
    uint64_t rows = 0;
    const float *vertex = gvx_solver_embeddings(solver, 0, &rows);
    printf("mean logit: edges %.3f, random pairs %.3f; vertex[0][0] = %g of %llu rows\n", edge / K, random / K, vertex[0],
           (unsigned long long)rows);
    gvx_solver_destroy(solver);
    gvs_graph_destroy(graph);
    return edge / K > random / K + 0.5 ? 0 : 1;
}
