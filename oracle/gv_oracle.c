/* TEST INFRASTRUCTURE ONLY — CPU oracle for the node-embedding hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (graphvite_amd/) never links, imports
 * or falls back to anything in oracle/.
 *
 * Every function restates, in plain sequential C, one piece of the reference
 * (DeepGraphLearning/graphvite v0.2.2, paths relative to /root/reference):
 *
 *   gvo_sigmoid            include/util/math.h:30-33
 *   gvo_lr                 include/core/optimizer.h:77-79,132-134
 *   gvo_train              include/instance/gpu/graph.cuh:54-94 (SGD), :122-166 (1 moment),
 *                          :196-241 (Adam) + instance/model/graph.h:40-85 + core/optimizer.h:161-210
 *   gvo_predict            include/instance/gpu/graph.cuh:265-278
 *   gvo_alias_build        include/base/alias_table.cuh:84-128
 *   gvo_alias_sample       include/base/alias_table.cuh:148-152
 *   gvo_alias_sample_gpu   include/base/alias_table.cuh:174-182 (gpu::Sample narrows to Float first)
 *   gvo_partition          include/core/solver.h:873-887 (+ locations :399-410)
 *   gvo_schedule           include/core/solver.h:519-575 (non-tied branch; GraphSolver never ties)
 *   gvo_negative_weights   include/core/solver.h:1264-1278
 *   gvo_sample_edges       include/core/solver.h:1012-1055
 *   gvo_sample_walks       include/instance/graph.cuh:376-450 (biased = 0), :298-373 (biased = 1)
 *   gvo_edge_edge_weights  include/instance/graph.cuh:656-677
 *
 * Two functions restate THIS repo's documented ABI rather than the reference (the
 * reference takes its uniforms from cuRAND XORWOW, which is unavailable here and pinned
 * by nothing — SURVEY.md §8c): gvo_philox4x32 (Philox4x32-10, Salmon et al. SC'11,
 * checked against the Random123 known-answer vectors in tests) and gvo_negative_draw /
 * gvo_host_uniforms (include/gvk.h "RNG contract").
 *
 * Parity status: the reference ships no tests or golden vectors, so this
 * oracle is pinned against the reference's own code compiled for the host, and against
 * fixtures generated from those builds (tests/golden/, script committed):
 *   - arithmetic (model, optimizers, schedule, sigmoid): oracle/ref_harness.cpp -> _ref/libgvref.so;
 *   - alias table build / sample: oracle/ref_alias_harness.cpp -> _ref/libgvref_alias.so;
 *   - partition, schedule, edge sampler (pools record for record), the alias tables of the
 *     walk samplers: oracle/ref_solver_harness.cpp -> _ref/libgvref_solver.so (the
 *     reference's solver front end over an emulated CUDA runtime / cuRAND, ref_stubs/).
 * The walk samplers consume their uniforms in lockstep order (gvo_sample_walks, what the product's samplers do) or
 * walk by walk (gvo_sample_walks_reference_order): the latter reproduces the pools of the reference's own
 * sample_random_walk / sample_biased_random_walk record for record, the former is the same walk logic with the
 * i.i.d. uniforms taken in another order (and is what the product is pinned against, bit for bit).
 * gvo_sample_pairs / gvo_sample_walks_device restate device samplers the reference lacks.
 * gvo_class_table_build / gvo_negative_draw_class restate the negative sampler by weight classes (include/gvk.h): the
 * distribution of the reference's one-slot-per-row table (solver.h:1264-1278) drawn as class, then row; the tests
 * reconstruct every row's probability from the class table and compare it with weight / sum of weights.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GVO_EPS 1e-15f /* util/common.h:28 */

enum { GVO_SGD = 0, GVO_MOMENTUM, GVO_ADAGRAD, GVO_RMSPROP, GVO_ADAM }; /* optimizer.h:27-34 */

float gvo_sigmoid(float x) { return x > 0 ? 1 / (1 + expf(-x)) : expf(x) / (expf(x) + 1); }

float gvo_lr(float init_lr, int linear, int batch_id, int num_batch) {
    float s = 1;
    if (linear) {
        s = 1 - (float)batch_id / num_batch;
        if (s < 1e-4f) s = 1e-4f;
    }
    return init_lr * s;
}

/* hp = {momentum | alpha | beta1, beta2, epsilon} (optimizer.h:110-122) */
static float gvo_update(int type, float lr, float wd, const float *hp, float param, float grad, float weight,
                        float *m1, float *m2) {
    float reg = weight * (grad + wd * param);
    switch (type) {
        case GVO_SGD: return lr * weight * (grad + wd * param); /* optimizer.h:161-164 */
        case GVO_MOMENTUM:                                      /* :170-175 */
            *m1 = hp[0] * *m1 + (1 - hp[0]) * reg;
            return lr * *m1;
        case GVO_ADAGRAD: /* :181-186 */
            *m1 += reg * reg;
            return lr * reg / (sqrtf(*m1) + hp[2]);
        case GVO_RMSPROP: /* :192-197 */
            *m1 = hp[0] * *m1 + (1 - hp[0]) * reg * reg;
            return lr * reg / sqrtf(*m1 + hp[2]);
        default: /* Adam :203-210 */
            *m1 = hp[0] * *m1 + (1 - hp[0]) * reg;
            *m2 = hp[1] * *m2 + (1 - hp[1]) * reg * reg;
            return lr * *m1 / (sqrtf(*m2) + hp[2]);
    }
}

/* One batch, samples processed strictly in order. batch = {tail, head} u32 records. */
int gvo_train(int dim, int type, float *vertex, float *context, float *vm1, float *cm1, float *vm2, float *cm2,
              const uint32_t *batch, const uint32_t *negatives, float *loss, int batch_size, int k, float lr,
              float wd, float negative_weight, const float *hp) {
    float *buf = (float *)malloc(sizeof(float) * dim);
    float dummy1 = 0, dummy2 = 0;
    if (!buf) return -1;
    for (int s = 0; s < batch_size; s++) {
        size_t head = batch[2 * s + 1];
        float *v = vertex + head * dim;
        memcpy(buf, v, sizeof(float) * dim);
        float sample_loss = 0;
        for (int j = 0; j <= k; j++) {
            size_t tail = j < k ? negatives[(size_t)s * k + j] : batch[2 * s];
            int label = j == k;
            float *c = context + tail * dim;
            float logit = 0;
            for (int i = 0; i < dim; i++) logit += buf[i] * c[i];
            float prob = gvo_sigmoid(logit);
            float gradient, weight;
            if (label) {
                gradient = prob - 1;
                weight = 1;
                sample_loss += weight * -logf(prob + GVO_EPS);
            } else {
                gradient = prob;
                weight = negative_weight;
                sample_loss += weight * -logf(1 - prob + GVO_EPS);
            }
            for (int i = 0; i < dim; i++) {
                float vi = buf[i], ci = c[i];
                float *pvm1 = vm1 ? vm1 + head * dim + i : &dummy1, *pcm1 = cm1 ? cm1 + tail * dim + i : &dummy1;
                float *pvm2 = vm2 ? vm2 + head * dim + i : &dummy2, *pcm2 = cm2 ? cm2 + tail * dim + i : &dummy2;
                buf[i] -= gvo_update(type, lr, wd, hp, vi, gradient * ci, weight, pvm1, pvm2);
                c[i] -= gvo_update(type, lr, wd, hp, ci, gradient * vi, weight, pcm1, pcm2);
            }
        }
        loss[s] = sample_loss / (1 + k * negative_weight);
        memcpy(v, buf, sizeof(float) * dim);
    }
    free(buf);
    return 0;
}

/* ---- hub rows trained by chains (the product's gvk_hot_build / gvk_train_episode_hot, include/gvk.h; no counterpart in
 * the reference — its kernel trains every sample the same way, gpu/graph.cuh:36-95) ----------------------------------------
 * The work lists of one batch in sample order: chain c < kv is head row c — per sample with that head its k negatives
 * (label 0) then its tail (label 1, bit 31); chain kv + r is context row r — the head of every sample r is the tail (label
 * 1) or a negative (label 0) of.  chain_start [kv + kc + 1], entries [2 (k + 1) batch_size].  Returns the number of entries. */
size_t gvo_hot_lists(const uint32_t *batch, const uint32_t *negatives, int batch_size, int k, uint32_t kv, uint32_t kc,
                     uint32_t *chain_start, uint32_t *entries) {
    const uint32_t chains = kv + kc;
    uint32_t *cursor = (uint32_t *)calloc(chains + 1, sizeof(uint32_t));
    if (!cursor) return 0;
    for (int s = 0; s < batch_size; s++) {
        const uint32_t tail = batch[2 * s], head = batch[2 * s + 1];
        if (head < kv) cursor[head] += (uint32_t)(k + 1);
        if (tail < kc) cursor[kv + tail]++;
        for (int j = 0; j < k; j++)
            if (negatives[(size_t)s * k + j] < kc) cursor[kv + negatives[(size_t)s * k + j]]++;
    }
    uint32_t running = 0;
    for (uint32_t c = 0; c < chains; c++) {
        const uint32_t n = cursor[c];
        chain_start[c] = cursor[c] = running;
        running += n;
    }
    chain_start[chains] = running;
    for (int s = 0; s < batch_size; s++) {
        const uint32_t tail = batch[2 * s], head = batch[2 * s + 1];
        if (head < kv) {
            for (int j = 0; j < k; j++) entries[cursor[head]++] = negatives[(size_t)s * k + j];
            entries[cursor[head]++] = tail | 0x80000000u;
        }
        for (int j = 0; j < k; j++) {
            const uint32_t n = negatives[(size_t)s * k + j];
            if (n < kc) entries[cursor[kv + n]++] = head;
        }
        if (tail < kc) entries[cursor[kv + tail]++] = head | 0x80000000u;
    }
    free(cursor);
    return running;
}

/* Executor-simulator experiments (gvo_set_hub_snapshot): when set, a chain reads a partner that is a hub row — id below the count —
 * from these copies instead of the tables: "what if the chains of several units read hub partners as the GROUP of units found
 * them" (one launch for several units).  NULL (default): from the tables the caller passes. */
static const float *gvo_snapshot_vertex = NULL, *gvo_snapshot_context = NULL;
static uint32_t gvo_snapshot_kv = 0, gvo_snapshot_kc = 0;
void gvo_set_hub_snapshot(const float *vertex_hub_rows, uint32_t kv, const float *context_hub_rows, uint32_t kc) {
    gvo_snapshot_vertex = vertex_hub_rows, gvo_snapshot_kv = kv, gvo_snapshot_context = context_hub_rows, gvo_snapshot_kc = kc;
}

/* One chain task: entries [begin, end) applied one after the other to a copy of the own row; partner rows are only read. */
static void gvo_chain(int dim, float *own, const float *partner, const uint32_t *entries, uint32_t begin, uint32_t end,
                      float lr, float wd, float negative_weight, const float *hub_snapshot, uint32_t hub_count) {
    float dummy1 = 0, dummy2 = 0;
    for (uint32_t p = begin; p < end; p++) {
        const size_t id = entries[p] & 0x7fffffffu;
        const float *c = hub_snapshot && id < hub_count ? hub_snapshot + id * dim : partner + id * dim;
        const int label = entries[p] >> 31;
        float logit = 0;
        for (int i = 0; i < dim; i++) logit += own[i] * c[i];
        const float prob = gvo_sigmoid(logit);
        const float gradient = label ? prob - 1 : prob, weight = label ? 1 : negative_weight;
        for (int i = 0; i < dim; i++) own[i] -= gvo_update(0, lr, wd, NULL, own[i], gradient * c[i], weight, &dummy1, &dummy2);
    }
}

/* The chains [first_chain, last_chain) of one unit (SGD): each over its entries in list order on its own row, which it reads
 * from `own_table` and leaves there; partner rows are read from `partner_vertex` / `partner_context` (the caller passes the
 * tables as the unit found them).  A chain longer than cap entries is trained by up to max_tasks tasks side by side, as one
 * workgroup of the product trains it (train_long_chains, graphvite_amd/csrc/gvk_chains.hip): task g owns the consecutive
 * entries [first + g per, first + (g + 1) per), per = cap or — past max_tasks tasks of cap entries (0 = no limit) —
 * ceil(n / max_tasks).  The tasks work in ROUNDS of gvo_round_steps entries each (0: one round, the product's form until
 * round 4): in round r every task applies its entries [r steps, (r + 1) steps) in sequence to the row as the round found it
 * under the weight decay of the tasks before it in the round, and the round's end states are composed (below) into the row the
 * next round starts from — so no more than max_tasks x gvo_round_steps entries ever work side by side from the same state,
 * however many updates a hub row meets in a unit.  (The order in which a chain's entries are applied is then round by round,
 * task by task: any order of a unit's samples is a sequential order.)
 * gvo_set_long_task(t > 0) (executor-simulator experiments): tasks of t entries, as many as it takes, one round; t = 1 is
 * "every entry of a long chain on its own" (measured in round 5: profiles/r5/experiments/r5_entries_side_by_side.txt). */
static uint32_t gvo_long_task = 0, gvo_round_steps = 0;
void gvo_set_long_task(uint32_t entries) { gvo_long_task = entries; }
void gvo_set_round_steps(uint32_t steps) { gvo_round_steps = steps; }

static int gvo_hot_chains(int dim, float *vertex, float *context, const float *partner_vertex, const float *partner_context,
                          float lr, float wd, float negative_weight, uint32_t kv, const uint32_t *chain_start,
                          const uint32_t *entries, uint32_t cap, uint32_t max_tasks, int k, uint32_t first_chain,
                          uint32_t last_chain) {
    /* the tasks' sum in double: after * own and total * row are nearly equal numbers (exact as double products), and a long chain
     * adds thousands of their differences */
    float *own = (float *)malloc(sizeof(float) * dim);
    double *sum = (double *)malloc(sizeof(double) * dim);
    if (!own || !sum) return -1;
    /* the decay factors of an entry in double: 1 - lr wd rounded to float is off by 3e-8, i.e. by 2e-4 of its distance from 1,
     * which a chain of a thousand entries raises to 3e-5 of the row */
    const double decay_positive = 1.0 - (double)lr * (double)wd, decay_negative = 1.0 - (double)lr * (double)negative_weight * (double)wd;
    for (uint32_t chain = first_chain; chain < last_chain; chain++) {
        float *row = chain < kv ? vertex + (size_t)chain * dim : context + (size_t)(chain - kv) * dim;
        const float *partner = chain < kv ? partner_context : partner_vertex;
        const float *snapshot = chain < kv ? gvo_snapshot_context : gvo_snapshot_vertex;  /* a head row's partners are context rows */
        const uint32_t snapshot_count = chain < kv ? gvo_snapshot_kc : gvo_snapshot_kv;
        const uint32_t first = chain_start[chain], last = chain_start[chain + 1], n = last - first;
        if (n <= cap) {
            gvo_chain(dim, row, partner, entries, first, last, lr, wd, negative_weight, snapshot, snapshot_count);
            continue;
        }
        uint32_t per = gvo_long_task ? gvo_long_task : cap;
        if (!gvo_long_task && max_tasks && (uint64_t)per * max_tasks < n) per = (n + max_tasks - 1) / max_tasks;
        (void)k;
        const uint32_t steps = gvo_round_steps && !gvo_long_task && per > gvo_round_steps ? gvo_round_steps : per;
        /* tasks: weight decay composes in closed form (a factor per entry that depends on its label only), so every task
         * starts from the row as the decay of the entries before it leaves it, and its end state is carried through the
         * decay of the entries after it: row <- total row + sum over tasks (after end - total row) */
        for (uint32_t done = 0; done < per; done += steps) {  /* one round: the segment [done, done + steps) of every task's range */
            uint32_t positives_all = 0, entries_all = 0;
            for (uint32_t begin = first; begin < last; begin += per) {
                const uint32_t end = last - begin > per ? begin + per : last;
                for (uint32_t p = begin + done; p < end && p < begin + done + steps; p++) positives_all += entries[p] >> 31, entries_all++;
            }
            if (!entries_all) break;
            const float total = (float)(pow(decay_positive, (double)positives_all) * pow(decay_negative, (double)(entries_all - positives_all)));
            memset(sum, 0, sizeof(double) * dim);
            uint32_t positives_before = 0, entries_before = 0;
            for (uint32_t begin = first; begin < last; begin += per) {
                const uint32_t end = last - begin > per ? begin + per : last;
                const uint32_t from = begin + done < end ? begin + done : end, to = end - from > steps ? from + steps : end;
                if (from == to) continue;  /* a task whose range ended in an earlier round */
                uint32_t positives_inside = 0;
                for (uint32_t p = from; p < to; p++) positives_inside += entries[p] >> 31;
                const uint32_t positives_after = positives_all - positives_before - positives_inside;
                const uint32_t entries_after = entries_all - entries_before - (to - from);
                const float before = (float)(pow(decay_positive, (double)positives_before) * pow(decay_negative, (double)(entries_before - positives_before)));
                const float after = (float)(pow(decay_positive, (double)positives_after) * pow(decay_negative, (double)(entries_after - positives_after)));
                for (int i = 0; i < dim; i++) own[i] = before * row[i];
                gvo_chain(dim, own, partner, entries, from, to, lr, wd, negative_weight, snapshot, snapshot_count);
                for (int i = 0; i < dim; i++) sum[i] += (double)after * (double)own[i] - (double)total * (double)row[i];
                positives_before += positives_inside, entries_before += to - from;
            }
            for (int i = 0; i < dim; i++) row[i] = (float)((double)total * (double)row[i] + sum[i]);
        }
    }
    free(own), free(sum);
    return 0;
}

/* The chains of one unit, BOTH families from the tables as the unit finds them: a chain reads its own row and its partners,
 * hub rows included, from that state, so a sample between two hub rows updates both from their old values, as the reference
 * does (model/graph.h:47-58).  In place: the hub rows of both tables end as the chains leave them. */
int gvo_hot_unit_chains(int dim, float *vertex, float *context, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                        const uint32_t *chain_start, const uint32_t *entries, uint32_t cap, uint32_t max_tasks, int k) {
    /* the context-row chains read the head table, whose hub rows are still untouched; the head-row chains then read the hub
     * rows of the context table from a copy taken before.  Rows that are not hub rows are the same in both. */
    float *c0 = (float *)malloc(sizeof(float) * dim * (kc ? kc : 1)), *seen = (float *)malloc(sizeof(float) * dim * (kc ? kc : 1));
    if (!c0 || !seen) return -1;
    memcpy(c0, context, sizeof(float) * dim * kc);
    int rc = gvo_hot_chains(dim, vertex, context, vertex, context, lr, wd, negative_weight, kv, chain_start, entries, cap, max_tasks, k, kv, kv + kc);
    if (!rc && kv) {
        memcpy(seen, context, sizeof(float) * dim * kc);
        memcpy(context, c0, sizeof(float) * dim * kc);
        rc = gvo_hot_chains(dim, vertex, context, vertex, context, lr, wd, negative_weight, kv, chain_start, entries, cap, max_tasks, k, 0, kv);
        memcpy(context, seen, sizeof(float) * dim * kc);
    }
    free(c0), free(seen);
    return rc;
}

/* The pairs of one unit: every sample in order as gvo_train, except that hub rows (the first kv / kc rows: in the tables as
 * the unit's chains left them) are read and never written.  before_vertex / before_context non-NULL (lerp): the hub rows as
 * those chains FOUND them — a sample then reads a hub row where its chain was when it met the sample, approximated by the
 * straight line from the row before the chains to the row after them: sample s of n reads before + (s + 1/2) / n (now - before).
 *
 * gvo_set_pairs_concurrent(1) (executor simulator, GVH_PAIRS=concurrent): the samples of the unit run as one launch of the
 * product runs them — every sample reads its rows as the UNIT found them (not as the samples before it left them), and of two
 * samples that write the same row the later one's row stays: what Hogwild inside a launch does to rows that are not hub rows. */
static int gvo_pairs_concurrent = 0;
void gvo_set_pairs_concurrent(int on) { gvo_pairs_concurrent = on != 0; }
/* experiment: with before_vertex / before_context given, every sample reads hub rows at this fixed place of the chains' way (0 = as the unit
 * found them) instead of at its own place in the unit; negative: off */
static float gvo_pairs_at = -1;
void gvo_set_pairs_at(float at) { gvo_pairs_at = at; }

typedef struct {
    uint64_t *keys;   /* (table << 32 | row) + 1; 0 = empty */
    uint32_t *slots;  /* index of the row's saved original */
    float *rows;      /* saved originals, dim floats each */
    size_t mask, used;
} gvo_originals;

static const float *gvo_original_of(const gvo_originals *o, int dim, int table, size_t row) {
    const uint64_t key = ((uint64_t)table << 32 | (uint64_t)row) + 1;
    for (size_t h = (size_t)(key * 0x9E3779B97F4A7C15ull) & o->mask;; h = (h + 1) & o->mask) {
        if (o->keys[h] == key) return o->rows + (size_t)o->slots[h] * dim;
        if (o->keys[h] == 0) return NULL;
    }
}

static void gvo_save_original(gvo_originals *o, int dim, int table, size_t row, const float *current) {
    const uint64_t key = ((uint64_t)table << 32 | (uint64_t)row) + 1;
    for (size_t h = (size_t)(key * 0x9E3779B97F4A7C15ull) & o->mask;; h = (h + 1) & o->mask) {
        if (o->keys[h] == key) return;
        if (o->keys[h] == 0) {
            o->keys[h] = key, o->slots[h] = (uint32_t)o->used;
            memcpy(o->rows + o->used * dim, current, sizeof(float) * dim);
            o->used++;
            return;
        }
    }
}

int gvo_train_pairs_hot(int dim, float *vertex, float *context, const uint32_t *batch, const uint32_t *negatives, float *loss,
                        int batch_size, int k, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                        const float *before_vertex, const float *before_context) {
    float *buf = (float *)malloc(sizeof(float) * dim), *hub = (float *)malloc(sizeof(float) * dim);
    float *own = (float *)malloc(sizeof(float) * dim);
    float dummy1 = 0, dummy2 = 0;
    if (!buf || !hub || !own) return -1;
    gvo_originals seen = {NULL, NULL, NULL, 0, 0};
    if (gvo_pairs_concurrent) {
        size_t capacity = 1;
        while (capacity < 4 * (size_t)(k + 2) * (size_t)(batch_size > 0 ? batch_size : 1)) capacity <<= 1;
        seen.keys = (uint64_t *)calloc(capacity, sizeof(uint64_t)), seen.slots = (uint32_t *)malloc(capacity * sizeof(uint32_t));
        seen.rows = (float *)malloc(sizeof(float) * dim * (size_t)(k + 2) * (size_t)(batch_size > 0 ? batch_size : 1));
        seen.mask = capacity - 1;
        if (!seen.keys || !seen.slots || !seen.rows) return -1;
    }
    for (int s = 0; s < batch_size; s++) {
        const float at = gvo_pairs_at >= 0 ? gvo_pairs_at : (s + 0.5f) / batch_size;
        const size_t head = batch[2 * s + 1];
        if (head < kv && before_vertex) {
            for (int i = 0; i < dim; i++) buf[i] = before_vertex[head * dim + i] + at * (vertex[head * dim + i] - before_vertex[head * dim + i]);
        } else {
            const float *start = gvo_pairs_concurrent && head >= kv ? gvo_original_of(&seen, dim, 0, head) : NULL;
            memcpy(buf, start ? start : vertex + head * dim, sizeof(float) * dim);
        }
        float sample_loss = 0;
        size_t last = 0;
        int have = 0;
        for (int j = 0; j <= k; j++) {
            const size_t tail = j < k ? negatives[(size_t)s * k + j] : batch[2 * s];
            const int label = j == k;
            float *c = context + tail * dim;
            const float *from = c;
            if (tail < kc) {
                /* a hub row is not written here, but a sample whose consecutive targets are the same row sees its own
                 * update (the kernel carries the updated registers over), as it does for any other row */
                if (have && tail == last) {
                    from = own;
                } else if (before_context) {
                    for (int i = 0; i < dim; i++) hub[i] = before_context[tail * dim + i] + at * (c[i] - before_context[tail * dim + i]);
                    from = hub;
                }
            } else if (gvo_pairs_concurrent) {
                if (have && tail == last) {
                    from = own;  /* its own update, carried in registers */
                } else {
                    const float *start = gvo_original_of(&seen, dim, 1, tail);
                    if (start) from = start;
                }
            }
            float logit = 0;
            for (int i = 0; i < dim; i++) logit += buf[i] * from[i];
            const float prob = gvo_sigmoid(logit);
            float gradient, weight;
            if (label) {
                gradient = prob - 1, weight = 1;
                sample_loss += weight * -logf(prob + GVO_EPS);
            } else {
                gradient = prob, weight = negative_weight;
                sample_loss += weight * -logf(1 - prob + GVO_EPS);
            }
            if (gvo_pairs_concurrent && tail >= kc) gvo_save_original(&seen, dim, 1, tail, c);
            for (int i = 0; i < dim; i++) {
                const float vi = buf[i], ci = from[i];
                buf[i] -= gvo_update(0, lr, wd, NULL, vi, gradient * ci, weight, &dummy1, &dummy2);
                const float cn = ci - gvo_update(0, lr, wd, NULL, ci, gradient * vi, weight, &dummy1, &dummy2);
                if (tail >= kc) c[i] = cn;
                own[i] = cn;
            }
            last = tail, have = 1;
        }
        loss[s] = sample_loss / (1 + k * negative_weight);
        if (head >= kv) {
            if (gvo_pairs_concurrent) gvo_save_original(&seen, dim, 0, head, vertex + head * dim);
            memcpy(vertex + head * dim, buf, sizeof(float) * dim);
        }
    }
    free(buf), free(hub), free(own);
    free(seen.keys), free(seen.slots), free(seen.rows);
    return 0;
}

/* One unit (a batch or a part of one) in the product's serialized form (gvk_train_episode_hot with GVK_HOT_SERIALIZED): the
 * unit's chains (gvo_hot_unit_chains), then its pairs (gvo_train_pairs_hot; `lerp`: hub rows read along the chains' way). */
int gvo_train_hot(int dim, float *vertex, float *context, const uint32_t *batch, const uint32_t *negatives, float *loss,
                  int batch_size, int k, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                  const uint32_t *chain_start, const uint32_t *entries, uint32_t cap, uint32_t max_tasks, int lerp) {
    float *v0 = NULL, *c0 = NULL;
    if (lerp) {
        v0 = (float *)malloc(sizeof(float) * dim * (kv ? kv : 1)), c0 = (float *)malloc(sizeof(float) * dim * (kc ? kc : 1));
        if (!v0 || !c0) return -1;
        memcpy(v0, vertex, sizeof(float) * dim * kv);
        memcpy(c0, context, sizeof(float) * dim * kc);
    }
    int rc = gvo_hot_unit_chains(dim, vertex, context, lr, wd, negative_weight, kv, kc, chain_start, entries, cap, max_tasks, k);
    if (!rc)
        rc = gvo_train_pairs_hot(dim, vertex, context, batch, negatives, loss, batch_size, k, lr, wd, negative_weight, kv, kc, v0, c0);
    free(v0), free(c0);
    return rc;
}

/* One unit in the serialized form of the MOMENT optimizers' chains (gvk_train_episode_hot with GVK_HOT_SERIALIZED and Momentum /
 * AdaGrad / RMSprop / Adam: train_moment_chains, graphvite_amd/csrc/gvk_chains.hip).  Their update does not compose, so every chain
 * is ONE sequential task: the hub row and its moment rows (optimizer.h:170-210 on the own row; gpu/graph.cuh:104-242 is the loop
 * this splits by row), the entries in list order, partner rows — hub rows included — as the unit found them.  Then the unit's
 * pairs: every sample in order as gvo_train, but a hub row and its moment rows are read (a sample's own steps move its copies)
 * and never written. */
/* Executor-simulator experiment (round 6): 1 = the pairs of a unit read a hub row as the unit FOUND it (before its chains) instead of as
 * its chains left it — a sample then computes its partner's update from the hub row without the hub's own step for that very sample. */
static int gvo_pairs_read_before = 0;
void gvo_set_pairs_read_before(int on) { gvo_pairs_read_before = on; }  /* 2 (experiment): on the straight line from before to after, at the sample's place in the unit */

int gvo_train_hot_moments(int dim, int type, float *vertex, float *context, float *vm1, float *cm1, float *vm2, float *cm2,
                          const uint32_t *batch, const uint32_t *negatives, float *loss, int batch_size, int k, float lr, float wd,
                          float negative_weight, const float *hp, uint32_t kv, uint32_t kc, const uint32_t *chain_start,
                          const uint32_t *entries) {
    float *v0 = (float *)malloc(sizeof(float) * dim * (kv ? kv : 1)), *c0 = (float *)malloc(sizeof(float) * dim * (kc ? kc : 1));
    float *buf = (float *)malloc(sizeof(float) * dim * 6);
    float dummy = 0;
    if (!v0 || !c0 || !buf) return -1;
    memcpy(v0, vertex, sizeof(float) * dim * kv);
    memcpy(c0, context, sizeof(float) * dim * kc);
    for (uint32_t chain = 0; chain < kv + kc; chain++) {
        const int is_vertex = chain < kv;
        const size_t row = is_vertex ? chain : chain - kv;
        float *own = (is_vertex ? vertex : context) + row * dim;
        float *m1 = (is_vertex ? vm1 : cm1) + row * dim, *m2 = type == GVO_ADAM ? (is_vertex ? vm2 : cm2) + row * dim : NULL;
        for (uint32_t p = chain_start[chain]; p < chain_start[chain + 1]; p++) {
            const size_t id = entries[p] & 0x7fffffffu;
            const int label = entries[p] >> 31;
            const float *c = is_vertex ? (id < kc ? c0 + id * dim : context + id * dim) : (id < kv ? v0 + id * dim : vertex + id * dim);
            float logit = 0;
            for (int i = 0; i < dim; i++) logit += own[i] * c[i];
            const float prob = gvo_sigmoid(logit);
            const float gradient = label ? prob - 1 : prob, weight = label ? 1 : negative_weight;
            for (int i = 0; i < dim; i++) own[i] -= gvo_update(type, lr, wd, hp, own[i], gradient * c[i], weight, m1 + i, m2 ? m2 + i : &dummy);
        }
    }
    /* the pairs */
    float *v = buf, *tv1 = buf + dim, *tv2 = buf + 2 * dim, *hub = buf + 3 * dim, *h1 = buf + 4 * dim, *h2 = buf + 5 * dim;
    for (int s = 0; s < batch_size; s++) {
        const size_t head = batch[2 * s + 1];
        const int hub_head = head < kv;
        memcpy(v, hub_head && gvo_pairs_read_before ? v0 + head * dim : vertex + head * dim, sizeof(float) * dim);
        if (hub_head && gvo_pairs_read_before == 2)
            for (int i = 0; i < dim; i++) v[i] += (s + 0.5f) / batch_size * (vertex[head * dim + i] - v[i]);
        float *pm1 = vm1 + head * dim, *pm2 = type == GVO_ADAM ? vm2 + head * dim : NULL;
        if (hub_head) {  /* the hub head's moment rows: copies */
            memcpy(tv1, pm1, sizeof(float) * dim), pm1 = tv1;
            if (pm2) memcpy(tv2, pm2, sizeof(float) * dim), pm2 = tv2;
        }
        float sample_loss = 0;
        size_t last = 0;
        int have = 0;
        for (int j = 0; j <= k; j++) {
            const size_t tail = j < k ? negatives[(size_t)s * k + j] : batch[2 * s];
            const int label = j == k;
            float *c = context + tail * dim, *q1 = cm1 + tail * dim, *q2 = type == GVO_ADAM ? cm2 + tail * dim : NULL;
            if (tail < kc) {  /* a hub target: copies, carried over when the next target is the same row */
                if (!(have && tail == last)) {
                    memcpy(hub, gvo_pairs_read_before ? c0 + tail * dim : c, sizeof(float) * dim), memcpy(h1, q1, sizeof(float) * dim);
                    if (gvo_pairs_read_before == 2)
                        for (int i = 0; i < dim; i++) hub[i] += (s + 0.5f) / batch_size * (c[i] - hub[i]);
                    if (q2) memcpy(h2, q2, sizeof(float) * dim);
                }
                c = hub, q1 = h1, q2 = q2 ? h2 : NULL;
            }
            float logit = 0;
            for (int i = 0; i < dim; i++) logit += v[i] * c[i];
            const float prob = gvo_sigmoid(logit);
            float gradient, weight;
            if (label) {
                gradient = prob - 1, weight = 1;
                sample_loss += weight * -logf(prob + GVO_EPS);
            } else {
                gradient = prob, weight = negative_weight;
                sample_loss += weight * -logf(1 - prob + GVO_EPS);
            }
            for (int i = 0; i < dim; i++) {
                const float vi = v[i], ci = c[i];
                v[i] -= gvo_update(type, lr, wd, hp, vi, gradient * ci, weight, pm1 + i, pm2 ? pm2 + i : &dummy);
                c[i] -= gvo_update(type, lr, wd, hp, ci, gradient * vi, weight, q1 + i, q2 ? q2 + i : &dummy);
            }
            last = tail, have = 1;
        }
        loss[s] = sample_loss / (1 + k * negative_weight);
        if (!hub_head) memcpy(vertex + head * dim, v, sizeof(float) * dim);
    }
    free(v0), free(c0), free(buf);
    return 0;
}

void gvo_predict(int dim, const float *vertex, const float *context, const uint32_t *batch, float *logits,
                 int batch_size) {
    for (int s = 0; s < batch_size; s++) {
        const float *v = vertex + (size_t)batch[2 * s + 1] * dim, *c = context + (size_t)batch[2 * s] * dim;
        float logit = 0;
        for (int i = 0; i < dim; i++) logit += v[i] * c[i];
        logits[s] = logit;
    }
}

/* ---- alias table ------------------------------------------------------------------ */

/* index_bytes = 4 (uint32 alias) or 8 (uint64 alias; the edge table uses size_t, solver.h:123) */
int gvo_alias_build(const float *w, size_t n, float *prob, void *alias, int index_bytes) {
    if (n == 0) return -1;
    uint32_t *a32 = (uint32_t *)alias;
    uint64_t *a64 = (uint64_t *)alias;
    size_t *large = (size_t *)malloc(sizeof(size_t) * n), *little = (size_t *)malloc(sizeof(size_t) * n);
    if (!large || !little) return -1;
    /* FIFO queues (std::queue in the reference). An index enters `little` at most once (at the start,
     * or when it drops out of `large`), so a linear array of n slots suffices; an index can re-enter
     * `large` many times but never more than n are live, so `large` is a ring of capacity n. */
    size_t lh = 0, lt = 0, gh = 0, gt = 0, gcount = 0;
    double norm = 0;
    memcpy(prob, w, sizeof(float) * n);
    for (size_t i = 0; i < n; i++) norm += prob[i];
    norm = norm / n;
    for (size_t i = 0; i < n; i++) prob[i] /= norm; /* float /= double -> rounded to float */
    for (size_t i = 0; i < n; i++) {
        if (prob[i] < 1)
            little[lt++] = i;
        else {
            large[gt] = i;
            gt = (gt + 1) % n;
            gcount++;
        }
    }
#define SET_ALIAS(i, j)            \
    do {                           \
        if (index_bytes == 8)      \
            a64[i] = (uint64_t)(j); \
        else                       \
            a32[i] = (uint32_t)(j); \
    } while (0)
    while (lh < lt && gcount > 0) {
        size_t i = little[lh++], j = large[gh];
        gh = (gh + 1) % n;
        gcount--;
        SET_ALIAS(i, j);
        prob[j] = prob[i] + prob[j] - 1;
        if (prob[j] < 1)
            little[lt++] = j;
        else {
            large[gt] = j;
            gt = (gt + 1) % n;
            gcount++;
        }
    }
    while (lh < lt) {
        size_t i = little[lh++];
        SET_ALIAS(i, i);
    }
    while (gcount > 0) {
        size_t i = large[gh];
        gh = (gh + 1) % n;
        gcount--;
        SET_ALIAS(i, i);
    }
#undef SET_ALIAS
    free(large);
    free(little);
    return 0;
}

uint64_t gvo_alias_sample(const float *prob, const void *alias, int index_bytes, uint64_t count, double rand1,
                          double rand2) {
    uint64_t index = (uint64_t)(rand1 * count);
    float p = (float)rand2;
    if (p < prob[index]) return index;
    return index_bytes == 8 ? ((const uint64_t *)alias)[index] : ((const uint32_t *)alias)[index];
}

/* gpu::Sample: both uniforms are narrowed to Float before sample() widens them again */
uint32_t gvo_alias_sample_gpu(const float *prob, const uint32_t *alias, uint32_t count, double rand1, double rand2) {
    float r1 = (float)rand1, r2 = (float)rand2;
    return (uint32_t)gvo_alias_sample(prob, alias, 4, count, (double)r1, (double)r2);
}

/* ---- Philox4x32-10 and the repo's RNG contract (include/gvk.h) -------------------- */

void gvo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define GVO_TAG_NEG 0x6e656721u  /* "neg!" */
#define GVO_TAG_HOST 0x686f7374u /* "host" */

/* Negative j of sample `sample_id` in batch `batch_id`:
 *   words = philox(ctr = {sample_id, batch_id, j / 2, TAG_NEG}, key = seed);  (w_a, w_b) = words[2*(j&1) ..]
 *   index = (w_a * count) >> 32;  u = (w_b >> 8) * 2^-24;  result = u < prob[index] ? index : alias[index] */
uint32_t gvo_negative_draw(const float *prob, const uint32_t *alias, uint32_t count, uint64_t seed,
                           uint32_t batch_id, uint32_t sample_id, uint32_t j) {
    uint32_t ctr[4] = {sample_id, batch_id, j / 2, GVO_TAG_NEG}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    gvo_philox4x32(ctr, key, w);
    uint32_t wa = w[2 * (j & 1)], wb = w[2 * (j & 1) + 1];
    uint32_t index = (uint32_t)(((uint64_t)wa * count) >> 32);
    float u = (float)(wb >> 8) * (1.0f / 16777216.0f);
    return u < prob[index] ? index : alias[index];
}

void gvo_negative_draw_batch(const float *prob, const uint32_t *alias, uint32_t count, uint64_t seed,
                              uint32_t batch_id, int batch_size, int k, uint32_t *out) {
    for (int s = 0; s < batch_size; s++)
        for (int j = 0; j < k; j++)
            out[(size_t)s * k + j] = gvo_negative_draw(prob, alias, count, seed, batch_id, (uint32_t)s, (uint32_t)j);
}

/* Negatives by weight classes (include/gvk.h "Negative sampling by weight classes"; the reference has one slot per row,
 * solver.h:1264-1278 — same distribution).  Classes: maximal runs of consecutive rows of equal weight; an alias table
 * over the class masses (rows * weight) built exactly as alias_table.cuh:84-128 builds any table (gvo_alias_build). */
#define GVO_TAG_NEG_CLASS 0x6e656743u /* "negC" */

/* first / count / prob / alias: [n] capacity; returns the number of classes */
uint32_t gvo_class_table_build(const float *weights, size_t n, uint32_t *first, uint32_t *count, float *prob, uint32_t *alias) {
    float *mass = (float *)malloc(n * sizeof(float));
    size_t classes = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && weights[j] == weights[i]) j++;
        first[classes] = (uint32_t)i, count[classes] = (uint32_t)(j - i);
        mass[classes++] = (float)((double)(j - i) * (double)weights[i]);
        i = j;
    }
    gvo_alias_build(mass, classes, prob, alias, 4);
    free(mass);
    return (uint32_t)classes;
}

uint32_t gvo_negative_draw_class(const uint32_t *first, const uint32_t *count, const float *prob, const uint32_t *alias,
                                 uint32_t num_class, uint64_t seed, uint32_t batch_id, uint32_t sample_id, uint32_t j) {
    uint32_t ctr[4] = {sample_id, batch_id, j, GVO_TAG_NEG_CLASS}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    gvo_philox4x32(ctr, key, w);
    uint32_t slot = (uint32_t)(((uint64_t)w[0] * num_class) >> 32);
    float u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    uint32_t c = u < prob[slot] ? slot : alias[slot];
    return first[c] + (uint32_t)(((uint64_t)w[2] * count[c]) >> 32);
}

void gvo_negative_draw_class_batch(const uint32_t *first, const uint32_t *count, const float *prob, const uint32_t *alias,
                                    uint32_t num_class, uint64_t seed, uint32_t batch_id, int batch_size, int k,
                                    uint32_t *out) {
    for (int s = 0; s < batch_size; s++)
        for (int j = 0; j < k; j++)
            out[(size_t)s * k + j] = gvo_negative_draw_class(first, count, prob, alias, num_class, seed, batch_id,
                                                             (uint32_t)s, (uint32_t)j);
}

#define GVO_TAG_POS 0x706f7321u /* "pos!" */

/* Device-side positive sampling of include/gvk.h (gvk_sample_pairs): out[t] = block_pairs[draw(first + t)] */
void gvo_sample_pairs(const float *prob, const uint32_t *alias, const uint32_t *block_pairs, uint32_t count,
                      uint64_t seed, uint64_t first, size_t n, uint32_t *out) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (size_t t = 0; t < n; t++) {
        uint64_t i = first + t;
        uint32_t ctr[4] = {(uint32_t)i, (uint32_t)(i >> 32), 0, GVO_TAG_POS}, w[4];
        gvo_philox4x32(ctr, key, w);
        uint32_t index = (uint32_t)(((uint64_t)w[0] * count) >> 32);
        float u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
        uint32_t edge = u < prob[index] ? index : alias[index];
        out[2 * t] = block_pairs[2 * (size_t)edge];
        out[2 * t + 1] = block_pairs[2 * (size_t)edge + 1];
    }
}

#define GVO_TAG_WALK 0x77616c6bu /* "walk" */

static int gvo_sorted_has(const uint32_t *sorted_nb, const uint64_t *flat, uint32_t x, uint32_t u) {
    uint64_t lo = flat[x], hi = flat[x + 1];
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (sorted_nb[mid] == u) return 1;
        if (sorted_nb[mid] < u)
            lo = mid + 1;
        else
            hi = mid;
    }
    return 0;
}

/* Device-side random-walk sampling of include/gvk.h (gvk_sample_walks), thread by thread. */
int gvo_sample_walks_device(const uint64_t *flat, const uint32_t *edges_uv, const float *edge_prob,
                            const uint32_t *edge_alias, uint32_t D, const float *nb_prob, const uint32_t *nb_alias,
                            const uint32_t *sorted_nb, const uint32_t *local, int biased, float p, float q,
                            uint64_t seed, uint64_t first_walk, uint32_t *pool, size_t pool_pairs, int L, int aug,
                            int shuffle_base) {
    if (aug < 1 || aug > 16 || aug > L || shuffle_base < 1 || pool_pairs % (size_t)shuffle_base) return -1;
    uint64_t per_walk = (uint64_t)aug * L - (uint64_t)aug * (aug - 1) / 2, sb = (uint64_t)shuffle_base;
    uint64_t walks = (pool_pairs + per_walk - 1) / per_walk, stride = pool_pairs / sb;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    float fmax = 1.0f;
    if (1.0f / p > fmax) fmax = 1.0f / p;
    if (1.0f / q > fmax) fmax = 1.0f / q;
    for (uint64_t t = 0; t < walks; t++) {
        uint64_t walk = first_walk + t, begin = t * per_walk;
        uint64_t end = begin + per_walk < pool_pairs ? begin + per_walk : pool_pairs, offset = begin;
        uint32_t draw = 0, window[16];
        while (offset < end) {
            uint32_t ctr[4] = {(uint32_t)walk, (uint32_t)(walk >> 32), draw++, GVO_TAG_WALK}, w[4];
            gvo_philox4x32(ctr, key, w);
            uint32_t index = (uint32_t)(((uint64_t)w[0] * D) >> 32);
            float u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
            uint64_t edge = u < edge_prob[index] ? index : edge_alias[index];
            uint32_t previous = edges_uv[2 * edge], current = edges_uv[2 * edge + 1];
            window[0] = local[previous];
            int j = 1;
            for (;;) {
                uint32_t row = local[current];
                int back = j < aug ? j : aug;
                for (int k = 1; k <= back && offset < end; k++) {
                    uint64_t slot = offset % sb * stride + offset / sb;
                    pool[2 * slot] = row;
                    pool[2 * slot + 1] = window[(j - k) % aug];
                    offset++;
                }
                window[j % aug] = row;
                if (j == L || offset >= end) break;
                uint64_t base = flat[current], degree = flat[current + 1] - base;
                if (degree == 0) break;
                uint32_t next;
                for (;;) {
                    uint32_t c2[4] = {(uint32_t)walk, (uint32_t)(walk >> 32), draw++, GVO_TAG_WALK};
                    gvo_philox4x32(c2, key, w);
                    index = (uint32_t)(((uint64_t)w[0] * (uint32_t)degree) >> 32);
                    u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
                    uint32_t neighbor = u < nb_prob[base + index] ? index : nb_alias[base + index];
                    next = edges_uv[2 * (base + neighbor) + 1];
                    if (!biased) break;
                    float f = next == previous ? 1.0f / p : (gvo_sorted_has(sorted_nb, flat, next, previous) ? 1.0f : 1.0f / q);
                    if ((float)(w[2] >> 8) * (1.0f / 16777216.0f) * fmax < f) break;
                }
                previous = current;
                current = next;
                j++;
            }
        }
    }
    return 0;
}

/* Host uniform stream `stream`: doubles number 2i and 2i+1 come from
 *   philox(ctr = {i_lo, i_hi, stream, TAG_HOST}, key = seed); d = (((u64)w_hi << 32 | w_lo) >> 11) * 2^-53 */
void gvo_host_uniforms(uint64_t seed, uint32_t stream, uint64_t first, size_t n, double *out) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (size_t t = 0; t < n; t++) {
        uint64_t idx = first + t, i = idx / 2;
        uint32_t ctr[4] = {(uint32_t)i, (uint32_t)(i >> 32), stream, GVO_TAG_HOST}, w[4];
        gvo_philox4x32(ctr, key, w);
        uint64_t bits = idx & 1 ? ((uint64_t)w[3] << 32 | w[2]) : ((uint64_t)w[1] << 32 | w[0]);
        out[t] = (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    }
}

/* ---- partition / schedule ---------------------------------------------------------- */

typedef struct {
    float w;
    uint32_t id;
} gvo_wid;

static int gvo_cmp_desc(const void *a, const void *b) {
    const gvo_wid *x = (const gvo_wid *)a, *y = (const gvo_wid *)b;
    if (x->w > y->w) return -1;
    if (x->w < y->w) return 1;
    /* std::sort leaves ties unspecified in the reference; this repo fixes them by ascending id */
    return x->id < y->id ? -1 : x->id > y->id;
}

/* part[v], local[v] for every vertex; part_sizes[P]. */
int gvo_partition(const float *weights, uint32_t n, int P, int32_t *part, uint32_t *local, uint32_t *part_sizes) {
    gvo_wid *order = (gvo_wid *)malloc(sizeof(gvo_wid) * (n ? n : 1));
    if (!order) return -1;
    for (uint32_t i = 0; i < n; i++) {
        order[i].w = weights[i];
        order[i].id = i;
    }
    qsort(order, n, sizeof(gvo_wid), gvo_cmp_desc);
    for (int p = 0; p < P; p++) part_sizes[p] = 0;
    for (uint32_t i = 0; i < n; i++) {
        int pid = (int)(i % (uint32_t)(P * 2));
        if (pid > P * 2 - 1 - pid) pid = P * 2 - 1 - pid;
        part[order[i].id] = pid;
        local[order[i].id] = part_sizes[pid]++;
    }
    free(order);
    return 0;
}

/* out[step][worker] = {head_part, tail_part}; returns number of steps. out needs (P/W)^2*W*W*2 ints. */
int gvo_schedule(int P, int W, int32_t *out) {
    int steps = 0;
    if (P == 1) {
        out[0] = 0;
        out[1] = 0;
        return 1;
    }
    for (int x = 0; x < P; x += W)
        for (int y = 0; y < P; y += W)
            for (int offset = 0; offset < W; offset++) {
                for (int i = 0; i < W; i++) {
                    out[(steps * W + i) * 2] = x + (i + offset) % W;
                    out[(steps * W + i) * 2 + 1] = y + i;
                }
                steps++;
            }
    return steps;
}

void gvo_negative_weights(const float *vertex_weights, const uint32_t *global_ids, uint32_t n, float exponent,
                          float *out) {
    for (uint32_t i = 0; i < n; i++) out[i] = powf(vertex_weights[global_ids[i]], exponent);
}

/* ---- samplers ---------------------------------------------------------------------- */

typedef struct {
    const double *r;
    size_t n, pos;
} gvo_rand;

static double gvo_next(gvo_rand *g) { return g->pos < g->n ? g->r[g->pos++] : (g->pos++, 0.0); }

/* Base edge sampler.  edges_uv = flattened directed edges {u, v}; pools[hp * P + tp] -> {tail, head}
 * records of capacity pool_size.  Fills [start, end) of every block pool.  Returns randoms consumed,
 * or (size_t)-1 if the supplied stream was too short. */
size_t gvo_sample_edges(const uint32_t *edges_uv, const float *edge_prob, const uint64_t *edge_alias,
                        uint64_t num_edge_entries, const int32_t *part, const uint32_t *local, int P,
                        uint32_t **pools, int start, int end, int sample_batch_size, int tail_filter,
                        const double *rnd, size_t n_rnd) {
    gvo_rand g = {rnd, n_rnd, 0};
    if (start >= end) return 0;
    int *offsets = (int *)malloc(sizeof(int) * P * P);
    uint32_t *hu = (uint32_t *)malloc(sizeof(uint32_t) * sample_batch_size * 2);
    for (int i = 0; i < P * P; i++) offsets[i] = start;
    int num_complete = 0;
    /* tail_filter >= 0 (this repo's one-column-per-GPU mode): only blocks (*, tail_filter) are filled and
     * waited for; the caller passes the edge list / table restricted to edges that end in that partition */
    const int target = tail_filter < 0 ? P * P : P;
    while (num_complete < target) {
        for (int i = 0; i < sample_batch_size; i++) {
            double r1 = gvo_next(&g), r2 = gvo_next(&g);
            uint64_t e = gvo_alias_sample(edge_prob, edge_alias, 8, num_edge_entries, r1, r2);
            hu[2 * i] = edges_uv[2 * e];
            hu[2 * i + 1] = edges_uv[2 * e + 1];
        }
        for (int i = 0; i < sample_batch_size; i++) {
            uint32_t h = hu[2 * i], t = hu[2 * i + 1];
            int hp = part[h], tp = part[t];
            int *offset = &offsets[hp * P + tp];
            if (tail_filter >= 0 && tp != tail_filter) continue;
            if (*offset < end) {
                uint32_t *pool = pools[hp * P + tp];
                pool[2 * (size_t)*offset] = local[t];
                pool[2 * (size_t)*offset + 1] = local[h];
                if (++*offset == end) num_complete++;
            }
        }
    }
    free(offsets);
    free(hu);
    return g.pos > g.n ? (size_t)-1 : g.pos;
}

/* Random-walk sampler (biased = 0: per-vertex tables indexed by CSR slot; biased = 1: node2vec
 * per-edge tables, table of directed edge e starts at ee_offsets[e] and has deg(v) entries). */
static size_t gvo_sample_walks_in_order(int reference_order, int biased, const uint32_t *edges_uv, const float *edge_prob,
                                       const uint64_t *edge_alias, uint64_t num_edge_entries,
                                       const uint64_t *flat_offsets /* [N+1] */, const float *nb_prob,
                                       const uint32_t *nb_alias, const uint64_t *ee_offsets, const int32_t *part,
                                       const uint32_t *local, int P, uint32_t **pools, int pool_size, int start, int end,
                                       int walk_length, int walk_batch, int augmentation_step, int shuffle_base,
                                       int tail_filter, const uint32_t *sorted_nb, float p, float q, const double *rnd,
                                       size_t n_rnd) {
    gvo_rand g = {rnd, n_rnd, 0};
    if (start >= end) return 0;
    if (pool_size % shuffle_base) return (size_t)-2;
    int L = walk_length;
    int *offsets = (int *)malloc(sizeof(int) * P * P);
    uint32_t *chains = (uint32_t *)malloc(sizeof(uint32_t) * walk_batch * (L + 1));
    int *lengths = (int *)malloc(sizeof(int) * walk_batch);
    for (int i = 0; i < P * P; i++) offsets[i] = start;
    int num_complete = 0;
    const int target = tail_filter < 0 ? P * P : P; /* walks: pairs ending elsewhere are drawn, then dropped */
    uint64_t *edge_ids = (uint64_t *)malloc(sizeof(uint64_t) * walk_batch);
    uint32_t *currents = (uint32_t *)malloc(sizeof(uint32_t) * walk_batch);
    uint64_t *proposals = (uint64_t *)malloc(sizeof(uint64_t) * walk_batch);
    int *pending = (int *)malloc(sizeof(int) * walk_batch);
    float fmax = 1.0f;
    if (biased == 2) {
        if (1.0f / p > fmax) fmax = 1.0f / p;
        if (1.0f / q > fmax) fmax = 1.0f / q;
    }
    while (num_complete < target) {
        /* This repo's samplers advance the walks of a round in lockstep (all start edges, then step 2 of every
         * live walk, ...), where the reference finishes walk i before starting walk i + 1; the walk itself —
         * start edge by weight, next node from the vertex / edge alias table, stop at a node without out-edges —
         * is the reference's.  Only the order in which the i.i.d. uniforms are consumed differs. */
        /* reference_order: the reference's own order (graph.cuh:320-348, 399-424) — walk i is finished before walk
         * i + 1 starts, two uniforms per alias draw, taken as they come.  Same walks as below for the same uniforms
         * per draw; used to pin this restatement record for record against pools the reference's samplers filled. */
        for (int i = 0; reference_order && i < walk_batch; i++) {
            uint32_t *chain = chains + (size_t)i * (L + 1);
            double r1 = gvo_next(&g), r2 = gvo_next(&g);
            uint64_t edge = gvo_alias_sample(edge_prob, edge_alias, 8, num_edge_entries, r1, r2);
            uint32_t current = edges_uv[2 * edge + 1];
            chain[0] = edges_uv[2 * edge], chain[1] = current;
            lengths[i] = L;
            for (int j = 2; j <= L; j++) {
                uint64_t deg = flat_offsets[current + 1] - flat_offsets[current];
                if (!deg) {
                    lengths[i] = j - 1;
                    break;
                }
                r1 = gvo_next(&g), r2 = gvo_next(&g);
                uint64_t base = biased == 1 ? ee_offsets[edge] : flat_offsets[current];
                edge = flat_offsets[current] + (uint32_t)gvo_alias_sample(nb_prob + base, nb_alias + base, 4, deg, r1, r2);
                current = edges_uv[2 * edge + 1];
                chain[j] = current;
            }
        }
        for (int i = 0; !reference_order && i < walk_batch; i++) {
            uint32_t *chain = chains + (size_t)i * (L + 1);
            double r1 = gvo_next(&g), r2 = gvo_next(&g);
            edge_ids[i] = gvo_alias_sample(edge_prob, edge_alias, 8, num_edge_entries, r1, r2);
            chain[0] = edges_uv[2 * edge_ids[i]];
            currents[i] = edges_uv[2 * edge_ids[i] + 1];
            chain[1] = currents[i];
            lengths[i] = L;
        }
        for (int j = 2; !reference_order && j <= L; j++) {
            int num_pending = 0;
            for (int i = 0; i < walk_batch; i++) {
                if (lengths[i] < L) continue; /* stopped earlier */
                if (flat_offsets[currents[i] + 1] == flat_offsets[currents[i]])
                    lengths[i] = j - 1;
                else
                    pending[num_pending++] = i;
            }
            int num_live = num_pending;
            /* biased == 2 (this repo's O(|E|)-memory node2vec): propose from the per-vertex table, accept with
             * probability f / fmax (third uniform), re-propose for the rejected walks until all have moved */
            while (num_pending) {
                int rejected = 0;
                for (int n = 0; n < num_pending; n++) {
                    int i = pending[n];
                    uint32_t current = currents[i];
                    uint64_t deg = flat_offsets[current + 1] - flat_offsets[current];
                    double r1 = gvo_next(&g), r2 = gvo_next(&g);
                    float r3 = biased == 2 ? (float)gvo_next(&g) : 0.0f;
                    uint64_t base = biased == 1 ? ee_offsets[edge_ids[i]] : flat_offsets[current];
                    uint32_t nb = (uint32_t)gvo_alias_sample(nb_prob + base, nb_alias + base, 4, deg, r1, r2);
                    proposals[i] = flat_offsets[current] + nb;
                    if (biased == 2) {
                        uint32_t x = edges_uv[2 * proposals[i] + 1], prev = edges_uv[2 * edge_ids[i]];
                        float f = x == prev ? 1.0f / p : (gvo_sorted_has(sorted_nb, flat_offsets, x, prev) ? 1.0f : 1.0f / q);
                        if (!(r3 * fmax < f)) pending[rejected++] = i;
                    }
                }
                num_pending = rejected;
            }
            (void)num_live;
            for (int i = 0; i < walk_batch; i++) {
                if (lengths[i] < L) continue;
                edge_ids[i] = proposals[i];
                currents[i] = edges_uv[2 * edge_ids[i] + 1];
                chains[(size_t)i * (L + 1) + j] = currents[i];
            }
        }
        for (int i = 0; i < walk_batch; i++) {
            const uint32_t *chain = chains + (size_t)i * (L + 1);
            for (int j = 0; j < lengths[i]; j++)
                for (int k = 1; k <= augmentation_step; k++) {
                    if (j + k > lengths[i]) break;
                    uint32_t h = chain[j], t = chain[j + k];
                    int hp = part[h], tp = part[t];
                    int *offset = &offsets[hp * P + tp];
                    if (tail_filter >= 0 && tp != tail_filter) continue;
                    if (*offset < end) {
                        uint32_t *pool = pools[hp * P + tp];
                        int shuffled = *offset % shuffle_base * (pool_size / shuffle_base) + *offset / shuffle_base;
                        pool[2 * (size_t)shuffled] = local[t];
                        pool[2 * (size_t)shuffled + 1] = local[h];
                        if (++*offset == end) num_complete++;
                    }
                }
        }
    }
    free(offsets);
    free(chains);
    free(lengths);
    free(edge_ids);
    free(currents);
    free(proposals);
    free(pending);
    return g.pos > g.n ? (size_t)-1 : g.pos;
}

size_t gvo_sample_walks(int biased, const uint32_t *edges_uv, const float *edge_prob, const uint64_t *edge_alias,
                        uint64_t num_edge_entries, const uint64_t *flat_offsets, const float *nb_prob,
                        const uint32_t *nb_alias, const uint64_t *ee_offsets, const int32_t *part,
                        const uint32_t *local, int P, uint32_t **pools, int pool_size, int start, int end,
                        int walk_length, int walk_batch, int augmentation_step, int shuffle_base, int tail_filter,
                        const uint32_t *sorted_nb, float p, float q, const double *rnd, size_t n_rnd) {
    return gvo_sample_walks_in_order(0, biased, edges_uv, edge_prob, edge_alias, num_edge_entries, flat_offsets, nb_prob,
                                     nb_alias, ee_offsets, part, local, P, pools, pool_size, start, end, walk_length,
                                     walk_batch, augmentation_step, shuffle_base, tail_filter, sorted_nb, p, q, rnd, n_rnd);
}

/* The same sampler consuming its uniforms walk by walk, as GraphSampler::sample_random_walk (biased 0) and
 * sample_biased_random_walk (biased 1) do (graph.cuh:376-450, 298-373); no rejection mode (biased 2 is this repo's). */
size_t gvo_sample_walks_reference_order(int biased, const uint32_t *edges_uv, const float *edge_prob,
                                        const uint64_t *edge_alias, uint64_t num_edge_entries,
                                        const uint64_t *flat_offsets, const float *nb_prob, const uint32_t *nb_alias,
                                        const uint64_t *ee_offsets, const int32_t *part, const uint32_t *local, int P,
                                        uint32_t **pools, int pool_size, int start, int end, int walk_length,
                                        int walk_batch, int augmentation_step, int shuffle_base, const double *rnd,
                                        size_t n_rnd) {
    if (biased != 0 && biased != 1) return (size_t)-2;
    return gvo_sample_walks_in_order(1, biased, edges_uv, edge_prob, edge_alias, num_edge_entries, flat_offsets, nb_prob,
                                     nb_alias, ee_offsets, part, local, P, pools, pool_size, start, end, walk_length,
                                     walk_batch, augmentation_step, shuffle_base, -1, NULL, 1.0f, 1.0f, rnd, n_rnd);
}

static int gvo_has_neighbor(const uint32_t *edges_uv, const uint64_t *flat_offsets, uint32_t x, uint32_t u) {
    for (uint64_t e = flat_offsets[x]; e < flat_offsets[x + 1]; e++)
        if (edges_uv[2 * e + 1] == u) return 1;
    return 0;
}

/* node2vec transition weights of directed edge e = (u -> v): one entry per out-edge (v -> x, w). */
void gvo_edge_edge_weights(const uint32_t *edges_uv, const float *edge_weights, const uint64_t *flat_offsets,
                           uint64_t e, float p, float q, float *out) {
    uint32_t u = edges_uv[2 * e], v = edges_uv[2 * e + 1];
    for (uint64_t f = flat_offsets[v]; f < flat_offsets[v + 1]; f++) {
        uint32_t x = edges_uv[2 * f + 1];
        float w = edge_weights[f];
        if (x == u)
            *out++ = w / p;
        else if (!gvo_has_neighbor(edges_uv, flat_offsets, x, u))
            *out++ = w / q;
        else
            *out++ = w;
    }
}
