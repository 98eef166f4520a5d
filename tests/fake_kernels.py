"""A stand-in for graphvite_amd.kernels.HipKernels that runs on CPU tensors by calling the ORACLE.

TEST INFRASTRUCTURE ONLY: it lets the `-m "not gpu"` suite exercise the solver's host logic (partitioning,
schedule, sampling, pool handling, batch-id / lr accounting, the all-gather exchange over gloo, write-back)
without a GPU.  It is injected explicitly (`GraphSolver(..., kernels=OracleKernels())`); the product never
imports it and has no CPU fallback of its own."""
import numpy as np
import torch

from oracle_lib import Oracle

_OPT = {"SGD": 0, "Momentum": 1, "AdaGrad": 2, "RMSprop": 3, "Adam": 4}


class OracleKernels(object):
    name = "oracle"
    device = "cpu"

    def __init__(self):
        self.oracle = Oracle()
        self.launches = []  # (batch_id, lr) per batch, for accounting tests

    @staticmethod
    def _np(t):
        return None if t is None else t.numpy()

    def _negatives(self, table, seed, batch_id, B, k):
        if table.dim() == 2:  # a class table (gvk_class_entry per weight class)
            e = table.numpy().view(np.dtype([("prob", np.float32), ("alias", np.uint32), ("first", np.uint32),
                                             ("count", np.uint32)])).reshape(-1)
            return self.oracle.negatives_by_class((e["first"], e["count"], e["prob"], e["alias"]), seed, batch_id, B, k)
        packed = table.numpy().view(np.dtype([("prob", np.float32), ("alias", np.uint32)]))
        prob, alias = np.ascontiguousarray(packed["prob"]), np.ascontiguousarray(packed["alias"])
        return self.oracle.negatives(prob, alias, seed, batch_id, B, k)

    def train(self, vertex, context, pairs, loss, optimizer, num_negative, negative_weight, negatives=None,
              table=None, seed=0, batch_id=0, moments=None, lr=None):
        B = pairs.shape[0]
        lr = optimizer.lr if lr is None else lr
        negs = negatives.numpy().view(np.uint32).reshape(B, num_negative) if negatives is not None else \
            self._negatives(table, seed, batch_id, B, num_negative)
        m = None if moments is None else [self._np(x) for x in moments]
        hp = (optimizer.hp0, optimizer.hp1, optimizer.epsilon)
        out = self.oracle.train(vertex.numpy(), context.numpy(), np.ascontiguousarray(pairs.numpy().view(np.uint32)),
                                np.ascontiguousarray(negs), lr, optimizer.weight_decay, negative_weight,
                                _OPT[optimizer.type], m, hp)
        loss.numpy()[:B] = out
        self.launches.append((batch_id, lr))

    def train_episode(self, vertex, context, pool, loss, optimizer, num_negative, negative_weight, table, seed,
                      first_batch_id, total_batches, num_batches, batch_size, moments=None, batch_id_stride=1):
        for i in range(num_batches):
            bid = first_batch_id + i * batch_id_stride
            scale = self.oracle.lr(1.0, optimizer.schedule == "linear", bid, total_batches)
            pairs = pool[i * batch_size * 2:(i + 1) * batch_size * 2].view(batch_size, 2)
            self.train(vertex, context, pairs, loss, optimizer, num_negative, negative_weight, table=table, seed=seed,
                       batch_id=bid, moments=moments, lr=np.float32(optimizer.lr) * np.float32(scale))

    def group_pairs(self, pool_in, pool_out, batch_size, num_batch, num_row):
        """Per batch: stable sort of the records by the low row bits of the head (what gvk_group_pairs promises)."""
        bits = max(int(num_row - 1).bit_length(), 1)
        rec = pool_in.numpy().view(np.uint32)[:2 * batch_size * num_batch].reshape(num_batch, batch_size, 2)
        out = pool_out.numpy().view(np.uint32)[:2 * batch_size * num_batch].reshape(num_batch, batch_size, 2)
        for i in range(num_batch):
            out[i] = rec[i][np.argsort(rec[i, :, 1] & np.uint32((1 << bits) - 1), kind="stable")]

    def sample_pairs(self, table, block_pairs, seed, first_index, pool, n):
        packed = table.numpy().view(np.dtype([("prob", np.float32), ("alias", np.uint32)]))
        out = self.oracle.sample_pairs(np.ascontiguousarray(packed["prob"]), np.ascontiguousarray(packed["alias"]),
                                       block_pairs.numpy().view(np.uint32), seed, first_index, n)
        pool.numpy().view(np.uint32)[:2 * n] = out.reshape(-1)

    @staticmethod
    def pack_edge_table(table, block_pairs):
        return torch.cat([table.view(torch.int32).view(-1, 2), block_pairs.view(-1, 2)], 1).contiguous().view(torch.int64)

    def sample_edges(self, edge_table, seed, first_index, pool, n):
        words = edge_table.view(torch.int32).view(-1, 4)
        self.sample_pairs(words[:, :2].contiguous().view(torch.int64).view(-1), words[:, 2:].contiguous().view(-1), seed,
                          first_index, pool, n)

    def sample_walks(self, walk_graph, seed, first_walk, pool, pool_pairs, walk_length, augmentation_step,
                     shuffle_base):
        g = walk_graph
        entry = np.dtype([("prob", np.float32), ("alias", np.uint32)])
        et, nt = g["edge_table"].numpy().view(entry), g["neighbor_table"].numpy().view(entry)
        snb = g.get("sorted_neighbors")
        out = self.oracle.sample_walks_device(
            g["flat_offsets"].numpy().view(np.uint64), g["edges_uv"].numpy().view(np.uint32).reshape(-1, 2),
            np.ascontiguousarray(et["prob"]), np.ascontiguousarray(et["alias"]), np.ascontiguousarray(nt["prob"]),
            np.ascontiguousarray(nt["alias"]), None if snb is None else snb.numpy().view(np.uint32),
            g["local"].numpy().view(np.uint32), bool(g.get("biased", False)), float(g.get("p", 1.0)),
            float(g.get("q", 1.0)), seed, first_walk, pool_pairs, walk_length, augmentation_step, shuffle_base)
        pool.numpy().view(np.uint32)[:2 * pool_pairs] = out.reshape(-1)

    def sample_walks_blocks(self, walk_graph, part, num_partition, seed, first_walk, pools, offsets, capacity, walk_length,
                            augmentation_step, shuffle_base, max_rounds=64):
        """gvk_sample_walks_blocks restated: the walks of the single-partition oracle sampler (vertex ids instead of
        rows), every pair binned into the pool of its block in walk order until the collected pools are full."""
        g, P = walk_graph, int(num_partition)
        entry = np.dtype([("prob", np.float32), ("alias", np.uint32)])
        et, nt = g["edge_table"].numpy().view(entry), g["neighbor_table"].numpy().view(entry)
        snb = g.get("sorted_neighbors")
        local = g["local"].numpy().view(np.uint32)
        part_of = part.numpy()
        where = offsets.numpy()
        out = pools.numpy().view(np.uint32).reshape(-1, 2)
        aug, L = int(augmentation_step), int(walk_length)
        per_walk = aug * L - aug * (aug - 1) // 2
        count = np.zeros(P * P, np.int64)
        walks, used = -(-capacity * P * P // per_walk), 0
        for _ in range(max_rounds):
            pairs = self.oracle.sample_walks_device(
                g["flat_offsets"].numpy().view(np.uint64), g["edges_uv"].numpy().view(np.uint32).reshape(-1, 2),
                np.ascontiguousarray(et["prob"]), np.ascontiguousarray(et["alias"]), np.ascontiguousarray(nt["prob"]),
                np.ascontiguousarray(nt["alias"]), None if snb is None else snb.numpy().view(np.uint32),
                np.arange(len(local), dtype=np.uint32), bool(g.get("biased", False)), float(g.get("p", 1.0)),
                float(g.get("q", 1.0)), seed, first_walk + used, walks * per_walk, L, aug, 1)
            used += walks
            block = part_of[pairs[:, 1]].astype(np.int64) * P + part_of[pairs[:, 0]]
            for b in np.flatnonzero(where >= 0):
                mine = pairs[block == b]
                take = mine[:max(capacity - count[b], 0)]
                slot = count[b] + np.arange(len(take))
                position = slot % shuffle_base * (capacity // shuffle_base) + slot // shuffle_base
                out[where[b] + position] = np.stack([local[take[:, 0]], local[take[:, 1]]], 1)
                count[b] += len(mine)
            if (count[where >= 0] >= capacity).all():
                return used
            walks = max(walks // 2, 64)
        raise RuntimeError("pools not full")

    def predict(self, vertex, context, pairs, logits):
        out = self.oracle.predict(vertex.numpy(), context.numpy(), np.ascontiguousarray(pairs.numpy().view(np.uint32)))
        logits.numpy()[:len(out)] = out
