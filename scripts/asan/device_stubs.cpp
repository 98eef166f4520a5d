#include "gvk.h"
#include "gvk_internal.h"
extern "C" {
int gvk_train(void *, int, const gvk_optimizer *, const gvk_tables *, const uint32_t *, const gvk_negative_source *, uint32_t, float *, int, int, float) { return GVK_EHIP; }
int gvk_train_episode(void *, int, const gvk_optimizer *, int, const gvk_tables *, const uint32_t *, const gvk_negative_source *, uint32_t, uint32_t, uint32_t, int, float *, int, int, float) { return GVK_EHIP; }
int gvk_predict(void *, int, const float *, const float *, const uint32_t *, float *, int) { return GVK_EHIP; }
int gvk_hot_plan(int, int, uint32_t, uint32_t, int, int, int, size_t *) { return GVK_EHIP; }
int gvk_hot_build(void *, void *, size_t, const uint32_t *, int, int, int, const gvk_negative_source *, uint32_t, uint32_t, uint32_t, uint32_t, int, int) { return GVK_EHIP; }
int gvk_train_episode_hot(void *, int, const gvk_optimizer *, int, const gvk_tables *, const uint32_t *, const gvk_negative_source *, uint32_t, uint32_t, uint32_t, int, float *, int, int, float, const void *, size_t, uint32_t, uint32_t, int, int, int, int) { return GVK_EHIP; }
int gvk_alias_sample(void *, const gvk_alias_entry *, uint32_t, const double *, uint32_t *, int) { return GVK_EHIP; }
int gvk_negative_draw(void *, const gvk_alias_entry *, uint32_t, uint64_t, uint32_t, uint32_t *, int, int) { return GVK_EHIP; }
int gvk_sample_pairs(void *, const gvk_alias_entry *, const uint32_t *, uint32_t, uint64_t, uint64_t, uint32_t *, size_t) { return GVK_EHIP; }
int gvk_sample_walks(void *, const gvk_walk_graph *, uint64_t, uint64_t, uint32_t *, size_t, int, int, int) { return GVK_EHIP; }
int gvk_sample_walks_blocks(void *, const gvk_walk_graph *, const int32_t *, int, uint64_t, uint64_t, uint64_t, uint32_t *, const uint64_t *, uint32_t *, uint32_t, int, int, int, int) { return GVK_EHIP; }
int gvk_describe_train(int, int, int, int, int, uint32_t, char *, size_t) { return GVK_EHIP; }
int gvk_group_pairs(void *, const uint32_t *, uint32_t *, void *, size_t *, int, int, int) { return GVK_EHIP; }
int gvk_set_tuning(int, int) { return GVK_OK; }
int gvk_sample_edges(void *, const gvk_edge_entry *, uint32_t, uint64_t, uint64_t, uint32_t *, size_t) { return GVK_EHIP; }
int gvk_negative_draw_classes(void *, const gvk_class_entry *, uint32_t, uint64_t, uint32_t, uint32_t *, int, int) { return GVK_EHIP; }
int gvk_probe_row_traffic(void *, int, float *, float *, const uint32_t *, const uint32_t *, float, int) { return GVK_EHIP; }
}
