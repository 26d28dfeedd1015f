#include "prof.h"

#include <vector>

#include "common.h"

namespace balf_prof {

bool g_on = false;

namespace {
struct Rec { int slot; hipEvent_t a, b; };
std::vector<hipEvent_t> g_pool;     // created once, reused
size_t g_used = 0;
std::vector<Rec> g_recs;
hipEvent_t g_pending = nullptr;
int g_pending_slot = -1;
// the `after` stamp of the previous launch of the current chain (launches issued back to back on one stream by one
// entry point, nothing else in between): the next launch of the chain takes it as its `before` stamp instead of
// recording another event -- an event record costs the stream ~4 us, 0.55 ms per 32-image step with a pair per launch
hipEvent_t g_chain_last = nullptr;
hipStream_t g_chain_stream = nullptr;
bool g_chain_on = false;

hipEvent_t take_event() {
    if (g_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_used++];
}
}  // namespace

void chain_begin() { g_chain_last = nullptr; g_chain_on = true; }
void chain_end() { g_chain_last = nullptr; g_chain_on = false; }

void before(int slot, hipStream_t st) {
    g_pending_slot = slot;
    if (g_chain_on && g_chain_last && g_chain_stream == st) {
        g_pending = g_chain_last;
        return;
    }
    g_pending = take_event();
    if (g_pending) (void)hipEventRecord(g_pending, st);
}

void after(hipStream_t st) {
    hipEvent_t e = take_event();
    g_chain_last = nullptr;
    if (!g_pending || !e) return;
    (void)hipEventRecord(e, st);
    g_recs.push_back({g_pending_slot, g_pending, e});
    g_pending = nullptr;
    g_chain_last = e;
    g_chain_stream = st;
}

}  // namespace balf_prof

static const char *kSlotNames[balf_prof::kNumSlots] = {
    "stage1_grid_branch", "stage1_block_branch", "stage1_se", "stage1_pool",
    "stage2_grid_branch", "stage2_block_branch", "stage2_se", "stage2_pool",
    "stage3_grid_branch", "stage3_block_branch", "stage3_se", "stage3_pool",
    "stage4_grid_branch", "stage4_block_branch", "stage4_se", "stage4_head",
    "nms_tile", "topk_select",
    "hardnet_conv1_2", "hardnet_conv3", "hardnet_conv4", "hardnet_conv5", "hardnet_conv6", "hardnet_fc",
    "patch_pyrdown", "patch_sample", "match_nn", "match_mutual", "greedy_keep", "greedy_kill"};

extern "C" int balf_profile_num_slots(void) { return balf_prof::kNumSlots; }

extern "C" const char *balf_profile_slot_name(int slot) {
    return (slot >= 0 && slot < balf_prof::kNumSlots) ? kSlotNames[slot] : nullptr;
}

extern "C" int balf_profile_begin(void) {
    balf_prof::chain_end();
    balf_prof::g_recs.clear();
    balf_prof::g_used = 0;
    balf_prof::g_on = true;
    return BALF_OK;
}

extern "C" int balf_profile_end(float *ms_total, int *launches) {
    using namespace balf_prof;
    g_on = false;
    if (!ms_total || !launches) return BALF_ERR_ARG;
    for (int i = 0; i < kNumSlots; ++i) { ms_total[i] = 0.0f; launches[i] = 0; }
    for (const Rec &r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return BALF_ERR_LAUNCH;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return BALF_ERR_LAUNCH;
        ms_total[r.slot] += ms;
        launches[r.slot] += 1;
    }
    g_recs.clear();
    g_used = 0;
    return BALF_OK;
}
