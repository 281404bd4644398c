// Op-table replay: one C-ABI call issues a whole pre-marshalled range of C-ABI launches.
//
// The training step of the hot path (reference main.py:416-445: forward, loss, backward, SGD) is a static list of ~500 C-ABI
// calls over fixed buffers (radar_depth_amd/engine.py).  Issuing it from a Python `for` loop costs a ctypes marshalling per call
// and keeps one host core busy per rank; with eight one-process-per-GPU ranks on a node that is the weak-scaling risk (SURVEY.md
// 8e).  A table stores, per op, the entry point and its arguments as 64-bit words (pointers and integers by value, floats as
// their bit pattern); arguments that are streams are SLOTS, patched from the `streams` array of each rd_optable_run call, so a
// table outlives stream rebinding and hipGraph capture.  rd_optable_run does nothing but call the same entry points the Python
// loop would -- there is no second implementation of any op.
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"

namespace {

using Thunk = int (*)(const uint64_t*);

template <typename T>
inline T word_as(uint64_t w) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t b = static_cast<uint32_t>(w);
        float f;
        memcpy(&f, &b, 4);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double f;
        memcpy(&f, &w, 8);
        return f;
    } else {
        return static_cast<T>(w);
    }
}

template <auto Fn>
struct ThunkOf;
template <typename... A, int (*Fn)(A...)>
struct ThunkOf<Fn> {
    template <size_t... I>
    static int call_(const uint64_t* w, std::index_sequence<I...>) {
        return Fn(word_as<A>(w[I])...);
    }
    static int call(const uint64_t* w) { return call_(w, std::index_sequence_for<A...>{}); }
    static constexpr int nargs = static_cast<int>(sizeof...(A));
};

struct Entry {
    Thunk fn;
    int nargs;
};

#define RD_ENTRY(name) {#name, Entry{&ThunkOf<&name>::call, ThunkOf<&name>::nargs}}

const std::unordered_map<std::string, Entry>& registry() {
    static const std::unordered_map<std::string, Entry> r = {
        RD_ENTRY(rd_gconv), RD_ENTRY(rd_gconv_ws), RD_ENTRY(rd_gconv_fused), RD_ENTRY(rd_gconv_bnbwd), RD_ENTRY(rd_gconv_bf16_t), RD_ENTRY(rd_gconv_split),
        RD_ENTRY(rd_wgrad), RD_ENTRY(rd_wgrad_reduce), RD_ENTRY(rd_wgrad_reduce_batched),
        RD_ENTRY(rd_wgrad_bf16_t), RD_ENTRY(rd_wgrad_bf16), RD_ENTRY(rd_wgrad_bf16_reduce), RD_ENTRY(rd_wgrad_split), RD_ENTRY(rd_wgrad_split_reduce),
        RD_ENTRY(rd_pack_weights_batched), RD_ENTRY(rd_fill),
        RD_ENTRY(rd_stem_fwd_t), RD_ENTRY(rd_stem_fwd_bf16_t), RD_ENTRY(rd_stem_fwd_split), RD_ENTRY(rd_conv16_split), RD_ENTRY(rd_stem_wgrad_t), RD_ENTRY(rd_stem_wgrad_split_t), RD_ENTRY(rd_stem_wgrad_split_bn_t), RD_ENTRY(rd_stem_dgrad_channel_t),
        RD_ENTRY(rd_bn_finalize), RD_ENTRY(rd_bn_eval_coeffs), RD_ENTRY(rd_bn_eval_coeffs_batched), RD_ENTRY(rd_bn_act_t),
        RD_ENTRY(rd_bn_bwd_reduce_t), RD_ENTRY(rd_bn_bwd_reduce_x_t), RD_ENTRY(rd_bn_bwd_reduce_x2_t),
        RD_ENTRY(rd_bn_bwd_apply_t), RD_ENTRY(rd_bn_bwd_apply_x_t), RD_ENTRY(rd_bn_bwd_apply_x2_t),
        RD_ENTRY(rd_bnact_maxpool_fwd_t), RD_ENTRY(rd_bnact_maxpool_bwd_stats_t), RD_ENTRY(rd_bnact_maxpool_bwd_apply_t),
        RD_ENTRY(rd_head_conv_fwd_t), RD_ENTRY(rd_head_conv_bwd_t), RD_ENTRY(rd_head_conv_dgrad_t), RD_ENTRY(rd_head_conv_wgrad_t), RD_ENTRY(rd_bilinear_fwd), RD_ENTRY(rd_bilinear_bwd),
        RD_ENTRY(rd_masked_l1_sums), RD_ENTRY(rd_masked_l1_bwd), RD_ENTRY(rd_masked_l2_sums), RD_ENTRY(rd_masked_l2_bwd),
        RD_ENTRY(rd_l1_total), RD_ENTRY(rd_smooth_fwd), RD_ENTRY(rd_smooth_bwd), RD_ENTRY(rd_uncertainty_total),
        RD_ENTRY(rd_radar_filter), RD_ENTRY(rd_sgd_step),
        RD_ENTRY(rd_event_record), RD_ENTRY(rd_stream_wait_event), RD_ENTRY(rd_allreduce_bucket), RD_ENTRY(rd_broadcast),
        RD_ENTRY(rd_debug_poison_lds),
        RD_ENTRY(rd_gconv_split_pre), RD_ENTRY(rd_wgrad_split_pre), RD_ENTRY(rd_split_pieces),
        RD_ENTRY(rd_wino_conv3x3), RD_ENTRY(rd_wino_pack_batched),
        RD_ENTRY(rd_bn_act_p), RD_ENTRY(rd_bn_bwd_apply_p), RD_ENTRY(rd_bn_bwd_apply_x_p), RD_ENTRY(rd_bn_bwd_apply_x2_p), RD_ENTRY(rd_bnact_maxpool_fwd_p),
    };
    return r;
}

struct Op {
    Thunk fn;
    int first_word, nargs;
    int first_patch, n_patch;
};

struct Table {
    std::vector<Op> ops;
    std::vector<uint64_t> words;
    std::vector<std::pair<int, int>> patches;   // (word index, stream slot)
    int max_slot = -1;
};

}  // namespace

extern "C" int rd_optable_create(void** table) {
    RD_CHECK_ARG(table != nullptr, "rd_optable_create: null out pointer");
    *table = new Table();
    return RD_OK;
}

extern "C" int rd_optable_destroy(void* table) {
    delete static_cast<Table*>(table);
    return RD_OK;
}

extern "C" int rd_optable_entry_args(const char* entry) {
    if (!entry) return RD_EINVAL;
    auto it = registry().find(entry);
    return it == registry().end() ? RD_EINVAL : it->second.nargs;
}

extern "C" int rd_optable_add(void* table, const char* entry, int32_t nargs, const uint64_t* words, const int32_t* stream_slots) {
    RD_CHECK_ARG(table && entry && (nargs == 0 || (words && stream_slots)), "rd_optable_add: null argument");
    Table* t = static_cast<Table*>(table);
    auto it = registry().find(entry);
    RD_CHECK_ARG(it != registry().end(), "rd_optable_add: '%s' is not a replayable entry point", entry);
    RD_CHECK_ARG(it->second.nargs == nargs, "rd_optable_add: %s takes %d arguments, got %d", entry, it->second.nargs, nargs);
    Op op{it->second.fn, static_cast<int>(t->words.size()), nargs, static_cast<int>(t->patches.size()), 0};
    for (int i = 0; i < nargs; ++i) {
        t->words.push_back(words[i]);
        if (stream_slots[i] >= 0) {
            t->patches.emplace_back(op.first_word + i, stream_slots[i]);
            if (stream_slots[i] > t->max_slot) t->max_slot = stream_slots[i];
            ++op.n_patch;
        }
    }
    t->ops.push_back(op);
    return static_cast<int>(t->ops.size()) - 1;
}

extern "C" int rd_optable_size(const void* table) { return table ? static_cast<int>(static_cast<const Table*>(table)->ops.size()) : RD_EINVAL; }

extern "C" int rd_optable_set_word(void* table, int32_t op, int32_t arg, uint64_t word) {
    Table* t = static_cast<Table*>(table);
    RD_CHECK_ARG(t && op >= 0 && op < static_cast<int>(t->ops.size()) && arg >= 0 && arg < t->ops[op].nargs, "rd_optable_set_word: out of range");
    t->words[t->ops[op].first_word + arg] = word;
    return RD_OK;
}

extern "C" int rd_optable_run(void* table, int32_t begin, int32_t end, void* const* streams, int32_t n_streams, int32_t* failed_op) {
    Table* t = static_cast<Table*>(table);
    RD_CHECK_ARG(t != nullptr, "rd_optable_run: null table");
    RD_CHECK_ARG(begin >= 0 && begin <= end && end <= static_cast<int>(t->ops.size()), "rd_optable_run: range [%d, %d) outside the table's %d ops",
                 begin, end, static_cast<int>(t->ops.size()));
    RD_CHECK_ARG(n_streams > t->max_slot && (streams != nullptr || t->max_slot < 0), "rd_optable_run: the table uses stream slot %d, %d streams given",
                 t->max_slot, n_streams);
    // stream slots are patched into a per-call copy of the op's words (<= 48 of them): the table itself is never written, so one
    // table may be replayed from several threads / streams at once and a failed run leaves no half-patched state behind
    constexpr int MAX_ARGS = 48;
    uint64_t local[MAX_ARGS];
    for (int i = begin; i < end; ++i) {
        const Op& op = t->ops[i];
        const uint64_t* words = t->words.data() + op.first_word;
        if (op.n_patch > 0) {
            RD_CHECK_ARG(op.nargs <= MAX_ARGS, "rd_optable_run: op %d has %d arguments", i, op.nargs);
            memcpy(local, words, sizeof(uint64_t) * op.nargs);
            for (int p = op.first_patch; p < op.first_patch + op.n_patch; ++p)
                local[t->patches[p].first - op.first_word] = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(streams[t->patches[p].second]));
            words = local;
        }
        const int rc = op.fn(words);
        if (rc != 0) {
            if (failed_op) *failed_op = i;
            return rc;
        }
    }
    return RD_OK;
}
