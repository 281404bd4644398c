// Op-table replay: one C-ABI call issues a whole pre-marshalled range of C-ABI launches.
//
// The training step of the hot path (reference main.py:416-445: forward, loss, backward, SGD) is a static list of ~500 C-ABI
// calls over fixed buffers (radar_depth_amd/engine.py).  Issuing it from a Python `for` loop costs a ctypes marshalling per call
// and keeps one host core busy per rank; with eight one-process-per-GPU ranks on a node that is the weak-scaling risk (SURVEY.md
// 8e).  A table stores, per op, the entry point and its arguments as 64-bit words (pointers and integers by value, floats as
// their bit pattern); arguments that are streams are SLOTS, patched from the `streams` array of each rd_optable_run call, so a
// table outlives stream rebinding and hipGraph capture.  rd_optable_run does nothing but call the same entry points the Python
// loop would -- there is no second implementation of any op.
#include <atomic>
#include <condition_variable>
#include <memory>
#include <string>
#include <thread>
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
        RD_ENTRY(rd_wino_conv3x3), RD_ENTRY(rd_wino_pack_batched), RD_ENTRY(rd_wino_conv3x3_bnbwd), RD_ENTRY(rd_gconv_split_bnbwd), RD_ENTRY(rd_gconv_split_pre_bnbwd),
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
    // multi-threaded issue (rd_optable_run_mt): per op the lane (= issuing thread) and the op that must have been ISSUED before it
    // (a hipStreamWaitEvent captures the event's most recent record at call time: the record has to be on its stream first)
    std::vector<int> lane, dep;
    int mt_lanes = 0, mt_begin = -1, mt_end = -1;
};

// ---- worker pool of rd_optable_run_mt: lanes 1.. are persistent threads parked on a condition variable between runs
struct MtJob {
    Table* t = nullptr;
    int begin = 0, end = 0, lanes = 0, device = 0;
    void* const* streams = nullptr;
    std::vector<std::atomic<int>>* issued = nullptr;      // per lane: index of the last op it has issued (begin - 1 at start)
    std::atomic<int> rc{0};
    std::atomic<int> failed{-1};
    char err[512] = "";        // rd_last_error() of the failing lane (the error text is thread-local: the caller's thread re-raises it)
};
struct MtPool {
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> threads;
    MtJob* job = nullptr;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;
    ~MtPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_start.notify_all();
        for (auto& th : threads) if (th.joinable()) th.join();
    }
};
MtPool& pool() { static MtPool p; return p; }

// issue the ops of one lane of [begin, end) in table order
void run_lane(MtJob& j, int lane) {
    Table* t = j.t;
    constexpr int MAX_ARGS = 48;
    uint64_t local[MAX_ARGS];
    std::vector<std::atomic<int>>& issued = *j.issued;
    for (int i = j.begin; i < j.end; ++i) {
        if (t->lane[i] != lane) continue;
        if (j.rc.load(std::memory_order_relaxed) != 0) break;
        const int d = t->dep[i];
        if (d >= j.begin && t->lane[d] != lane) {
            // the matching event record lives on another lane: wait until that lane has issued it
            std::atomic<int>& other = issued[t->lane[d]];
            int spins = 0;
            while (other.load(std::memory_order_acquire) < d) {
                if (j.rc.load(std::memory_order_relaxed) != 0) return;
                if (++spins > 64) std::this_thread::yield();
            }
        }
        const Op& op = t->ops[i];
        const uint64_t* words = t->words.data() + op.first_word;
        if (op.n_patch > 0) {
            memcpy(local, words, sizeof(uint64_t) * op.nargs);
            for (int p = op.first_patch; p < op.first_patch + op.n_patch; ++p)
                local[t->patches[p].first - op.first_word] = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(j.streams[t->patches[p].second]));
            words = local;
        }
        const int rc = op.fn(words);
        if (rc != 0) {
            int expect = 0;
            if (j.rc.compare_exchange_strong(expect, rc)) {
                snprintf(j.err, sizeof(j.err), "%s", rd_last_error());
                j.failed.store(i);
            }
            // release anybody waiting on this lane
            issued[lane].store(j.end, std::memory_order_release);
            return;
        }
        issued[lane].store(i, std::memory_order_release);
    }
    issued[lane].store(j.end, std::memory_order_release);
}

void worker_main(int lane) {
    MtPool& p = pool();
    uint64_t seen = 0;
    for (;;) {
        MtJob* j;
        {
            std::unique_lock<std::mutex> lk(p.mu);
            p.cv_start.wait(lk, [&] { return p.stop || (p.generation != seen && p.job && lane < p.job->lanes); });
            if (p.stop) return;
            seen = p.generation;
            j = p.job;
        }
        (void)hipSetDevice(j->device);
        run_lane(*j, lane);
        {
            std::lock_guard<std::mutex> lk(p.mu);
            if (--p.pending == 0) p.cv_done.notify_all();
        }
    }
}

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

// Multi-threaded issue of [begin, end): every op goes to the lane of its stream slot (slot s -> lane s % n_lanes; ops without a
// stream argument -> lane 0); each lane issues its ops in table order from its own host thread, so a stream's program order is the
// table's.  The one cross-lane rule: an rd_stream_wait_event is issued only after the LAST rd_event_record of the same event that
// precedes it in the table has been issued by its lane (hipStreamWaitEvent binds to the event's most recent record).  Why: the HIP
// launch path costs ~10-16 us of host time per op on this stack and a training step is 480-960 ops on three streams -- one host core
// per rank issues 15 of config 4's 18.5 ms (profiles/r05_host_time.txt).  Not for use under stream capture.
extern "C" int rd_optable_run_mt(void* table, int32_t begin, int32_t end, void* const* streams, int32_t n_streams, int32_t n_lanes, int32_t* failed_op) {
    Table* t = static_cast<Table*>(table);
    RD_CHECK_ARG(t != nullptr, "rd_optable_run_mt: null table");
    RD_CHECK_ARG(begin >= 0 && begin <= end && end <= static_cast<int>(t->ops.size()), "rd_optable_run_mt: range [%d, %d) outside the table's %d ops",
                 begin, end, static_cast<int>(t->ops.size()));
    RD_CHECK_ARG(n_streams > t->max_slot && (streams != nullptr || t->max_slot < 0), "rd_optable_run_mt: the table uses stream slot %d, %d streams given",
                 t->max_slot, n_streams);
    RD_CHECK_ARG(n_lanes >= 1 && n_lanes <= 8, "rd_optable_run_mt: 1..8 lanes");
    if (n_lanes == 1 || end - begin < 8) return rd_optable_run(table, begin, end, streams, n_streams, failed_op);
    static const Thunk f_record = registry().at("rd_event_record").fn, f_wait = registry().at("rd_stream_wait_event").fn;
    if (t->mt_lanes != n_lanes || t->mt_begin != begin || t->mt_end != end) {
        // lanes and dependences of this range (cached: a step replays the same range)
        const int n = static_cast<int>(t->ops.size());
        t->lane.assign(n, 0);
        t->dep.assign(n, -1);
        std::unordered_map<uint64_t, int> last_record;
        for (int i = begin; i < end; ++i) {
            const Op& op = t->ops[i];
            RD_CHECK_ARG(op.nargs <= 48, "rd_optable_run_mt: op %d has %d arguments", i, op.nargs);
            int slot = -1;
            for (int p = op.first_patch; p < op.first_patch + op.n_patch; ++p) slot = t->patches[p].second;
            t->lane[i] = slot < 0 ? 0 : slot % n_lanes;
            const uint64_t* w = t->words.data() + op.first_word;
            if (op.fn == f_record) last_record[w[0]] = i;
            else if (op.fn == f_wait) {
                auto it = last_record.find(w[1]);
                if (it != last_record.end()) t->dep[i] = it->second;
            }
        }
        t->mt_lanes = n_lanes; t->mt_begin = begin; t->mt_end = end;
    }
    std::vector<std::atomic<int>> issued(n_lanes);
    for (auto& a : issued) a.store(begin - 1);
    MtJob job;
    job.t = t; job.begin = begin; job.end = end; job.lanes = n_lanes; job.streams = streams; job.issued = &issued;
    if (hipGetDevice(&job.device) != hipSuccess) job.device = 0;
    MtPool& p = pool();
    {
        std::lock_guard<std::mutex> lk(p.mu);
        while (static_cast<int>(p.threads.size()) < n_lanes - 1) {
            const int lane = static_cast<int>(p.threads.size()) + 1;
            p.threads.emplace_back(worker_main, lane);
        }
        p.job = &job;
        p.pending = n_lanes - 1;
        ++p.generation;
    }
    p.cv_start.notify_all();
    run_lane(job, 0);
    {
        std::unique_lock<std::mutex> lk(p.mu);
        p.cv_done.wait(lk, [&] { return p.pending == 0; });
        p.job = nullptr;
    }
    const int rc = job.rc.load();
    if (rc != 0) {
        if (failed_op) *failed_op = job.failed.load();
        rd::set_error("%s", job.err);
    }
    return rc;
}
