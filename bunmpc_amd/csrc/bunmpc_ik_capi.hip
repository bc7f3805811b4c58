// C-ABI for the rigid-body model, InverseKinematics and KinoDynMP (include/bunmpc.h), host side.
// Mirrors (behaviour, not code) ISL/include/ik/inverse_kinematics.hpp + ISL/src/ik/*.cpp and
// ISL/src/motion_planner/kino_dyn.cpp; all numerics run in ik_ddp.hip / biconvex_admm.hip.
#include "../../include/bunmpc.h"
#include "ik_types.h"
#include "id_types.h"
#include "perturb_types.h"

namespace bunmpc {
int launch_wb_plan(const RobotModelDev *model, const bmpc_wb_plan_batch_t &d, hipStream_t st);   // plan_gen.hip
}

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

namespace {

int ik_fail(int code, const std::string &msg) { return bunmpc::set_error(code, msg); }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return ik_fail(BMPC_DEVICE_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct Dev {
    void *p = nullptr; size_t bytes = 0;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    double *d() const { return static_cast<double *>(p); }
};

}  // namespace

struct bmpc_model {
    bunmpc::RobotModelDev host;
    Dev dev;            // device copy, uploaded lazily
    bool uploaded = false;
    std::mutex upload_lock;     // batches of several host threads may share one model (bunmpc_amd/pipeline.py)
    int upload() {
        std::lock_guard<std::mutex> hold(upload_lock);
        if (uploaded) return BMPC_OK;
        HIP_TRY(dev.ensure(sizeof(host)));
        HIP_TRY(hipMemcpy(dev.p, &host, sizeof(host), hipMemcpyHostToDevice));
        uploaded = true;
        return BMPC_OK;
    }
    const bunmpc::RobotModelDev *dptr() const { return static_cast<const bunmpc::RobotModelDev *>(dev.p); }
};

// one residual cost of a node, as the reference's CostModelSum holds it (name -> item)
struct IkItem { int kind; double wt; int frame; std::vector<double> ref; };  // kind: 0 frame 1 com 2 mom 3 state 4 ctrl

struct bmpc_ik {
    const bmpc_model *model;
    int n_col;
    std::vector<double> dt;
    std::vector<std::map<std::string, IkItem>> nodes;   // n_col + 1 (last = terminal)
    // results of the last optimize
    std::vector<double> xs, us;
    int iters = 0, status = 0;
    double cost = 0, stop = 0;
    bool solved = false;
    Dev din, dws, dactive;
};

struct bmpc_kinodyn {
    const bmpc_model *model;
    double m; int n_eff, dyn_col, ik_col;
    bmpc_biconvex_t *dyn; bmpc_ik *ik;
    double wt_com = 0, wt_mom = 0;
    bool profile = false;
    double solve_times[3] = {0, 0, 0};
    Dev dtmp;
};

namespace {

int add_item(bmpc_ik *h, int t, const char *name, IkItem item) {
    if (t < 0 || t > h->n_col) return ik_fail(BMPC_BAD_ARG, "time step out of range");
    auto &m = h->nodes[t];
    const std::string key = name ? name : "";
    if (m.count(key)) {   // crocoddyl CostModelSum::addCost refuses duplicates with this message
        std::cout << "Warning: we couldn't add the " << key << " cost item, it already existed." << std::endl;
        return BMPC_OK;
    }
    m[key] = std::move(item);
    return BMPC_OK;
}

// pack the accumulated costs into the kernel's per-node task blocks
int pack_tasks(const bmpc_ik *h, std::vector<double> &tasks) {
    using namespace bunmpc;
    const int nn = h->n_col + 1;
    tasks.assign((size_t)nn * kNodeTaskDoubles, 0.0);
    for (int t = 0; t < nn; ++t) {
        double *tk = tasks.data() + (size_t)t * kNodeTaskDoubles;
        int slots = 0, ncom = 0, nmom = 0, nst = 0, nct = 0;
        for (const auto &kv : h->nodes[t]) {
            const IkItem &it = kv.second;
            switch (it.kind) {
            case 0:
                if (slots == kFrameSlots) return ik_fail(BMPC_BAD_ARG, "more than 4 frame-translation costs on one node");
                tk[5 * slots] = it.wt; tk[5 * slots + 1] = it.frame;
                for (int c = 0; c < 3; ++c) tk[5 * slots + 2 + c] = it.ref[c];
                ++slots; break;
            case 1:
                if (ncom++) return ik_fail(BMPC_BAD_ARG, "more than one CoM cost on one node");
                tk[5 * kFrameSlots] = it.wt;
                for (int c = 0; c < 3; ++c) tk[5 * kFrameSlots + 1 + c] = it.ref[c];
                break;
            case 2:
                if (nmom++) return ik_fail(BMPC_BAD_ARG, "more than one momentum cost on one node");
                tk[5 * kFrameSlots + 4] = it.wt;
                for (int c = 0; c < 6; ++c) tk[5 * kFrameSlots + 5 + c] = it.ref[c];
                break;
            case 3:
                if (nst++) return ik_fail(BMPC_BAD_ARG, "more than one state regularisation on one node");
                tk[5 * kFrameSlots + 11] = it.wt; break;
            case 4:
                if (nct++) return ik_fail(BMPC_BAD_ARG, "more than one control regularisation on one node");
                tk[5 * kFrameSlots + 12] = it.wt; break;
            }
        }
    }
    return BMPC_OK;
}

// the pieces of the scratch behind bmpc_ik_batch_t.active_list (ik_types.h::active_list_ints)
void set_list(bunmpc::IkBatchArgs &a, int *p) {
    using namespace bunmpc;
    a.list = p; a.count = a.list + 2 * (long)a.B; a.wcount = a.count + 2; a.wide = a.wcount + 2; a.err = a.wide + 2 * kWideMax;
    a.near = a.err + 2; a.xmeta = a.near + 2; a.xlist = a.xmeta + 4;
}
std::atomic<double> g_express_near{1.0};   // the express lane's trigger: |Q_u|^2 below this = "within reach of the stopping threshold"

bunmpc::IkBatchArgs make_args(int B, int T, int maxiter, const bmpc_model *model, const double *x0, const double *dt,
                              const double *tasks, const double *state_w, long s_sw, const double *x_reg,
                              const double *ctrl_w, long s_cw, double *ws, int *active) {
    bunmpc::IkBatchArgs a;
    a.B = B; a.T = T; a.maxiter = maxiter; a.model = model->dptr();
    a.x0 = x0; a.dt = dt; a.tasks = tasks; a.state_w = state_w; a.x_reg = x_reg; a.ctrl_w = ctrl_w;
    a.s_state_w = s_sw; a.s_ctrl_w = s_cw; a.ws = ws; a.active = active;
    a.s_x_reg = bunmpc::kNX; a.sn_state_w = a.sn_x_reg = a.sn_ctrl_w = 0; a.fwd_spec = 0; a.bwd_waves = 1;
    a.list = nullptr; a.count = nullptr; a.wide = nullptr; a.wcount = nullptr; a.err = nullptr; a.near = nullptr; a.xmeta = nullptr; a.xlist = nullptr;
    a.iter = 0; a.n_launch = B; a.near_stop = g_express_near.load();
    return a;
}

// the DDP iteration loop (SolverDDP::solve): three launches per iteration, stop when every problem is done
// below this many active problems the forward pass runs four step lengths of a problem side by side (one wave per
// problem: 1024 SIMDs on an MI355X)
// Process-wide DEFAULTS of the scheduling thresholds (the bmpc_ik_set_* entry points); a batch may carry its own in
// bmpc_ik_batch_t.sched.  A DDP loop reads them once, when it starts: a setter called meanwhile (another host thread) changes
// the next loop, never one that is running.
std::atomic<int> g_spec_line_search_below{1024};
std::atomic<int> g_spec_one_wave_above{0};      // (experiment) above this many active problems the speculative line search runs on ONE wave per problem; 0 = never
std::atomic<int> g_all_steps{0};  // at most this many active problems: all ten step lengths at once, three workgroups per problem
                                  // (0 = never, the default: measured on the MI355X it gains < 1 % on the Go2 H = 60 batch at <= 85 -- one workgroup of
                                  // three waves per CU is the forward kernel's residency -- and loses 2 % on Solo12, EXPERIMENTS.md 9)
std::atomic<int> g_gains_wave_below{512};   // at most this many active problems: the backward pass gives each a second wave for the gains
                                  // (two waves per problem on the MI355X's 1024 SIMDs; no effect on results)
std::atomic<int> g_blocking_waits{1};       // the DDP loop's host waits sleep on an interrupt (hipEventBlockingSync) instead of spinning
constexpr int kMaxIkCol = 255;    // (T + 1 <= 64 nodes: also the fused kernel and the express lane, whose per-node flags sit in LDS; longer horizons run the four lock-step kernels only)
constexpr int kMaxFusedCol = 63;

// thresholds of ONE DDP loop: field of bmpc_ik_batch_t.sched (0 = the process default, < 0 = never, n > 0 = n)
struct Sched { int spec_below, all_steps, gains_wave_below; int debug_inject = 0; int express_cap = 0; int fused_direct = 0; };
int sched_pick(int field, const std::atomic<int> &dflt) { return field == 0 ? dflt.load() : field < 0 ? 0 : field; }
std::atomic<int> g_express_cap{96};         // the express lane takes at most this many problems of a batch (0 = no express lane)
std::atomic<int> g_fused_direct{16};        // batches of at most this many problems run entirely inside the fused kernel (0 = never)
Sched default_sched() { return Sched{g_spec_line_search_below.load(), g_all_steps.load(), g_gains_wave_below.load(), 0, g_express_cap.load(), g_fused_direct.load()}; }

// Two host-mapped words and events per (device, stream), through which the kernels' active counter reaches the DDP loop.
// Keyed by the stream, not by the host thread: a stream's publishes are ordered among themselves, so a late publish of one
// batch can never overwrite the counter of another batch running on a different stream; and the few entries (one per stream
// ever used) live as long as the library, whatever threads come and go (bunmpc_amd/pipeline.py starts workers per call).
struct ActiveWord {
    int *host[2] = {nullptr, nullptr}, *dev[2] = {nullptr, nullptr};      // eight ints each: active, index-check code, list length, express taken, iterations (fused-direct)
    hipStream_t side = nullptr;                 // the express lane's stream (ik_fused_kernel beside the batch's own kernels)
    hipEvent_t x_go = nullptr, x_done = nullptr;
    // evs[0]: spinning waits, evs[1]: blocking waits (hipEventBlockingSync: the waiting host thread sleeps until the interrupt --
    // with eight ranks of up to three pool threads each on a 16-core cgroup, spinning waits would fight the other ranks' loops)
    hipEvent_t evs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    std::mutex in_use;      // one DDP loop at a time per stream (two host threads driving one stream would interleave anyway)
    int ensure() {
        if (host[0]) return BMPC_OK;
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&host[k]), 8 * sizeof(int), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev[k]), host[k], 0));
            HIP_TRY(hipEventCreateWithFlags(&evs[0][k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&evs[1][k], hipEventDisableTiming | hipEventBlockingSync));
        }
        // A stream of the HIGHEST priority: the HIP runtime multiplexes streams of one priority onto a few hardware queues, and
        // sharing one with the caller's stream would put the lane's persistent kernel (milliseconds) IN FRONT of the batch's own
        // kernels instead of beside them (seen in bench.py, whose process holds a dozen streams: 19.8 ms per batch solve instead
        // of 15.1); queues are per priority level, and the lane's workgroups should be placed first anyway.
        int pr_least = 0, pr_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&side, hipStreamNonBlocking, pr_greatest) != hipSuccess) {
            (void)hipGetLastError();        // no priorities on this runtime: an ordinary stream still gives correct results
            HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        }
        HIP_TRY(hipEventCreateWithFlags(&x_go, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&x_done, hipEventDisableTiming | hipEventBlockingSync));
        return BMPC_OK;
    }
};
std::mutex g_active_words_lock;
std::map<std::pair<int, hipStream_t>, ActiveWord *> &active_words() {
    static auto *m = new std::map<std::pair<int, hipStream_t>, ActiveWord *>;   // never destroyed: the HIP runtime may be gone at exit
    return *m;
}
ActiveWord *active_word_for(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> hold(g_active_words_lock);
    ActiveWord *&w = active_words()[{dev, st}];
    if (!w) w = new ActiveWord;
    return w;
}

// bmpc_ik_set_profile(1): hipEvents around every kernel of the DDP loop; the summed times per kernel of the last batch
// solve are read back with bmpc_ik_last_profile.  For measurement passes only (the events cost launch slots).
bool g_profile = false;
double g_last_profile[5] = {0, 0, 0, 0, 0};     // ms: state, calcdiff, backward, forward, everything else in the loop
std::mutex g_profile_lock;

int index_check_failed(int code) {
    static const char *what[] = {"", "active-list entry out of range", "active-list length out of range", "active-list append past its end",
                                 "wide-list entry out of range", "the fused kernel's tick watchdog fired"};
    return ik_fail(BMPC_DEVICE_ERROR, std::string("IK-DDP index check failed: ") + what[code > 0 && code < 6 ? code : 0] +
                   " (code " + std::to_string(code) + "); results of this batch are invalid");
}

int run_ddp(const bunmpc::IkBatchArgs &a0, hipStream_t st, int *iters_run, const Sched sched) {
    bunmpc::IkBatchArgs a = a0;
    a.fwd_spec = 0;
    const bool prof = g_profile;
    std::vector<hipEvent_t> pev;
    auto stamp = [&]() -> int {
        if (!prof) return BMPC_OK;
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(e, st));
        pev.push_back(e);
        return BMPC_OK;
    };
    ActiveWord *wp = active_word_for(st);
    if (!wp) return ik_fail(BMPC_DEVICE_ERROR, "hipGetDevice failed");
    ActiveWord &w = *wp;
    std::lock_guard<std::mutex> hold(w.in_use);
    if (int rc = w.ensure()) return rc;
    // Every return below (the error ones too) leaves with the side stream idle: work queued there reads xmeta / xlist, which the
    // next batch on this stream rewrites as soon as w.in_use is free.
    struct SideDrain {
        hipStream_t side; bool armed = false;
        ~SideDrain() { if (armed) (void)hipStreamSynchronize(side); }
    } side_drain{w.side};
    hipEvent_t *ev = w.evs[g_blocking_waits.load() ? 1 : 0];
    HIP_TRY(bunmpc::ik_launch_init(a, st));
    if (sched.debug_inject == 1 && a.list) HIP_TRY(hipMemsetAsync(a.list, 0x7f, sizeof(int), st));      // tests: an entry far out of range
    // A handful of problems (the single-problem handles of the drop-in classes above all): every one of them gets a CU of its own
    // from the first iteration on -- the whole DDP in ONE launch of the fused kernel, no host look in between.
    if (a.list && a.B <= sched.fused_direct && a.maxiter > 0 && a.T <= kMaxFusedCol) {
        a.iter = 0; a.n_launch = a.B;
        HIP_TRY(bunmpc::ik_launch_fused_tail(a, st));
        HIP_TRY(bunmpc::ik_launch_publish_active(a, 0, w.dev[0], st, a.B));
        HIP_TRY(hipEventRecord(ev[0], st));
        HIP_TRY(hipEventSynchronize(ev[0]));
        const volatile int *hw = static_cast<volatile int *>(w.host[0]);
        if (iters_run) *iters_run = hw[4];        // the most iterations any problem of the batch ran (one launch: there is no host loop to count)
        if (prof) {                               // no per-kernel split exists on this path: the profile of an earlier batch must not stand
            std::lock_guard<std::mutex> hold_p(g_profile_lock);
            for (int k = 0; k < 5; ++k) g_last_profile[k] = 0.0;
        }
        if (hw[1]) return index_check_failed(hw[1]);
        if (hw[0] != 0) return ik_fail(BMPC_DEVICE_ERROR, "the fused kernel left problems unsolved");
        return BMPC_OK;
    }
    // The host looks at the active counter after every iteration while many problems are iterating (iterations are long
    // there and the line-search mapping depends on it); once few are left it enqueues kTailChunk iterations per look --
    // kernels of a finished problem return at once, so an iteration too many costs a few microseconds.  And it looks
    // one chunk late: the next chunk is enqueued BEFORE the host waits for the counter of the one before, so the queue
    // never drains while the host turns around (at the end one chunk of no-op kernels runs out on its own).
    constexpr int kTailChunk = 3;
    // The express lane (ik_select_kernel / ik_fused_kernel): in front of the early iterations a one-workgroup kernel looks at the
    // batch and -- once, when it finds it converging fast with a thin tail of laggards -- moves the laggards-to-be off the active
    // list; the fused kernel enqueued behind it on the side stream runs them to the end at their own pace, on CUs of their own,
    // while the batch goes on without them.  The decision is the device's (the host would learn of it an iteration late); the
    // host enqueues the pair until a look tells it that the lane has taken its problems, or the window has passed.
    constexpr int kExpressFirstIter = 2, kExpressLastIter = 12;
    const int express_cap = a.list && a.B >= 64 && a.T <= kMaxFusedCol ? sched.express_cap : 0;
    bool express_taken = false;
    int express_enqueued = 0;
    int active = a.B, it = 0, it_end[2] = {0, 0};
    auto enqueue_chunk = [&](int slot) -> int {
        const int chunk = active <= sched.spec_below ? kTailChunk : 1;
        a.fwd_spec = active <= sched.all_steps ? 4 : active <= sched.spec_below / 3 ? 3 : active <= sched.spec_below ? 2 : 0;
        if (a.fwd_spec == 2 && g_spec_one_wave_above.load() > 0 && active > g_spec_one_wave_above.load()) a.fwd_spec = 1;
        a.bwd_waves = active <= sched.gains_wave_below ? 2 : 1;
        a.n_launch = active;        // the host's latest look at the counter: an upper bound of the active list's length
        for (int k = 0; k < chunk && it < a.maxiter; ++k, ++it) {
            a.iter = it;
            if (express_cap > 0 && !express_taken && it >= kExpressFirstIter && it <= kExpressLastIter) {
                HIP_TRY(bunmpc::ik_launch_select(a, express_cap, sched.debug_inject == 2, st));
                HIP_TRY(hipEventRecord(w.x_go, st));
                HIP_TRY(hipStreamWaitEvent(w.side, w.x_go, 0));
                side_drain.armed = true;
                HIP_TRY(bunmpc::ik_launch_fused_express(a, express_cap, w.side));
                HIP_TRY(hipEventRecord(w.x_done, w.side));
                ++express_enqueued;
            }
            if (int rc = stamp()) return rc;
            HIP_TRY(bunmpc::ik_launch_state(a, st));
            if (int rc = stamp()) return rc;
            HIP_TRY(bunmpc::ik_launch_calcdiff(a, st));
            if (int rc = stamp()) return rc;
            HIP_TRY(bunmpc::ik_launch_backward(a, st));
            if (int rc = stamp()) return rc;
            HIP_TRY(bunmpc::ik_launch_forward(a, st));
            if (int rc = stamp()) return rc;
        }
        HIP_TRY(bunmpc::ik_launch_publish_active(a, it, w.dev[slot], st));
        HIP_TRY(hipEventRecord(ev[slot], st));
        it_end[slot] = it;
        return BMPC_OK;
    };
    int slot = 0, it_done = 0, index_err = 0, all_active = a.B;
    if (a.maxiter > 0 && active > 0) {
        if (int rc = enqueue_chunk(slot)) return rc;
        for (;;) {
            const bool more = it < a.maxiter;
            if (more) { if (int rc = enqueue_chunk(slot ^ 1)) return rc; }
            HIP_TRY(hipEventSynchronize(ev[slot]));
            const volatile int *hw = static_cast<volatile int *>(w.host[slot]);
            all_active = hw[0];
            index_err = hw[1];
            active = hw[2] >= 0 ? hw[2] : hw[0];     // what the batch's own kernels still run over (the express lane's problems are not on the list)
            express_taken = express_taken || hw[3] != 0;
            it_done = it_end[slot];
            if (index_err) active = 0;       // an index of the list code was out of range: stop, report below
            if (active <= 0 || !more) {
                // the chunk enqueued ahead (no-op kernels and one more publish into the other word) must have drained before
                // this stream's words can serve another batch
                if (more) HIP_TRY(hipEventSynchronize(ev[slot ^ 1]));
                break;
            }
            slot ^= 1;
        }
    }
    if (iters_run) *iters_run = it_done;     // iterations up to the look that found every problem done (not the chunk enqueued ahead)
    if (express_enqueued > 0) {
        // the lane's kernel ends when its last problem has: the caller's stream is ordered behind it, and one more look tells
        // whether it left an error code (or a problem) behind
        HIP_TRY(hipStreamWaitEvent(st, w.x_done, 0));
        HIP_TRY(bunmpc::ik_launch_publish_active(a, it, w.dev[0], st));
        HIP_TRY(hipEventRecord(ev[0], st));
        HIP_TRY(hipEventSynchronize(ev[0]));
        const volatile int *hw = static_cast<volatile int *>(w.host[0]);
        all_active = hw[0];
        if (!index_err) index_err = hw[1];
        side_drain.armed = false;      // the look above came behind x_done: the side stream has drained
    }
    if (index_err) return index_check_failed(index_err);
    if (prof && !pev.empty()) {
        HIP_TRY(hipEventSynchronize(pev.back()));
        double acc[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i + 1 < pev.size(); ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, pev[i], pev[i + 1]));
            acc[i % 5 < 4 ? i % 5 : 4] += ms;     // five stamps per iteration: the fifth interval is the gap to the next iteration
        }
        for (hipEvent_t e : pev) (void)hipEventDestroy(e);
        std::lock_guard<std::mutex> hold(g_profile_lock);
        for (int k = 0; k < 5; ++k) g_last_profile[k] = acc[k];
    }
    return BMPC_OK;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------ model ----
bmpc_model_t *bmpc_model_create(int nj, const int *parent, const double *R, const double *p, const double *axis,
                                const double *mass, const double *com, const double *inertia, int nframes,
                                const int *frame_body, const double *frame_p) {
    using namespace bunmpc;
    if (nj != kMaxJoints) { ik_fail(BMPC_BAD_ARG, "the kernels are built for 12 revolute joints after the free-flyer"); return nullptr; }
    if (nframes < 0 || nframes > kMaxFrames) { ik_fail(BMPC_BAD_ARG, "too many frames"); return nullptr; }
    auto *m = new bmpc_model;
    RobotModelDev &h = m->host;
    std::memset(&h, 0, sizeof(h));
    h.nj = nj; h.nframes = nframes;
    for (int i = 0; i < nj; ++i) {
        h.parent[i] = parent[i];
        if (parent[i] >= i || parent[i] < -1) { delete m; ik_fail(BMPC_BAD_ARG, "joints must be listed parents first"); return nullptr; }
        std::memcpy(h.R[i], R + 9 * i, sizeof(double) * 9);
        h.R_identity[i] = 1;
        for (int c = 0; c < 9; ++c) if (R[9 * i + c] != (c % 4 == 0 ? 1.0 : 0.0)) h.R_identity[i] = 0;
        std::memcpy(h.p[i], p + 3 * i, sizeof(double) * 3);
        std::memcpy(h.axis[i], axis + 3 * i, sizeof(double) * 3);
    }
    // serial chains only: every joint has at most one child, chain_end = last descendant
    for (int i = 0; i < nj; ++i) {
        int nchild = 0;
        for (int j = 0; j < nj; ++j) nchild += parent[j] == i;
        if (nchild > 1) { delete m; ik_fail(BMPC_BAD_ARG, "branching below the base is not supported"); return nullptr; }
    }
    for (int i = nj - 1; i >= 0; --i) {
        h.chain_end[i] = i;
        for (int j = i + 1; j < nj; ++j) if (parent[j] == i) h.chain_end[i] = h.chain_end[j];
    }
    for (int i = 0; i < nj; ++i)   // the subtree of joint i must be the contiguous index range i..chain_end[i]
        for (int j = i + 1; j <= h.chain_end[i]; ++j)
            if (parent[j] != j - 1) { delete m; ik_fail(BMPC_BAD_ARG, "chains must be numbered contiguously"); return nullptr; }
    // the register-resident kinematics (rbd_quad.h) assume 4 legs of 3 joints: 3L hangs off the base, 3L+j off 3L+j-1
    for (int i = 0; i < nj; ++i)
        if (parent[i] != (i % 3 == 0 ? -1 : i - 1)) { delete m; ik_fail(BMPC_BAD_ARG, "expected 4 legs x 3 joints, numbered leg by leg"); return nullptr; }
    h.total_mass = 0;
    for (int b = 0; b <= nj; ++b) {
        h.mass[b] = mass[b]; h.total_mass += mass[b];
        std::memcpy(h.com[b], com + 3 * b, sizeof(double) * 3);
        const double *I = inertia + 9 * b;
        h.inertia[b][0] = I[0]; h.inertia[b][1] = I[1]; h.inertia[b][2] = I[2];
        h.inertia[b][3] = I[4]; h.inertia[b][4] = I[5]; h.inertia[b][5] = I[8];
        double *r = h.rec[b];
        for (int c = 0; c < 3; ++c) { r[c] = b > 0 ? h.p[b - 1][c] : 0.0; r[3 + c] = b > 0 ? h.axis[b - 1][c] : 0.0; r[7 + c] = h.com[b][c]; }
        r[6] = h.mass[b];
        for (int c = 0; c < 6; ++c) r[10 + c] = h.inertia[b][c];
    }
    for (int f = 0; f < nframes; ++f) {
        if (frame_body[f] < 0 || frame_body[f] > nj) { delete m; ik_fail(BMPC_BAD_ARG, "frame body out of range"); return nullptr; }
        h.frame_body[f] = frame_body[f];
        std::memcpy(h.frame_p[f], frame_p + 3 * f, sizeof(double) * 3);
    }
    return m;
}
void bmpc_model_destroy(bmpc_model_t *m) { delete m; }

int bmpc_wb_plan_batch_device(const bmpc_wb_plan_batch_t *d, void *hip_stream) {
    if (!d || !d->model) return ik_fail(BMPC_BAD_ARG, "null descriptor or model");
    if (d->B < 0 || d->n_col < 1 || d->ik_col < 1 || d->ik_col > d->n_col) return ik_fail(BMPC_BAD_ARG, "bad sizes");
    if (!d->gait || !d->x || !d->t0 || !d->v_des_body) return ik_fail(BMPC_BAD_ARG, "missing input array");
    if (!d->com || !d->feet0 || !d->v_des || !d->w_des || !d->hip_off || !d->amom || !d->x_init || !d->cnt_plan || !d->swing_time ||
        !d->dt || !d->X_nom || !d->X_ter || !d->ik_tasks)
        return ik_fail(BMPC_BAD_ARG, "missing output array");
    for (int j = 0; j < 4; ++j)
        if (d->foot_frame[j] < 0 || d->foot_frame[j] >= d->model->host.nframes) return ik_fail(BMPC_BAD_ARG, "foot frame out of range");
    if (d->B == 0) return BMPC_OK;
    bmpc_model *m = const_cast<bmpc_model *>(d->model);
    if (int rc = m->upload()) return rc;
    return bunmpc::launch_wb_plan(m->dptr(), *d, static_cast<hipStream_t>(hip_stream));
}
int bmpc_id_batch_device(const bmpc_id_batch_t *d, void *hip_stream) {
    using namespace bunmpc;
    if (!d || !d->model) return ik_fail(BMPC_BAD_ARG, "null descriptor or model");
    if (d->n < 0) return ik_fail(BMPC_BAD_ARG, "n < 0");
    if (d->n > 0 && (!d->q_des || !d->v_des || !d->a_des || !d->f)) return ik_fail(BMPC_BAD_ARG, "missing desired-state / force array");
    if ((d->q == nullptr) != (d->v == nullptr)) return ik_fail(BMPC_BAD_ARG, "q and v must be given together");
    if (d->n > 0 && !d->tau_ff && !d->tau_fb && !d->action && !d->state) return ik_fail(BMPC_BAD_ARG, "no output array");
    for (int i = 0; i < 12; ++i)
        if (d->action && d->kp[i] == 0.0) return ik_fail(BMPC_BAD_ARG, "kp = 0 with the pd_target action asked for");
    IdLaunch a;
    a.d = *d;
    if (!d->q) { a.d.q = d->q_des; a.d.v = d->v_des; a.d.s_q = d->s_q_des; a.d.s_v = d->s_v_des; }
    const RobotModelDev &h = d->model->host;
    for (int L = 0; L < 4; ++L) { a.leg_foot[L] = -1; a.leg_foot_k[L] = 0; a.leg_foot_p[L][0] = a.leg_foot_p[L][1] = a.leg_foot_p[L][2] = 0.0; }
    for (int j = 0; j < 4; ++j) {
        const int f = d->foot_frame[j];
        if (f < 0 || f >= h.nframes) return ik_fail(BMPC_BAD_ARG, "foot frame out of range");
        const int body = h.frame_body[f];
        if (body < 1) return ik_fail(BMPC_BAD_ARG, "an end effector on the base: each must hang off a leg");
        const int L = (body - 1) / 3;
        if (a.leg_foot[L] >= 0) return ik_fail(BMPC_BAD_ARG, "two end effectors on one leg");
        a.leg_foot[L] = j; a.leg_foot_k[L] = (body - 1) % 3;
        for (int c = 0; c < 3; ++c) a.leg_foot_p[L][c] = h.frame_p[f][c];
    }
    if (d->n == 0) return BMPC_OK;
    bmpc_model *m = const_cast<bmpc_model *>(d->model);
    if (int rc = m->upload()) return rc;
    a.model = m->dptr();
    return launch_id_batch(a, static_cast<hipStream_t>(hip_stream));
}

int bmpc_perturb_batch_device(const bmpc_perturb_batch_t *d, void *hip_stream) {
    using namespace bunmpc;
    if (!d || !d->model) return ik_fail(BMPC_BAD_ARG, "null descriptor or model");
    if (d->B < 0 || d->K < 1) return ik_fail(BMPC_BAD_ARG, "B < 0 or K < 1");
    for (int j = 0; j < 4; ++j)
        if (d->foot_frame[j] < 0 || d->foot_frame[j] >= d->model->host.nframes) return ik_fail(BMPC_BAD_ARG, "foot frame out of range");
    for (int g = 0; g < 4; ++g)
        if (!(d->sigma[g] >= 0.0)) return ik_fail(BMPC_BAD_ARG, "negative or NaN sigma");
    if (d->B == 0) return BMPC_OK;
    if (!d->q || !d->v || !d->contact || !d->z) return ik_fail(BMPC_BAD_ARG, "missing input array");
    if (!d->q_out || !d->v_out || !d->chosen) return ik_fail(BMPC_BAD_ARG, "missing output array");
    bmpc_model *m = const_cast<bmpc_model *>(d->model);
    if (int rc = m->upload()) return rc;
    PerturbLaunch a;
    a.d = *d;
    a.model = m->dptr();
    return launch_perturb(a, static_cast<hipStream_t>(hip_stream));
}

int bmpc_ik_set_profile(int on) { const int old = g_profile; g_profile = on != 0; return old; }
void bmpc_ik_last_profile(double *ms5) {
    std::lock_guard<std::mutex> hold(g_profile_lock);
    for (int k = 0; k < 5; ++k) ms5[k] = g_last_profile[k];
}
// self test of the state operators (host arrays): x0, x1 [n][37], dx [n][36] -> diff(x0, x1) and x0 (+) dx by the quaternion
// versions the forward pass uses (dq, iq) and by the rotation-matrix versions (dr, ir)
int bmpc_ik_selftest_state_ops(const double *x0, const double *x1, const double *dx, int n, double *dq, double *dr, double *iq, double *ir) {
    using namespace bunmpc;
    if (!x0 || !x1 || !dx || !dq || !dr || !iq || !ir || n < 1) return ik_fail(BMPC_BAD_ARG, "bad selftest arguments");
    Dev in, out;
    const size_t nx = (size_t)n * kNX, nd = (size_t)n * kNDX;
    HIP_TRY(in.ensure(sizeof(double) * (2 * nx + nd)));
    HIP_TRY(out.ensure(sizeof(double) * (2 * nx + 2 * nd)));
    HIP_TRY(hipMemcpy(in.d(), x0, sizeof(double) * nx, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(in.d() + nx, x1, sizeof(double) * nx, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(in.d() + 2 * nx, dx, sizeof(double) * nd, hipMemcpyHostToDevice));
    HIP_TRY(ik_launch_state_ops_selftest(in.d(), in.d() + nx, in.d() + 2 * nx, n, out.d(), out.d() + nd, out.d() + 2 * nd, out.d() + 2 * nd + nx, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dq, out.d(), sizeof(double) * nd, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dr, out.d() + nd, sizeof(double) * nd, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(iq, out.d() + 2 * nd, sizeof(double) * nx, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ir, out.d() + 2 * nd + nx, sizeof(double) * nx, hipMemcpyDeviceToHost));
    return BMPC_OK;
}
int bmpc_ik_set_all_steps(int n_active) { return g_all_steps.exchange(n_active); }
int bmpc_ik_set_blocking_waits(int on) { return g_blocking_waits.exchange(on != 0); }
int bmpc_ik_set_express_capacity(int n) { return g_express_cap.exchange(n); }
int bmpc_ik_set_fused_direct_max(int n) { return g_fused_direct.exchange(n); }
double bmpc_ik_set_express_near(double stop) { return g_express_near.exchange(stop); }
void bmpc_ik_kernel_occupancy(int *out8) { bunmpc::ik_kernel_occupancy(out8); }
int bmpc_ik_batch_struct_size(void) { return (int)sizeof(bmpc_ik_batch_t); }
int bmpc_ik_set_speculative_below(int n_active) { return g_spec_line_search_below.exchange(n_active); }
int bmpc_ik_set_spec_one_wave_above(int n_active) { return g_spec_one_wave_above.exchange(n_active); }
double bmpc_model_total_mass(const bmpc_model_t *m) { return m ? m->host.total_mass : 0.0; }

// ----------------------------------------------------------- InverseKinematics ----
bmpc_ik_t *bmpc_ik_create(const bmpc_model_t *model, int n_col) {
    if (!model || n_col < 1) { ik_fail(BMPC_BAD_ARG, "bad InverseKinematics arguments"); return nullptr; }
    if (n_col > kMaxIkCol) { ik_fail(BMPC_BAD_ARG, "n_col > 255 is not supported"); return nullptr; }
    auto *h = new bmpc_ik;
    h->model = model; h->n_col = n_col;
    h->dt.assign(n_col, 0.0);
    h->nodes.resize(n_col + 1);
    return h;
}
void bmpc_ik_destroy(bmpc_ik_t *h) { delete h; }
int bmpc_ik_n_col(const bmpc_ik_t *h) { return h ? h->n_col : 0; }

int bmpc_ik_setup_costs(bmpc_ik_t *h, const double *dt, int n) {   // inverse_kinematics.cpp:37-52
    if (!h || !dt) return ik_fail(BMPC_BAD_ARG, "null argument");
    if (n < h->n_col) return ik_fail(BMPC_BAD_ARG, "dt must have n_col entries");
    h->dt.assign(dt, dt + h->n_col);
    return BMPC_OK;
}

#define IK_CHECK(...)                                                              \
    if (!h) return ik_fail(BMPC_BAD_ARG, "null handle");                           \
    { const void *ptrs_[] = {__VA_ARGS__};                                         \
      for (const void *p_ : ptrs_) if (!p_) return ik_fail(BMPC_BAD_ARG, "null array argument"); }

static int check_frame(const bmpc_ik *h, int frame) {
    if (frame < 0 || frame >= h->model->host.nframes) return ik_fail(BMPC_BAD_ARG, "frame id out of range");
    return BMPC_OK;
}

int bmpc_ik_add_position_tracking_task(bmpc_ik_t *h, int frame, int sn, int en, const double *traj3, double wt, const char *name) {
    IK_CHECK(traj3);
    if (int rc = check_frame(h, frame)) return rc;
    for (int i = sn; i < en; ++i) {   // end_effector_tasks.cpp:8-19: name + std::to_string(i)
        const std::string nm = std::string(name ? name : "") + std::to_string(i);
        if (int rc = add_item(h, i, nm.c_str(), IkItem{0, wt, frame, {traj3[0], traj3[1], traj3[2]}})) return rc;
    }
    return BMPC_OK;
}
int bmpc_ik_add_position_tracking_task_single(bmpc_ik_t *h, int frame, const double *traj3, double wt, const char *name, int time_step) {
    IK_CHECK(traj3);
    if (int rc = check_frame(h, frame)) return rc;
    if (time_step < 0 || time_step >= h->n_col) return ik_fail(BMPC_BAD_ARG, "time step out of range");
    return add_item(h, time_step, name, IkItem{0, wt, frame, {traj3[0], traj3[1], traj3[2]}});
}
int bmpc_ik_add_terminal_position_tracking_task(bmpc_ik_t *h, int frame, const double *traj3, double wt, const char *name) {
    IK_CHECK(traj3);
    if (int rc = check_frame(h, frame)) return rc;
    return add_item(h, h->n_col, name, IkItem{0, wt, frame, {traj3[0], traj3[1], traj3[2]}});
}
int bmpc_ik_add_velocity_tracking_task(bmpc_ik_t *h) {   // end_effector_tasks.cpp:49-56: a stub in the reference too
    IK_CHECK(h);
    std::cout << "function not implemented" << std::endl;
    return BMPC_OK;
}
static int add_traj(bmpc_ik *h, int kind, int width, int sn, int en, const double *traj, int rows, double wt, const char *name, int is_terminal) {
    if (!is_terminal) {
        if (sn < 0 || en > h->n_col || rows < en - sn) return ik_fail(BMPC_BAD_ARG, "trajectory shorter than the node range");
        for (int i = sn; i < en; ++i) {
            std::vector<double> r(traj + (size_t)(i - sn) * width, traj + (size_t)(i - sn + 1) * width);
            if (int rc = add_item(h, i, name, IkItem{kind, wt, -1, r})) return rc;
        }
        return BMPC_OK;
    }
    if (rows < 1) return ik_fail(BMPC_BAD_ARG, "empty trajectory");
    return add_item(h, h->n_col, name, IkItem{kind, wt, -1, std::vector<double>(traj, traj + width)});
}
int bmpc_ik_add_com_position_tracking_task(bmpc_ik_t *h, int sn, int en, const double *traj, int rows, double wt, const char *name, int is_terminal) {
    IK_CHECK(traj);
    return add_traj(h, 1, 3, sn, en, traj, rows, wt, name, is_terminal);   // com_tasks.cpp:8-28
}
int bmpc_ik_add_centroidal_momentum_tracking_task(bmpc_ik_t *h, int sn, int en, const double *traj, int rows, double wt, const char *name, int is_terminal) {
    IK_CHECK(traj);
    return add_traj(h, 2, 6, sn, en, traj, rows, wt, name, is_terminal);   // com_tasks.cpp:30-50
}
// regularisation items carry their own vectors (ref = [stateWeights 36 | x_reg 37] or [controlWeights 18]): the acyclic
// generator gives every node its own (abstract_acyclic_gen.py:225-290)
static IkItem state_item(double wt, const double *w, const double *xr) {
    IkItem it{3, wt, -1, std::vector<double>(w, w + bunmpc::kNDX)};
    it.ref.insert(it.ref.end(), xr, xr + bunmpc::kNX);
    return it;
}
static IkItem ctrl_item(double wt, const double *w) { return IkItem{4, wt, -1, std::vector<double>(w, w + bunmpc::kNV)}; }
int bmpc_ik_add_state_regularization_cost(bmpc_ik_t *h, int sn, int en, double wt, const char *name, const double *w36, const double *xreg37, int is_terminal) {
    IK_CHECK(w36, xreg37);   // regularization_costs.cpp:8-36
    if (is_terminal) return add_item(h, h->n_col, name, state_item(wt, w36, xreg37));
    if (sn < 0 || en > h->n_col) return ik_fail(BMPC_BAD_ARG, "node range out of bounds");
    for (int i = sn; i < en; ++i) if (int rc = add_item(h, i, name, state_item(wt, w36, xreg37))) return rc;
    return BMPC_OK;
}
int bmpc_ik_add_state_regularization_cost_single(bmpc_ik_t *h, int time_step, double wt, const char *name, const double *w36, const double *xreg37) {
    IK_CHECK(w36, xreg37);
    if (time_step < 0 || time_step >= h->n_col) return ik_fail(BMPC_BAD_ARG, "time step out of range");
    return add_item(h, time_step, name, state_item(wt, w36, xreg37));
}
int bmpc_ik_add_ctrl_regularization_cost(bmpc_ik_t *h, int sn, int en, double wt, const char *name, const double *w18, const double *ureg18, int is_terminal) {
    IK_CHECK(w18);   // regularization_costs.cpp:66-93; u_reg is ignored there (ResidualModelControl(state_))
    (void)ureg18;
    if (is_terminal) return add_item(h, h->n_col, name, ctrl_item(wt, w18));
    if (sn < 0 || en > h->n_col) return ik_fail(BMPC_BAD_ARG, "node range out of bounds");
    for (int i = sn; i < en; ++i) if (int rc = add_item(h, i, name, ctrl_item(wt, w18))) return rc;
    return BMPC_OK;
}
int bmpc_ik_add_ctrl_regularization_cost_single(bmpc_ik_t *h, int time_step, double wt, const char *name, const double *w18, const double *ureg18) {
    IK_CHECK(w18);
    (void)ureg18;
    if (time_step < 0 || time_step >= h->n_col) return ik_fail(BMPC_BAD_ARG, "time step out of range");
    return add_item(h, time_step, name, ctrl_item(wt, w18));
}

int bmpc_ik_workspace_doubles(int n_col) { return (int)bunmpc::IkLayout::make(n_col).total; }

// InverseKinematics::optimize (inverse_kinematics.cpp:54-71): ShootingProblem + SolverDDP::solve(), then fresh cost sums
int bmpc_ik_optimize(bmpc_ik_t *h, const double *x0) {
    using namespace bunmpc;
    IK_CHECK(x0);
    auto *model = const_cast<bmpc_model *>(h->model);
    const int T = h->n_col, nn = T + 1;
    std::vector<double> tasks;
    if (int rc = pack_tasks(h, tasks)) return rc;     // host-side validation first: its errors need no GPU
    if (int rc = model->upload()) return rc;
    const IkLayout L = IkLayout::make(T);
    // per-node regularisation vectors (nodes without the cost keep a neutral reference; their weight is 0 anyway)
    std::vector<double> sw((size_t)nn * kNDX, 0.0), xr((size_t)nn * kNX, 0.0), cw((size_t)nn * kNV, 0.0);
    for (int t = 0; t < nn; ++t) {
        xr[(size_t)t * kNX + 6] = 1.0;
        for (const auto &kv : h->nodes[t]) {
            const IkItem &it = kv.second;
            if (it.kind == 3) {
                std::copy(it.ref.begin(), it.ref.begin() + kNDX, sw.begin() + (size_t)t * kNDX);
                std::copy(it.ref.begin() + kNDX, it.ref.end(), xr.begin() + (size_t)t * kNX);
            } else if (it.kind == 4) std::copy(it.ref.begin(), it.ref.end(), cw.begin() + (size_t)t * kNV);
        }
    }
    // staging: x0 | dt | tasks | state_w | x_reg | ctrl_w
    std::vector<double> stage;
    auto push = [&](const double *p, size_t n) { size_t o = stage.size(); stage.insert(stage.end(), p, p + n); return o; };
    const size_t o_x0 = push(x0, kNX), o_dt = push(h->dt.data(), T), o_tk = push(tasks.data(), tasks.size()),
                 o_sw = push(sw.data(), sw.size()), o_xr = push(xr.data(), xr.size()), o_cw = push(cw.data(), cw.size());
    HIP_TRY(h->din.ensure(sizeof(double) * stage.size()));
    HIP_TRY(h->dws.ensure(sizeof(double) * (size_t)L.total));
    HIP_TRY(h->dactive.ensure(sizeof(int) * (1 + active_list_ints(1))));     // the counter + a one-problem active list (index checks, fused kernel)
    HIP_TRY(hipMemcpy(h->din.p, stage.data(), sizeof(double) * stage.size(), hipMemcpyHostToDevice));
    const double *d = h->din.d();
    IkBatchArgs a = make_args(1, T, 100, model, d + o_x0, d + o_dt, d + o_tk, d + o_sw, 0, d + o_xr, d + o_cw, 0, h->dws.d(),
                              static_cast<int *>(h->dactive.p));
    a.sn_state_w = kNDX; a.sn_x_reg = kNX; a.sn_ctrl_w = kNV;
    set_list(a, static_cast<int *>(h->dactive.p) + 1);
    if (int rc = run_ddp(a, nullptr, nullptr, default_sched())) return rc;
    h->xs.resize((size_t)nn * kNX); h->us.resize((size_t)T * kNV);
    double scal[16];
    HIP_TRY(hipMemcpy(h->xs.data(), h->dws.d() + L.xs, sizeof(double) * h->xs.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->us.data(), h->dws.d() + L.us, sizeof(double) * h->us.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(scal, h->dws.d() + L.scal, sizeof(scal), hipMemcpyDeviceToHost));
    h->iters = (int)scal[S_ITERS]; h->status = (int)scal[S_STATUS]; h->cost = scal[S_COST]; h->stop = scal[S_STOP];
    h->solved = true;
    for (auto &m : h->nodes) m.clear();   // rcost_arr_[i] / tcost_model_ replaced by empty CostModelSums
    return BMPC_OK;
}
int bmpc_ik_get_xs(const bmpc_ik_t *h, double *xs) {
    IK_CHECK(xs);
    if (!h->solved) return ik_fail(BMPC_BAD_ARG, "optimize has not been called");
    std::memcpy(xs, h->xs.data(), sizeof(double) * h->xs.size());
    return BMPC_OK;
}
int bmpc_ik_get_us(const bmpc_ik_t *h, double *us) {
    IK_CHECK(us);
    if (!h->solved) return ik_fail(BMPC_BAD_ARG, "optimize has not been called");
    std::memcpy(us, h->us.data(), sizeof(double) * h->us.size());
    return BMPC_OK;
}
static int com_mom(bmpc_ik *h, double *com, double *mom) {
    using namespace bunmpc;
    if (!h->solved) return ik_fail(BMPC_BAD_ARG, "optimize has not been called");
    const int nn = h->n_col + 1;
    Dev out;
    HIP_TRY(out.ensure(sizeof(double) * nn * 9));
    const IkLayout L = IkLayout::make(h->n_col);
    HIP_TRY(ik_launch_com_mom(h->model->dptr(), h->dws.d() + L.xs, out.d(), out.d() + nn * 3, nn, nullptr));
    if (com) HIP_TRY(hipMemcpy(com, out.d(), sizeof(double) * nn * 3, hipMemcpyDeviceToHost));
    if (mom) HIP_TRY(hipMemcpy(mom, out.d() + nn * 3, sizeof(double) * nn * 6, hipMemcpyDeviceToHost));
    return BMPC_OK;
}
int bmpc_ik_return_opt_com(bmpc_ik_t *h, double *com) { IK_CHECK(com); return com_mom(h, com, nullptr); }   // :73-82
int bmpc_ik_return_opt_mom(bmpc_ik_t *h, double *mom) { IK_CHECK(mom); return com_mom(h, nullptr, mom); }   // :84-93
int bmpc_ik_last_stats(const bmpc_ik_t *h, int *iters, int *status, double *cost, double *stop) {
    IK_CHECK(h);
    if (iters) *iters = h->iters;
    if (status) *status = h->status;
    if (cost) *cost = h->cost;
    if (stop) *stop = h->stop;
    return BMPC_OK;
}

// batch: many independent IK problems, everything on the device
int bmpc_ik_solve_batch_device(const bmpc_ik_batch_t *d, void *hip_stream) {
    using namespace bunmpc;
    if (!d || !d->model) return ik_fail(BMPC_BAD_ARG, "null batch descriptor / model");
    if (d->B < 0 || d->n_col < 1 || d->maxiter < 1) return ik_fail(BMPC_BAD_ARG, "bad sizes");
    if (d->n_col > kMaxIkCol) return ik_fail(BMPC_BAD_ARG, "n_col > 255 is not supported");
    if (!d->x0 || !d->dt || !d->tasks || !d->state_w || !d->x_reg || !d->ctrl_w || !d->ws || !d->active)
        return ik_fail(BMPC_BAD_ARG, "missing array");
    if (d->B == 0) return BMPC_OK;
    auto *model = const_cast<bmpc_model *>(d->model);
    if (int rc = model->upload()) return rc;
    IkBatchArgs a = make_args(d->B, d->n_col, d->maxiter, model, d->x0, d->dt, d->tasks, d->state_w, d->s_state_w, d->x_reg,
                              d->ctrl_w, d->s_ctrl_w, d->ws, d->active);
    a.s_x_reg = d->s_x_reg ? d->s_x_reg : kNX;
    a.sn_state_w = d->sn_state_w; a.sn_x_reg = d->sn_x_reg; a.sn_ctrl_w = d->sn_ctrl_w;
    if (d->active_list) set_list(a, d->active_list);
    int iters = 0;
    const Sched sched{sched_pick(d->sched.spec_below, g_spec_line_search_below), sched_pick(d->sched.all_steps_below, g_all_steps),
                      sched_pick(d->sched.gains_wave_below, g_gains_wave_below), d->sched.debug_inject, sched_pick(d->sched.express_cap, g_express_cap),
                      g_fused_direct.load()};
    int rc = run_ddp(a, static_cast<hipStream_t>(hip_stream), &iters, sched);
    if (d->iters_run) *d->iters_run = iters;
    return rc;
}
int bmpc_ik_centroidal_state_device(const bmpc_model_t *model, const double *x, double *out9, int B, void *hip_stream) {
    if (!model || !x || !out9) return ik_fail(BMPC_BAD_ARG, "null argument");
    auto *m = const_cast<bmpc_model *>(model);
    if (int rc = m->upload()) return rc;
    HIP_TRY(bunmpc::ik_launch_centroidal_state(m->dptr(), x, out9, B, static_cast<hipStream_t>(hip_stream)));
    return BMPC_OK;
}
void bmpc_ik_layout(int n_col, long *offsets8) {   // xs, us, scal, K, kff, fs, Lx, Lqq offsets for callers that read the workspace
    const bunmpc::IkLayout L = bunmpc::IkLayout::make(n_col);
    offsets8[0] = L.xs; offsets8[1] = L.us; offsets8[2] = L.scal; offsets8[3] = L.K; offsets8[4] = L.kff;
    offsets8[5] = L.fs; offsets8[6] = L.Lx; offsets8[7] = L.Lqq;
}

int bmpc_ik_set_gains_wave_below(int n_active) { return g_gains_wave_below.exchange(n_active); }
long bmpc_ik_active_list_ints(long B) { return bunmpc::active_list_ints(B); }

void bmpc_ik_layout_trace(int n_col, long *offset, int *iters, int *width) {   // the per-iteration telemetry rows of a problem's workspace
    const bunmpc::IkLayout L = bunmpc::IkLayout::make(n_col);
    if (offset) *offset = L.trace;
    if (iters) *iters = bunmpc::kTraceIters;
    if (width) *width = bunmpc::kTraceDoubles;
}

// batch of KinoDynMP::optimize calls, device resident: centroidal state of (q, v) -> ADMM (cold start)
// -> tracking references -> IK-DDP
int bmpc_kinodyn_solve_batch_device(const bmpc_kinodyn_batch_t *d, void *hip_stream) {
    if (!d || !d->x || !d->ik.model) return ik_fail(BMPC_BAD_ARG, "null KinoDyn batch descriptor");
    if (d->dyn.B != d->ik.B || d->ik.n_col > d->dyn.n_col) return ik_fail(BMPC_BAD_ARG, "inconsistent batch sizes / horizons");
    if (d->dyn.B == 0) return BMPC_OK;
    auto *model = const_cast<bmpc_model *>(d->ik.model);
    if (int rc = model->upload()) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(bunmpc::ik_launch_centroidal_state(model->dptr(), d->x, const_cast<double *>(d->dyn.x_init), d->dyn.B, st));
    if (int rc = bmpc_biconvex_solve_batch_device(&d->dyn, hip_stream)) return rc;
    HIP_TRY(bunmpc::ik_launch_fill_refs(const_cast<double *>(d->ik.tasks), d->dyn.X, d->dyn.m, d->dyn.B, d->dyn.n_col,
                                        d->ik.n_col, st));
    return bmpc_ik_solve_batch_device(&d->ik, hip_stream);
}

// ------------------------------------------------------------------- KinoDynMP ----
bmpc_kinodyn_t *bmpc_kinodyn_create(const bmpc_model_t *model, double m, int n_eff, int dyn_col, int ik_col) {
    if (!model || ik_col < 1 || ik_col > dyn_col) { ik_fail(BMPC_BAD_ARG, "bad KinoDynMP arguments"); return nullptr; }
    auto *h = new bmpc_kinodyn;
    h->model = model; h->m = m; h->n_eff = n_eff; h->dyn_col = dyn_col; h->ik_col = ik_col;
    h->dyn = bmpc_biconvex_create(m, dyn_col, n_eff);
    h->ik = bmpc_ik_create(model, ik_col);
    if (!h->dyn || !h->ik) { bmpc_biconvex_destroy(h->dyn); bmpc_ik_destroy(h->ik); delete h; return nullptr; }
    std::cout << "Initialized Kino-Dyn planner" << std::endl;   // kino_dyn.cpp:8
    return h;
}
void bmpc_kinodyn_destroy(bmpc_kinodyn_t *h) { if (h) { bmpc_biconvex_destroy(h->dyn); bmpc_ik_destroy(h->ik); delete h; } }
bmpc_biconvex_t *bmpc_kinodyn_return_dyn(bmpc_kinodyn_t *h) { return h ? h->dyn : nullptr; }
bmpc_ik_t *bmpc_kinodyn_return_ik(bmpc_kinodyn_t *h) { return h ? h->ik : nullptr; }
int bmpc_kinodyn_set_com_tracking_weight(bmpc_kinodyn_t *h, double w) { IK_CHECK(h); h->wt_com = w; return BMPC_OK; }
int bmpc_kinodyn_set_mom_tracking_weight(bmpc_kinodyn_t *h, double w) { IK_CHECK(h); h->wt_mom = w; return BMPC_OK; }
int bmpc_kinodyn_compute_solve_times(bmpc_kinodyn_t *h) { IK_CHECK(h); h->profile = true; return BMPC_OK; }
int bmpc_kinodyn_return_solve_times(const bmpc_kinodyn_t *h, double *t3) { IK_CHECK(t3); std::memcpy(t3, h->solve_times, sizeof(h->solve_times)); return BMPC_OK; }

// KinoDynMP::optimize (kino_dyn.cpp:39-81)
int bmpc_kinodyn_optimize(bmpc_kinodyn_t *h, const double *q, const double *v, int dyn_iters, int kino_dyn_iters) {
    using namespace bunmpc;
    IK_CHECK(q, v);
    (void)kino_dyn_iters;   // unused in the reference as well
    const auto t1 = std::chrono::steady_clock::now();
    auto *model = const_cast<bmpc_model *>(h->model);
    if (int rc = model->upload()) return rc;
    double x0[kNX];
    std::memcpy(x0, q, sizeof(double) * kNQ);
    std::memcpy(x0 + kNQ, v, sizeof(double) * kNV);
    // computeCentroidalMomentum(q, v) -> [com, vcom, hg.angular]
    HIP_TRY(h->dtmp.ensure(sizeof(double) * (kNX + 9)));
    HIP_TRY(hipMemcpy(h->dtmp.p, x0, sizeof(x0), hipMemcpyHostToDevice));
    HIP_TRY(ik_launch_centroidal_state(model->dptr(), h->dtmp.d(), h->dtmp.d() + kNX, 1, nullptr));
    double c9[9];
    HIP_TRY(hipMemcpy(c9, h->dtmp.d() + kNX, sizeof(c9), hipMemcpyDeviceToHost));
    // set_warm_starts (kino_dyn.cpp:83-99): X = tile(c9), F = 0, P = 0
    const int H = h->dyn_col, nx = 9 * (H + 1), nf = 3 * h->n_eff * H;
    std::vector<double> X(nx), F(nf, 0.0), P(nx, 0.0);
    for (int i = 0; i <= H; ++i) std::memcpy(X.data() + 9 * i, c9, sizeof(c9));
    if (int rc = bmpc_biconvex_set_warm_start_vars(h->dyn, X.data(), F.data(), P.data())) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    int rc_dyn = bmpc_biconvex_optimize(h->dyn, c9, dyn_iters);
    if (rc_dyn != BMPC_OK && rc_dyn != BMPC_DIVERGED) return rc_dyn;
    const auto t3 = std::chrono::steady_clock::now();
    std::vector<double> com((size_t)(H + 1) * 3), mom((size_t)(H + 1) * 6);
    bmpc_biconvex_return_opt_com(h->dyn, com.data());
    bmpc_biconvex_return_opt_mom(h->dyn, mom.data());
    const int T = h->ik_col;
    // kino_dyn.cpp:53-56
    if (int rc = bmpc_ik_add_centroidal_momentum_tracking_task(h->ik, 0, T, mom.data(), T, h->wt_mom, "mom_track", 0)) return rc;
    if (int rc = bmpc_ik_add_centroidal_momentum_tracking_task(h->ik, 0, T, mom.data() + 6 * T, 1, h->wt_mom, "mom_track_ter", 1)) return rc;
    if (int rc = bmpc_ik_add_com_position_tracking_task(h->ik, 0, T, com.data(), T, h->wt_com, "com_track", 0)) return rc;
    if (int rc = bmpc_ik_add_com_position_tracking_task(h->ik, 0, T, com.data() + 3 * T, 1, h->wt_com, "com_track", 1)) return rc;
    const auto t4 = std::chrono::steady_clock::now();
    if (int rc = bmpc_ik_optimize(h->ik, x0)) return rc;
    const auto t5 = std::chrono::steady_clock::now();
    if (h->profile) {
        h->solve_times[0] = std::chrono::duration<double>(t3 - t2).count();
        h->solve_times[1] = std::chrono::duration<double>(t5 - t4).count();
        h->solve_times[2] = std::chrono::duration<double>(t5 - t1).count();
        std::cout << "Dyn optimize time : " << h->solve_times[0] << std::endl;
        std::cout << "Kin optimize time : " << h->solve_times[1] << std::endl;
        std::cout << "Total optimize time : " << h->solve_times[2] << std::endl;
        std::cout << "===============================================" << std::endl;
    }
    return rc_dyn;
}

}  // extern "C"
