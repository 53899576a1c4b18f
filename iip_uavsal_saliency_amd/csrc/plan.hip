// Native launch plan: the host side records the ~150 kernel launches of one
// UAVSal.forward once (descriptors with resolved device pointers), then replays them
// from C++ -- either as a plain launch loop or as one captured hipGraph -- so the per-call
// Python / ctypes overhead is a single call.  Also hosts the hipEvent timing used by
// bench.py (events recorded on the same stream the kernels are launched on).
#include <vector>
#include <new>
#include "common.h"

namespace {
enum OpKind { OP_CONV = 0, OP_DW, OP_STEM, OP_BILINEAR, OP_TDIFF, OP_TSUM, OP_LAYOUT, OP_GUARD, OP_COPY, OP_FUSED_IR, OP_FORK, OP_JOIN,
              OP_WINO_IN, OP_WINO_OUT, OP_DW_DOT, OP_FILL };
constexpr int MAX_LANES = 8;

// Lanes: lane 0 is the caller's stream; lanes 1..7 are private streams on which independent
// branches of the forward (prior nets, ASPP branches, the temporal branch of an STBlock) run
// concurrently with the main chain.  OP_FORK(l): lane l waits for everything recorded on lane 0 so
// far; OP_JOIN(l): lane 0 waits for lane l.  Under stream capture the same event waits become
// graph edges, so the captured hipGraph has parallel branches.
struct Op {
    int kind;
    int lane;
    union {
        uavsal_conv_desc conv;
        uavsal_dw_desc dw;
        uavsal_stem_desc stem;
        uavsal_bilinear_desc bil;
        uavsal_tdiff_desc td;
        uavsal_tsum_desc ts;
        uavsal_layout_desc lay;
        uavsal_guard_desc guard;
        uavsal_copy_desc copy;
        uavsal_fused_ir_desc fir;
        uavsal_wino_desc wino;
        uavsal_dw_dot_desc dwdot;
        uavsal_fill_desc fill;
    } u;
};

int run_op(const Op& op, uavsal_stream_t s) {
    switch (op.kind) {
        case OP_CONV: return uavsal_conv_gemm(&op.u.conv, s);
        case OP_DW: return uavsal_dw3x3(&op.u.dw, s);
        case OP_STEM: return uavsal_stem_conv(&op.u.stem, s);
        case OP_BILINEAR: return uavsal_bilinear_ac(&op.u.bil, s);
        case OP_TDIFF: return uavsal_tdiff(&op.u.td, s);
        case OP_TSUM: return uavsal_tsum(&op.u.ts, s);
        case OP_LAYOUT: return uavsal_layout(&op.u.lay, s);
        case OP_GUARD: return uavsal_guard(&op.u.guard, s);
        case OP_COPY: return uavsal_copy_rows(&op.u.copy, s);
        case OP_FUSED_IR: return uavsal_fused_ir(&op.u.fir, s);
        case OP_WINO_IN: return uavsal_wino_input(&op.u.wino, s);
        case OP_WINO_OUT: return uavsal_wino_output(&op.u.wino, s);
        case OP_DW_DOT: return uavsal_dw3x3_dot(&op.u.dwdot, s);
        case OP_FILL: return uavsal_fill(&op.u.fill, s);
    }
    return UAVSAL_EINVAL;
}
}  // namespace

struct uavsal_plan {
    std::vector<Op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int cur_lane = 0;
    bool lanes_on = true;
    hipStream_t side[MAX_LANES] = {};
    std::vector<hipEvent_t> events;      // one per fork/join op, created on first use
    // error reporting: a device word the kernels OR UAVSAL_ERR_* into, a device-visible host mirror that the
    // guard op fills, and an event recorded after every full run so the host can wait for exactly that run
    int32_t* err_dev = nullptr;
    int32_t* err_host = nullptr;          // hipHostMalloc'ed (mapped)
    int32_t* err_host_dev = nullptr;      // its device address
    hipEvent_t done = nullptr;
    bool ran = false;
};

extern "C" uavsal_plan* uavsal_plan_create(void) { return new (std::nothrow) uavsal_plan(); }

// the error words are allocated on first use, so that a plan can be created (and inspected) without a device
static bool ensure_error_words(uavsal_plan* p) {
    if (p->err_dev && p->err_host && p->done) return true;
    bool ok = hipMalloc((void**)&p->err_dev, 64) == hipSuccess && hipMemset(p->err_dev, 0, 64) == hipSuccess &&
              hipHostMalloc((void**)&p->err_host, 64, hipHostMallocMapped) == hipSuccess;
    if (ok) {
        p->err_host[0] = 0;
        ok = hipHostGetDevicePointer((void**)&p->err_host_dev, p->err_host, 0) == hipSuccess &&
             hipEventCreateWithFlags(&p->done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    return ok;
}

extern "C" int32_t* uavsal_plan_error_word(uavsal_plan* p) { return (p && ensure_error_words(p)) ? p->err_dev : nullptr; }

extern "C" int uavsal_plan_add_guard(uavsal_plan* p, float* b0, int64_t n0, float* b1, int64_t n1, float* b2, int64_t n2) {
    if (!p) return UAVSAL_EINVAL;
    if (p->exec) return UAVSAL_ESTATE;
    if (!ensure_error_words(p)) return UAVSAL_ESTATE;
    Op op; op.kind = OP_GUARD; op.lane = 0;
    op.u.guard.err = p->err_dev; op.u.guard.host_err = p->err_host_dev;
    op.u.guard.buf[0] = b0; op.u.guard.n[0] = n0;
    op.u.guard.buf[1] = b1; op.u.guard.n[1] = n1;
    op.u.guard.buf[2] = b2; op.u.guard.n[2] = n2;
    for (int b = 0; b < 3; ++b)
        if ((op.u.guard.buf[b] == nullptr) != (op.u.guard.n[b] == 0) || op.u.guard.n[b] < 0) return UAVSAL_EINVAL;
    p->ops.push_back(op);
    return (int)p->ops.size() - 1;
}

extern "C" int uavsal_plan_status(uavsal_plan* p, int wait) {
    if (!p) return UAVSAL_EINVAL;
    if (!p->ran) return 0;
    if (wait) {
        const hipError_t e = hipEventSynchronize(p->done);
        if (e != hipSuccess) return (int)e;
    } else {
        const hipError_t e = hipEventQuery(p->done);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
        if (e != hipSuccess) return (int)e;
    }
    const int32_t code = *(volatile int32_t*)p->err_host;
    if (code == 0) return 0;
    // rare path: clear both words (the run is over: the event has completed) and report once
    p->err_host[0] = 0;
    const hipError_t e = hipMemset(p->err_dev, 0, 64);
    if (e != hipSuccess) return (int)e;
    return UAVSAL_EDEVICE;
}

extern "C" void uavsal_plan_destroy(uavsal_plan* p) {
    if (!p) return;
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    for (hipEvent_t e : p->events) if (e) hipEventDestroy(e);
    for (int i = 1; i < MAX_LANES; ++i) if (p->side[i]) hipStreamDestroy(p->side[i]);
    if (p->done) hipEventDestroy(p->done);
    if (p->err_host) hipHostFree(p->err_host);
    if (p->err_dev) hipFree(p->err_dev);
    delete p;
}

extern "C" int uavsal_plan_set_lane(uavsal_plan* p, int lane) {
    if (!p || lane < 0 || lane >= MAX_LANES) return UAVSAL_EINVAL;
    p->cur_lane = lane;
    return 0;
}

extern "C" int uavsal_plan_enable_lanes(uavsal_plan* p, int on) {
    if (!p) return UAVSAL_EINVAL;
    if (p->exec) return UAVSAL_ESTATE;
    p->lanes_on = on != 0;
    return 0;
}

static int add_sync(uavsal_plan* p, int kind, int lane) {
    if (!p || lane < 1 || lane >= MAX_LANES) return UAVSAL_EINVAL;
    if (p->exec) return UAVSAL_ESTATE;
    Op op; op.kind = kind; op.lane = lane;
    p->ops.push_back(op);
    return (int)p->ops.size() - 1;
}
extern "C" int uavsal_plan_add_fork(uavsal_plan* p, int lane) { return add_sync(p, OP_FORK, lane); }
extern "C" int uavsal_plan_add_join(uavsal_plan* p, int lane) { return add_sync(p, OP_JOIN, lane); }

#define UAVSAL_ADD(fn, KIND, field, T)                                   \
    extern "C" int fn(uavsal_plan* p, const T* d) {                      \
        if (!p || !d) return UAVSAL_EINVAL;                              \
        if (p->exec) return UAVSAL_ESTATE;                               \
        Op op; op.kind = KIND; op.lane = p->cur_lane; op.u.field = *d;   \
        p->ops.push_back(op);                                            \
        return (int)p->ops.size() - 1;                                   \
    }
UAVSAL_ADD(uavsal_plan_add_conv, OP_CONV, conv, uavsal_conv_desc)
UAVSAL_ADD(uavsal_plan_add_dw, OP_DW, dw, uavsal_dw_desc)
UAVSAL_ADD(uavsal_plan_add_stem, OP_STEM, stem, uavsal_stem_desc)
UAVSAL_ADD(uavsal_plan_add_bilinear, OP_BILINEAR, bil, uavsal_bilinear_desc)
UAVSAL_ADD(uavsal_plan_add_tdiff, OP_TDIFF, td, uavsal_tdiff_desc)
UAVSAL_ADD(uavsal_plan_add_tsum, OP_TSUM, ts, uavsal_tsum_desc)
UAVSAL_ADD(uavsal_plan_add_layout, OP_LAYOUT, lay, uavsal_layout_desc)
UAVSAL_ADD(uavsal_plan_add_copy, OP_COPY, copy, uavsal_copy_desc)
UAVSAL_ADD(uavsal_plan_add_fused_ir, OP_FUSED_IR, fir, uavsal_fused_ir_desc)
UAVSAL_ADD(uavsal_plan_add_wino_input, OP_WINO_IN, wino, uavsal_wino_desc)
UAVSAL_ADD(uavsal_plan_add_wino_output, OP_WINO_OUT, wino, uavsal_wino_desc)
UAVSAL_ADD(uavsal_plan_add_dw_dot, OP_DW_DOT, dwdot, uavsal_dw_dot_desc)
UAVSAL_ADD(uavsal_plan_add_fill, OP_FILL, fill, uavsal_fill_desc)

extern "C" int uavsal_plan_patch_ptr(uavsal_plan* p, int op, int slot, void* ptr) {
    if (!p || op < 0 || op >= (int)p->ops.size() || !ptr) return UAVSAL_EINVAL;
    if (p->exec) return UAVSAL_ESTATE;
    Op& o = p->ops[op];
    switch (o.kind) {
        case OP_CONV:
            if (!uavsal_aligned16(ptr)) return UAVSAL_EALIGN;
            if (slot == 0) o.u.conv.a = (const float*)ptr;
            else if (slot == 1) o.u.conv.out = (float*)ptr;
            else return UAVSAL_EINVAL;
            return 0;
        case OP_DW_DOT:
            if (slot == 0) { if (!uavsal_aligned16(ptr)) return UAVSAL_EALIGN; o.u.dwdot.in = (const float*)ptr; }
            else if (slot == 1) o.u.dwdot.out = (float*)ptr;
            else return UAVSAL_EINVAL;
            return 0;
        case OP_STEM:
            if (slot == 0) { o.u.stem.in = (const float*)ptr; o.u.stem.in_u8 = nullptr; }
            else if (slot == 1) { o.u.stem.in_u8 = (const uint8_t*)ptr; o.u.stem.in = nullptr; }
            else return UAVSAL_EINVAL;
            return 0;
        case OP_LAYOUT:
            if (slot == 0) o.u.lay.in = (const float*)ptr;
            else if (slot == 1) o.u.lay.out = (float*)ptr;
            else return UAVSAL_EINVAL;
            return 0;
        case OP_GUARD:
            if (slot < 0 || slot > 2 || !o.u.guard.buf[slot]) return UAVSAL_EINVAL;
            o.u.guard.buf[slot] = (float*)ptr;
            return 0;
    }
    return UAVSAL_ESHAPE;
}

extern "C" int uavsal_plan_size(const uavsal_plan* p) { return p ? (int)p->ops.size() : UAVSAL_EINVAL; }

extern "C" int uavsal_plan_run(uavsal_plan* p, int first, int last, uavsal_stream_t stream) {
    if (!p) return UAVSAL_EINVAL;
    const int n = (int)p->ops.size();
    if (last < 0 || last > n) last = n;
    if (first < 0 || first > last) return UAVSAL_EINVAL;
    // lanes are honoured only when the whole plan runs; a sub-range (per-op timing) is launched flat
    const bool lanes = p->lanes_on && first == 0 && last == n;
    hipStream_t main_s = (hipStream_t)stream;
    if (p->events.size() < p->ops.size()) p->events.resize(p->ops.size(), nullptr);
    for (int i = first; i < last; ++i) {
        const Op& op = p->ops[i];
        if (op.kind == OP_FORK || op.kind == OP_JOIN) {
            if (!lanes) continue;
            // (side lanes on high-priority queues, hipStreamCreateWithPriority: the whole step takes 8.8 ms instead of 4.5)
            if (!p->side[op.lane] &&
                hipStreamCreateWithFlags(&p->side[op.lane], hipStreamNonBlocking) != hipSuccess) return (int)hipGetLastError();
            if (!p->events[i] &&
                hipEventCreateWithFlags(&p->events[i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
            hipStream_t from = op.kind == OP_FORK ? main_s : p->side[op.lane];
            hipStream_t to = op.kind == OP_FORK ? p->side[op.lane] : main_s;
            hipError_t e = hipEventRecord(p->events[i], from);
            if (e == hipSuccess) e = hipStreamWaitEvent(to, p->events[i], 0);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        uavsal_stream_t s = (lanes && op.lane > 0) ? (uavsal_stream_t)p->side[op.lane] : stream;
        if (lanes && op.lane > 0 && !p->side[op.lane]) return UAVSAL_ESTATE;    // op on a lane that was never forked
        const int e = run_op(op, s);
        if (e) return e;
    }
    if (last == n) {       // the run that ends the plan (its guard op) -- whole, or the second part of a run issued in two ranges
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main_s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone && p->done) {     // (a captured run is marked by uavsal_plan_graph_launch)
            const hipError_t e = hipEventRecord(p->done, main_s);
            if (e != hipSuccess) return (int)e;
            p->ran = true;
        }
    }
    return 0;
}

extern "C" int uavsal_plan_graph_build(uavsal_plan* p, uavsal_stream_t stream) {
    if (!p) return UAVSAL_EINVAL;
    if (p->exec) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return (int)e;
    const int r = uavsal_plan_run(p, 0, -1, stream);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(s, &g);
    if (r) { if (g) hipGraphDestroy(g); return r; }
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(g); return (int)e; }
    p->graph = g; p->exec = ex;
    return 0;
}

extern "C" int uavsal_plan_graph_launch(uavsal_plan* p, uavsal_stream_t stream) {
    if (!p) return UAVSAL_EINVAL;
    if (!p->exec) return UAVSAL_ESTATE;
    hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    if (e == hipSuccess && p->done) {
        e = hipEventRecord(p->done, (hipStream_t)stream);
        if (e == hipSuccess) p->ran = true;
    }
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int uavsal_plan_time(uavsal_plan* p, int first, int last, int iters, uavsal_stream_t stream, float* ms) {
    if (!p || !ms || iters <= 0) return UAVSAL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return (int)hipGetLastError();
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return (int)hipGetLastError(); }
    int r = 0;
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && !r; ++i) r = uavsal_plan_run(p, first, last, stream);
    hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (r) return r;
    if (e != hipSuccess) return (int)e;
    *ms = t / iters;
    return 0;
}

extern "C" int uavsal_abi_version(void) { return UAVSAL_ABI_VERSION; }

extern "C" int uavsal_sizeof_desc(int which) {
    switch (which) {
        case 0: return (int)sizeof(uavsal_conv_desc);
        case 1: return (int)sizeof(uavsal_dw_desc);
        case 2: return (int)sizeof(uavsal_stem_desc);
        case 3: return (int)sizeof(uavsal_bilinear_desc);
        case 4: return (int)sizeof(uavsal_tdiff_desc);
        case 5: return (int)sizeof(uavsal_tsum_desc);
        case 6: return (int)sizeof(uavsal_layout_desc);
        case 7: return (int)sizeof(uavsal_post_desc);
        case 8: return (int)sizeof(uavsal_guard_desc);
        case 9: return (int)sizeof(uavsal_copy_desc);
        case 10: return (int)sizeof(uavsal_fused_ir_desc);
        case 11: return (int)sizeof(uavsal_wino_desc);
        case 12: return (int)sizeof(uavsal_dw_dot_desc);
        case 13: return (int)sizeof(uavsal_fill_desc);
    }
    return UAVSAL_EINVAL;
}

extern "C" const char* uavsal_build_info(void) {
#define UAVSAL_STR2(x) #x
#define UAVSAL_STR(x) UAVSAL_STR2(x)
    return "libuavsal_hip gfx950 abi " UAVSAL_STR(UAVSAL_ABI_VERSION) " (" __DATE__ " " __TIME__ ")";
}
