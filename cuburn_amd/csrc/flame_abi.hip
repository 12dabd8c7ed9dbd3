// flame_abi.hip — host side of libflame_hip.so: the C ABI of include/flame_hip.h.
//
// Holds what cuburn's RenderManager / Framebuffers / Renderer hold on the CUDA side
// (cuburn/render.py:40-170, 225-262): device buffers, the stream, persistent walker and
// RNG state, and the per-frame launch sequences of render.py:289-372 and filters.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <atomic>
#include <vector>
#include <algorithm>
#include "kernels.h"
#include "flame_device.h"

static thread_local std::string g_err;
static int fail(int code, const char *what, const char *file, int line, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    else snprintf(buf, sizeof buf, "%s (%s:%d)", what, file, line);
    g_err = buf;
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(e_ == hipErrorOutOfMemory ? FL_E_NOMEM : FL_E_HIP, #x, __FILE__, __LINE__, e_); } while (0)
#define REQUIRE(c, msg) do { if (!(c)) return fail(FL_E_INVAL, msg, __FILE__, __LINE__); } while (0)

struct EvPair { hipEvent_t a, b; };

// One "lane" = a stream with its own framebuffers, sample log and per-frame parameter buffers.
// Consecutive frames alternate between two lanes so that the drain + filter + output kernels of
// frame k (bandwidth / TA bound) overlap the iterate kernel of frame k+1 (issue / latency bound),
// the role of stream_a / stream_b in the reference (cuburn/render.py:253-262,432-433).
struct Lane {
    hipStream_t stream = nullptr;
    size_t nbins = 0;
    float4 *d_front = nullptr, *d_back = nullptr, *d_side = nullptr;
    float *d_blur = nullptr;          // 1-channel scratch [nbins]
    u64 *d_atom = nullptr;
    uint32_t *d_hot = nullptr;
    void *d_outpix = nullptr;         // w*h*8 bytes
    size_t outpix_bytes = 0;
    // binned accumulate: sample log + directory.  Two sets: in a frame of several launches the tile
    // accumulate + flush of launch k run on `aux` while launch k+1 iterates on `stream` into the other set
    // (the role of the reference's alternating streams inside a frame, cuburn/render.py:340-369)
    uint32_t *d_log[2] = {nullptr, nullptr}, *d_dir[2] = {nullptr, nullptr};
    size_t log_words[2] = {0, 0}, dir_words[2] = {0, 0};
    hipStream_t aux = nullptr;
    hipEvent_t ev_it[2] = {nullptr, nullptr}, ev_ac[2] = {nullptr, nullptr};   // iterate k queued / drains of launch k done
    float *d_params = nullptr;        // [nslots * pstride] one block per temporal sample = per walker slot (grow-only)
    uint64_t params_serial = 0;       // serial of the genome whose parameters the blocks hold (0: none / just allocated)
    size_t params_floats = 0;
    u64 *d_palette = nullptr;         // [FL_PAL_H * FL_PAL_W]
    // cross-lane ordering of the state both lanes share
    hipEvent_t ev_interp_done = nullptr;   // genome staging buffers + palette RNG states
    hipEvent_t ev_iter_done = nullptr;     // walkers + their RNG states
    hipEvent_t ev_out_done = nullptr;      // output-dither RNG states
    bool interp_rec = false, iter_rec = false, out_rec = false;
    // deferred ends of the filter chain (see flush_pending): a `yuv` that has not run yet, and the DE — all eight directions,
    // the first normalising the accumulator (pend_in_mode: 1 raw, 2 raw YUV), the last un-normalising (+ a logscale to apply
    // on the way, + a colorclip if that is the call that flushes it): run_de_finish
    bool pend_yuv = false, pend_finish = false, pend_log = false;
    float pend_k1 = 0.0f, pend_k2 = 0.0f;
    int pend_in_mode = 0;
    float pend_dp[5] = {0, 0, 0, 0, 0}, pend_k7[7] = {0, 0, 0, 0, 0, 0, 0};
    fl_dim pend_dim = {0, 0, 0, 0, 0};
};

struct fl_ctx {
    int device = 0;
    bool own_stream = false;
    static const int kMaxLanes = 4;
    int nlanes = 2, cur = 0;          // FLAME_LANES (1..4; default 2): consecutive frames go round the lanes
    Lane lanes[kMaxLanes];
    uint32_t nslots = 0, nwalkers = 0;
    uint32_t sub_log2 = 0;                        // 512 slots of 8 waves / 256 of 16: 2 / 4 temporal samples per workgroup (iter.hip "Sub-blocks of four waves")
    uint32_t ntemporal() const { return nslots << sub_log2; }      // temporal samples = parameter blocks per frame (>= FL_NTEMPORAL)
    int nw = 4;                       // waves per iterate workgroup
    fl_mwc *d_rng = nullptr;          // [nwalkers]: walkers | palette rows (64*256) | output dither (FL_NOUT)
    float4 *d_points = nullptr;       // [nslots*NT]
    u64 *d_counters = nullptr;
    uint32_t *d_sort = nullptr; size_t sort_words = 0;     // radix sort scratch (grow-only): digit counts + chunk totals
    uint32_t bin_rounds = 16, bin_parts = 0;      // bin_parts 0: chosen per image (see do_iter_launch)
    uint32_t launch_rounds = 0;                   // FLAME_LAUNCH_ROUNDS: write-enabled rounds per binned launch (0: FL_BIN_MAX_ROUNDS) — the sample log of a launch is nslots x 256 x rounds x 4 bytes
    uint32_t round_counter = 0;
    static const uint32_t kFrames = 8;            // frames that may be in flight (reference: 2)
    hipEvent_t ev_begin_[kFrames] = {}, ev_end_[kFrames] = {};
    uint32_t frame_lane[kFrames] = {};
    uint32_t frame_seq = 0;                        // id of the current frame = frame_seq - 1
    std::vector<EvPair> pool, iter_ev, accum_ev, flush_ev, filt_ev, de_ev;
    size_t pool_used = 0;
    bool timing = true;
    static const uint32_t kDepEvents = 16;
    hipEvent_t dep_ev[kDepEvents] = {};           // fl_stream_dependency
    uint32_t dep_next = 0;
    // environment switches, read once when the context is created (listed in include/flame_hip.h)
    bool env_bin_wide = false, env_no_intra = false;
    bool use_rtc = true;                    // FLAME_RTC=0: always the interpreter kernel
    uint32_t n_spec_launch = 0, n_interp_launch = 0;      // iterate launches by kernel since fl_timings_reset (fl_launch_stats)
};
#define L(c) ((c)->lanes[(c)->cur])
#define OTHER(c) ((c)->lanes[((c)->cur + (c)->nlanes - 1) % (c)->nlanes])      /* the previous frame's lane: it touched the shared state last */
#define FL_NOUT 65536u                // RNG states reserved for the output dither kernel

struct fl_genome {
    uint64_t serial = 0;                    // unique per created genome (a lane remembers whose parameters its blocks hold)
    std::vector<int32_t> prog;
    IterSpec spec;                          // structure for the run-time specialised iterate kernel (rtc.hip)
    hipFunction_t rtc_fn[5][2][4] = {};     // [nw 4 / 8 / 16 / 8 in halves / 16 in quarters][count][acc] once compiled
    unsigned rtc_epoch = 0;                 // module-cache epoch the handles above belong to
    bool rtc_failed = false;                // compile / load failed once: stay on the interpreter kernel
    uint32_t nops = 0, nrows = 0, pstride = 0;
    int32_t *d_prog = nullptr, *d_ops = nullptr;
    float *d_times = nullptr, *d_knots = nullptr, *d_ptimes = nullptr;
    float4 *d_pals = nullptr;
    uint32_t npal = 0;
    // pinned staging for asynchronous uploads: a small ring so that packing frame k+1 on the
    // host never overwrites bytes a queued copy of frame k still has to read
    static const int kStage = 4;
    unsigned char *h_stage[kStage] = {};
    hipEvent_t ev_stage[kStage] = {};
    size_t stage_bytes = 0;
    int stage_next = 0;
};

static const int kKnownVars[] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,
    34,35,36,37,38,39,40,41,42,43,44,45,46,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,
    74,75,76,77,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,97,98};

// Kernel timing keeps one event pair per launch since the last fl_timings_reset(); a long render
// that never asks for timings stops recording after kMaxTimed launches instead of growing forever.
static const size_t kMaxTimed = 8192;

static EvPair *ev_begin(fl_ctx *c, std::vector<EvPair> &list)
{
    if (!c->timing || c->pool_used >= kMaxTimed) return nullptr;
    if (c->pool.capacity() < kMaxTimed) c->pool.reserve(kMaxTimed);      // pointers into the pool are held across nested pairs: never reallocate
    if (c->pool_used == c->pool.size()) {
        EvPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        c->pool.push_back(p);
    }
    EvPair p = c->pool[c->pool_used++];
    list.push_back(p);
    hipEventRecord(p.a, L(c).stream);
    return &c->pool[c->pool_used - 1];
}
// a pair whose events are recorded by the kernel launch itself (hipExtLaunchKernelGGL)
static EvPair *ev_pair(fl_ctx *c, std::vector<EvPair> &list)
{
    if (!c->timing || c->pool_used >= kMaxTimed) return nullptr;
    if (c->pool.capacity() < kMaxTimed) c->pool.reserve(kMaxTimed);
    if (c->pool_used == c->pool.size()) {
        EvPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        c->pool.push_back(p);
    }
    list.push_back(c->pool[c->pool_used++]);
    return &c->pool[c->pool_used - 1];
}
static void ev_end(fl_ctx *c, EvPair *p) { if (p) hipEventRecord(p->b, L(c).stream); }
static EvPair *ev_begin_on(fl_ctx *c, std::vector<EvPair> &list, hipStream_t st)
{
    EvPair *p = ev_pair(c, list);
    if (p) hipEventRecord(p->a, st);
    return p;
}
static void ev_end_on(EvPair *p, hipStream_t st) { if (p) hipEventRecord(p->b, st); }

#pragma GCC visibility push(default)
extern "C" {

int fl_abi_version(void) { return FL_ABI_VERSION; }
const char *fl_last_error(void) { return g_err.c_str(); }

void fl_calc_dim(uint32_t w, uint32_t h, fl_dim *o)
{
    o->w = w; o->h = h;
    o->aw = w + 2 * FL_GUTTER;
    o->ah = 16 * ((h + 2 * FL_GUTTER + 15) / 16);
    o->astride = 32 * ((o->aw + 31) / 32);
}

static void flush_pending(fl_ctx *c);

static void free_fb(fl_ctx *c)
{
    hipFree(L(c).d_front); hipFree(L(c).d_back); hipFree(L(c).d_side); hipFree(L(c).d_blur);
    hipFree(L(c).d_atom); hipFree(L(c).d_hot); hipFree(L(c).d_outpix);
    L(c).d_front = L(c).d_back = L(c).d_side = nullptr; L(c).d_blur = nullptr; L(c).d_atom = nullptr;
    L(c).d_hot = nullptr; L(c).d_outpix = nullptr; L(c).nbins = 0; L(c).outpix_bytes = 0;
    L(c).pend_yuv = L(c).pend_finish = L(c).pend_log = false;      // whatever was deferred dies with the buffers
}

// cuburn/render.py:121-161 Framebuffers.alloc / set_dim: grow-only; on OOM free everything
// and report FL_E_NOMEM so the caller survives an oversize frame.
static int ensure_fb(fl_ctx *c, const fl_dim &d)
{
    size_t nbins = (size_t)d.ah * d.astride, ob = (size_t)d.w * d.h * 8;
    if (L(c).nbins >= nbins && L(c).outpix_bytes >= ob) return FL_OK;
    hipStreamSynchronize(L(c).stream);
    free_fb(c);
    hipError_t e;
    if ((e = hipMalloc(&L(c).d_front, 16 * nbins)) || (e = hipMalloc(&L(c).d_back, 16 * nbins)) ||
        (e = hipMalloc(&L(c).d_side, 16 * nbins)) || (e = hipMalloc(&L(c).d_blur, 4 * nbins)) ||
        (e = hipMalloc(&L(c).d_atom, 8 * nbins)) || (e = hipMalloc(&L(c).d_hot, 4 * (nbins / 16))) ||
        (e = hipMalloc(&L(c).d_outpix, ob))) {
        free_fb(c);
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? FL_E_NOMEM : FL_E_HIP, "framebuffer allocation", __FILE__, __LINE__, e);
    }
    L(c).nbins = nbins; L(c).outpix_bytes = ob;
    return FL_OK;
}

static bool env_on(const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; }

int fl_ctx_create(int device, void *stream, const fl_mwc *seeds, uint32_t nseeds, uint32_t nslots, fl_ctx **out)
{
    REQUIRE(out && seeds, "null argument");
    REQUIRE((nslots >= FL_NTEMPORAL || nslots == FL_NTEMPORAL / 2 || nslots == FL_NTEMPORAL / 4) && nslots % 256 == 0 && nslots <= 16384,
            "nslots must be a multiple of 256 in [1024, 16384] (or 512 slots of 8 waves / 256 of 16: two / four temporal samples per workgroup)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FL_E_NODEV, "no HIP device", __FILE__, __LINE__);
    REQUIRE(device >= 0 && device < ndev, "bad device index");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FL_E_NODEV, "device is not gfx950 (kernels are built for MI355X only)", __FILE__, __LINE__);
    // waves per iterate workgroup (4, 8 or 16) follow from the size of the seed table
    int nw = 0;
    {
        const uint32_t fixed = FL_PAL_H * 256 + FL_NOUT;
        const uint32_t per_wave = nslots * 64u;
        if (nseeds == fixed + 16u * per_wave) nw = 16;
        else if (nseeds == fixed + 8u * per_wave) nw = 8;
        else if (nseeds == fixed + 4u * per_wave) nw = 4;
        else return fail(FL_E_INVAL, "nseeds must be nslots*64*NW + 64*256 + 65536 with NW = 4, 8 or 16", __FILE__, __LINE__);
    }
    fl_ctx *c = new fl_ctx;
    c->device = device;
    c->nw = nw;
    if (nslots < FL_NTEMPORAL && (uint32_t)nw * nslots != 4u * FL_NTEMPORAL)
        return fail(FL_E_INVAL, "512 slots need 8-wave workgroups, 256 slots 16-wave ones (a temporal sample per four waves: 1024 in all)", __FILE__, __LINE__);
    c->nslots = nslots;
    c->sub_log2 = nslots >= FL_NTEMPORAL ? 0u : nw == 8 ? 1u : 2u;
    c->nwalkers = nslots * (uint32_t)nw * 64 + FL_PAL_H * 256 + FL_NOUT;
    if (const char *e = getenv("FLAME_LANES")) { const int v = atoi(e); c->nlanes = v >= 1 && v <= fl_ctx::kMaxLanes ? v : 2; }
    if (const char *e = getenv("FLAME_BIN_ROUNDS")) { int v = atoi(e); if (v >= 1 && v <= FL_BIN_R_MAX) c->bin_rounds = (uint32_t)v; }
    if (const char *e = getenv("FLAME_BIN_PARTS")) { int v = atoi(e); if (v >= 1 && v <= 64) c->bin_parts = (uint32_t)v; }
    c->env_bin_wide = env_on("FLAME_BIN_WIDE");
    if (const char *e = getenv("FLAME_RTC")) c->use_rtc = strcmp(e, "0") != 0;
    c->env_no_intra = env_on("FLAME_NO_INTRA_OVERLAP");      // launches of a frame strictly in series on one stream
    if (const char *e = getenv("FLAME_LAUNCH_ROUNDS")) { int v = atoi(e); if (v >= 16 && v <= 4096) c->launch_rounds = (uint32_t)(v / 16 * 16); }
    if (stream) { c->lanes[0].stream = (hipStream_t)stream; c->own_stream = false; c->nlanes = 1; }   // caller's stream: one lane
    else c->own_stream = true;
    // every failure below leaves through fl_ctx_destroy, which frees whatever exists so far
    hipError_t e = hipSuccess;
    const char *what = "context allocation";
    do {
        if (c->own_stream)
            for (int i = 0; i < c->nlanes && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking);
        if (e) break;
        if ((e = hipMalloc(&c->d_rng, sizeof(fl_mwc) * (size_t)c->nwalkers))) break;
        if ((e = hipMalloc(&c->d_points, sizeof(float4) * (size_t)nslots * nw * 64))) break;
        if ((e = hipMalloc(&c->d_counters, 8 * 4))) break;
        for (int i = 0; i < c->nlanes && e == hipSuccess; ++i) {
            Lane &ln = c->lanes[i];
            if ((e = hipMalloc(&ln.d_palette, sizeof(u64) * FL_PAL_H * FL_PAL_W))) break;
            if ((e = hipEventCreateWithFlags(&ln.ev_interp_done, hipEventDisableTiming))) break;
            if ((e = hipEventCreateWithFlags(&ln.ev_iter_done, hipEventDisableTiming))) break;
            if ((e = hipEventCreateWithFlags(&ln.ev_out_done, hipEventDisableTiming))) break;
            if ((e = hipStreamCreateWithFlags(&ln.aux, hipStreamNonBlocking))) break;
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                if ((e = hipEventCreateWithFlags(&ln.ev_it[k], hipEventDisableTiming))) break;
                e = hipEventCreateWithFlags(&ln.ev_ac[k], hipEventDisableTiming);
            }
            if (e) break;
        }
        if (e) break;
        if ((e = hipMemcpy(c->d_rng, seeds, sizeof(fl_mwc) * (size_t)c->nwalkers, hipMemcpyHostToDevice))) break;
        if ((e = hipMemsetD32(c->d_points, 0x7fc00000, (size_t)nslots * nw * 64 * 4))) break;
        if ((e = hipMemset(c->d_counters, 0, 32))) break;
        for (uint32_t i = 0; i < fl_ctx::kFrames && e == hipSuccess; ++i) {
            if ((e = hipEventCreate(&c->ev_begin_[i]))) break;
            e = hipEventCreate(&c->ev_end_[i]);
        }
    } while (0);
    if (e != hipSuccess) {
        fl_ctx_destroy(c);
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? FL_E_NOMEM : FL_E_HIP, what, __FILE__, __LINE__, e);
    }
    *out = c;
    return FL_OK;
}

static void sync_all(fl_ctx *c)
{
    for (int i = 0; i < c->nlanes; ++i) { hipStreamSynchronize(c->lanes[i].stream); if (c->lanes[i].aux) hipStreamSynchronize(c->lanes[i].aux); }
}

void fl_ctx_destroy(fl_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    for (uint32_t i = 0; i < fl_ctx::kDepEvents; ++i) if (c->dep_ev[i]) hipEventDestroy(c->dep_ev[i]);
    for (int i = 0; i < fl_ctx::kMaxLanes; ++i) {
        if (c->lanes[i].stream) hipStreamSynchronize(c->lanes[i].stream);
        if (c->lanes[i].aux) hipStreamSynchronize(c->lanes[i].aux);
    }
    for (int i = 0; i < fl_ctx::kMaxLanes; ++i) {
        c->cur = i;
        free_fb(c);
        Lane &ln = c->lanes[i];
        hipFree(ln.d_params); hipFree(ln.d_palette);
        for (int k = 0; k < 2; ++k) {
            hipFree(ln.d_log[k]); hipFree(ln.d_dir[k]);
            if (ln.ev_it[k]) hipEventDestroy(ln.ev_it[k]);
            if (ln.ev_ac[k]) hipEventDestroy(ln.ev_ac[k]);
        }
        if (ln.aux) hipStreamDestroy(ln.aux);
        if (ln.ev_interp_done) hipEventDestroy(ln.ev_interp_done);
        if (ln.ev_iter_done) hipEventDestroy(ln.ev_iter_done);
        if (ln.ev_out_done) hipEventDestroy(ln.ev_out_done);
        if (c->own_stream && ln.stream) hipStreamDestroy(ln.stream);
    }
    hipFree(c->d_rng); hipFree(c->d_points); hipFree(c->d_counters); hipFree(c->d_sort);
    for (auto &p : c->pool) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (uint32_t i = 0; i < fl_ctx::kFrames; ++i) {
        if (c->ev_begin_[i]) hipEventDestroy(c->ev_begin_[i]);
        if (c->ev_end_[i]) hipEventDestroy(c->ev_end_[i]);
    }
    (void)hipGetLastError();
    delete c;
}

int fl_ctx_sync(fl_ctx *c) { REQUIRE(c, "null ctx"); sync_all(c); return FL_OK; }

// Make the current lane's stream wait for what the OTHER lane last did to state both share.
static int wait_other(fl_ctx *c, int what)
{
    if (c->nlanes < 2) return FL_OK;
    Lane &o = OTHER(c);
    if (what == 0 && o.interp_rec) HIPCHK(hipStreamWaitEvent(L(c).stream, o.ev_interp_done, 0));
    if (what == 1 && o.iter_rec) HIPCHK(hipStreamWaitEvent(L(c).stream, o.ev_iter_done, 0));
    if (what == 2 && o.out_rec) HIPCHK(hipStreamWaitEvent(L(c).stream, o.ev_out_done, 0));
    return FL_OK;
}

// Validate the program header before any kernel trusts it (the reference traps on device,
// cuburn/code/iter.py:254-257).  Variation numbers live in the parameter block as FL_OP_CONST
// ops and are checked with the op list.
static int check_prog(const int32_t *p, uint32_t n)
{
    REQUIRE(n >= FL_PROG_HDR && p[0] == FL_PROG_MAGIC, "bad program header");
    int nxf = p[1], hf = p[2], ps = p[3], cdf = p[4], xo = p[5], xs = p[6], vs = p[7];
    REQUIRE(nxf >= 1 && nxf <= FL_MAX_XFORMS && (hf == 0 || hf == 1), "bad xform count");
    REQUIRE(ps >= 6 + nxf && ps <= FL_MAX_PSTRIDE, "bad pstride");
    REQUIRE(cdf >= 6 && cdf + nxf <= xo, "bad cdf offset");
    REQUIRE(xs >= FL_XF_HDR + vs && (xs % 4) == 0 && vs >= 2 && vs <= 64, "bad record strides");
    REQUIRE(xo + (nxf + hf) * xs <= ps, "xform records exceed the block");
    return FL_OK;
}

static bool known_var(int id)
{
    for (int k : kKnownVars) if (k == id) return true;
    return false;
}

int fl_genome_create(fl_ctx *c, const int32_t *prog, uint32_t nprog, const int32_t *ops, uint32_t nops,
                     uint32_t nrows, fl_genome **out)
{
    REQUIRE(c && prog && ops && out, "null argument");
    int rc = check_prog(prog, nprog);
    if (rc) return rc;
    REQUIRE(nrows >= 1 && nrows <= 4096 && nops >= 1, "bad row / op count");
    uint32_t ps = prog[3];
    const int xo = prog[5], xs = prog[6], vs = prog[7], nrec = prog[1] + prog[2];
    std::vector<int> nvar_seen(nrec, -1);
    for (uint32_t i = 0; i < nops; ++i) {
        const int32_t *o = ops + 4 * i;
        REQUIRE(o[0] >= FL_OP_SPLINE && o[0] <= FL_OP_CONST, "bad op kind");
        REQUIRE(o[1] >= 0 && (uint32_t)o[1] < ps, "op destination out of range");
        if (o[0] != FL_OP_CONST) {
            // every word an op writes and every spline row it reads must lie inside the block / row table
            uint32_t ndst = 1, nsrc = 1;
            switch (o[0]) {
            case FL_OP_CAMERA: ndst = 6; nsrc = 4; break;
            case FL_OP_AFFINE: ndst = 6; nsrc = 6; break;
            case FL_OP_CDF:
                REQUIRE(o[3] >= 1 && o[3] <= FL_MAX_XFORMS, "bad CDF length");
                ndst = nsrc = (uint32_t)o[3];
                break;
            case FL_OP_PERSP: ndst = 3; break;
            default: break;
            }
            REQUIRE((uint32_t)o[1] + ndst <= ps, "op destination out of range");
            REQUIRE(o[2] >= 0 && (uint32_t)o[2] + nsrc <= nrows, "op row out of range");
            if (o[0] == FL_OP_RATIO2 || o[0] == FL_OP_PERSP) REQUIRE(o[3] >= 0 && (uint32_t)o[3] < nrows, "op row out of range");
            continue;
        }
        // structure words: xform word 14 (nvar | post << 8) or a variation number
        const int rel = o[1] - xo;
        REQUIRE(rel >= 0 && rel / xs < nrec, "structure word outside the xform records");
        const int w = rel % xs;
        if (w == 14) {
            const int nv = o[2] & 0xff;
            REQUIRE(FL_XF_HDR + nv * vs <= xs && (o[2] >> 9) == 0, "bad variation count");
            nvar_seen[rel / xs] = nv;
        } else {
            REQUIRE(w >= FL_XF_HDR && (w - FL_XF_HDR) % vs == 0, "misplaced structure word");
            if (!known_var(o[2])) return fail(FL_E_UNSUPPORTED, "unknown variation id", __FILE__, __LINE__);
        }
    }
    for (int i = 0; i < nrec; ++i) REQUIRE(nvar_seen[i] >= 0, "xform record without a variation count");
    // structure tables for the specialised kernel: counts / post flags / variation numbers per record
    IterSpec spec;
    spec.nxf = prog[1]; spec.has_final = prog[2]; spec.pstride = prog[3]; spec.cdf_off = prog[4];
    spec.xf_off = xo; spec.xf_stride = xs; spec.var_stride = vs;
    spec.nvar.assign(nrec, 0); spec.post.assign(nrec, 0); spec.vids.assign(nrec, std::vector<int>());
    for (int i = 0; i < nrec; ++i) spec.vids[i].assign(nvar_seen[i], -1);
    for (uint32_t i = 0; i < nops; ++i) {
        const int32_t *o = ops + 4 * i;
        if (o[0] != FL_OP_CONST) continue;
        const int rel = o[1] - xo, rec = rel / xs, w = rel % xs;
        if (w == 14) { spec.nvar[rec] = o[2] & 0xff; spec.post[rec] = (o[2] >> 8) & 1; }
        else {
            const int j = (w - FL_XF_HDR) / vs;
            if (j < (int)spec.vids[rec].size()) spec.vids[rec][j] = o[2];
        }
    }
    for (int i = 0; i < nrec; ++i)
        for (int v : spec.vids[i]) REQUIRE(v >= 0, "variation record without a variation number");
    HIPCHK(hipSetDevice(c->device));
    fl_genome *g = new fl_genome;
    { static std::atomic<uint64_t> next_serial{1}; g->serial = next_serial.fetch_add(1); }
    g->spec = spec;
    g->prog.assign(prog, prog + nprog);
    g->nops = nops; g->nrows = nrows; g->pstride = ps;
    g->stage_bytes = 2 * 4 * (size_t)nrows * FL_KNOTS + 16 * 256 * (FL_KNOTS - 1) + 4 * FL_KNOTS;
    hipError_t e = hipSuccess;
    do {        // a failure anywhere frees what exists so far (fl_genome_destroy tolerates a partial genome)
        if ((e = hipMalloc(&g->d_prog, 4 * nprog))) break;
        if ((e = hipMalloc(&g->d_ops, 16 * nops))) break;
        if ((e = hipMalloc(&g->d_times, 4 * (size_t)nrows * FL_KNOTS))) break;
        if ((e = hipMalloc(&g->d_knots, 4 * (size_t)nrows * FL_KNOTS))) break;
        if ((e = hipMalloc(&g->d_ptimes, 4 * FL_KNOTS))) break;
        if ((e = hipMalloc(&g->d_pals, 16 * 256 * FL_KNOTS))) break;
        if ((e = hipMemcpy(g->d_prog, prog, 4 * nprog, hipMemcpyHostToDevice))) break;
        if ((e = hipMemcpy(g->d_ops, ops, 16 * nops, hipMemcpyHostToDevice))) break;
        if ((e = hipMemset(g->d_pals, 0, 16 * 256 * FL_KNOTS))) break;
        for (int i = 0; i < fl_genome::kStage && e == hipSuccess; ++i) {
            if ((e = hipHostMalloc((void **)&g->h_stage[i], g->stage_bytes, hipHostMallocDefault))) break;
            e = hipEventCreateWithFlags(&g->ev_stage[i], hipEventDisableTiming);
        }
    } while (0);
    if (e != hipSuccess) {
        fl_genome_destroy(g);
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? FL_E_NOMEM : FL_E_HIP, "genome allocation", __FILE__, __LINE__, e);
    }
    *out = g;
    return FL_OK;
}

void fl_genome_destroy(fl_genome *g)
{
    if (!g) return;
    hipFree(g->d_prog); hipFree(g->d_ops); hipFree(g->d_times); hipFree(g->d_knots);
    hipFree(g->d_ptimes); hipFree(g->d_pals);
    for (int i = 0; i < fl_genome::kStage; ++i) { if (g->h_stage[i]) hipHostFree(g->h_stage[i]); if (g->ev_stage[i]) hipEventDestroy(g->ev_stage[i]); }
    delete g;
}

int fl_genome_upload(fl_ctx *c, fl_genome *g, const float *times, const float *knots,
                     const float *pal_rgba, const float *pal_times, uint32_t npal)
{
    REQUIRE(c && g && times && knots && pal_rgba && pal_times, "null argument");
    REQUIRE(npal >= 1 && npal < FL_KNOTS, "bad palette count");
    HIPCHK(hipSetDevice(c->device));
    const size_t nb = 4 * (size_t)g->nrows * FL_KNOTS, pb = 16 * 256 * (size_t)npal, tb = 4 * FL_KNOTS;
    const int slot = g->stage_next;
    g->stage_next = (slot + 1) % fl_genome::kStage;
    HIPCHK(hipEventSynchronize(g->ev_stage[slot]));        // the copy that last used this slot is done
    unsigned char *h = g->h_stage[slot];
    memcpy(h, times, nb);
    memcpy(h + nb, knots, nb);
    memcpy(h + 2 * nb, pal_rgba, pb);
    memcpy(h + 2 * nb + pb, pal_times, tb);
    { int rc = wait_other(c, 0); if (rc) return rc; }     // the other lane's interp still reads these buffers
    HIPCHK(hipMemcpyAsync(g->d_times, h, nb, hipMemcpyHostToDevice, L(c).stream));
    HIPCHK(hipMemcpyAsync(g->d_knots, h + nb, nb, hipMemcpyHostToDevice, L(c).stream));
    HIPCHK(hipMemcpyAsync(g->d_pals, h + 2 * nb, pb, hipMemcpyHostToDevice, L(c).stream));
    HIPCHK(hipMemcpyAsync(g->d_ptimes, h + 2 * nb + pb, tb, hipMemcpyHostToDevice, L(c).stream));
    HIPCHK(hipEventRecord(g->ev_stage[slot], L(c).stream));
    g->npal = npal;
    return FL_OK;
}

int fl_frame_begin(fl_ctx *c, uint32_t *frame_id)
{
    REQUIRE(c && frame_id, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t id = c->frame_seq++;
    const uint32_t k = id % fl_ctx::kFrames;
    c->cur = (int)(id % (uint32_t)c->nlanes);             // consecutive frames go round the lanes
    c->frame_lane[k] = (uint32_t)c->cur;
    HIPCHK(hipEventRecord(c->ev_begin_[k], L(c).stream));
    HIPCHK(hipEventRecord(c->ev_end_[k], L(c).stream));       // moved forward by fl_output
    *frame_id = id;
    return FL_OK;
}

int fl_interp(fl_ctx *c, fl_genome *g, uint32_t w, uint32_t h, float ts, float td)
{
    REQUIRE(c && g && g->npal, "genome not uploaded");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    // One parameter block per walker slot: slot s iterates temporal sample s of nslots, evaluated
    // at ts + s*td/nslots, so that every temporal sample receives the same number of iterations
    // whatever the slot count (the reference: one block column per each of its 1024 temporal
    // samples, cuburn/render.py:303-307,343-346; cuburn/code/iter.py:165,184).
    const size_t need = (size_t)c->ntemporal() * g->pstride;     // (workgroups in sub-blocks: two or four blocks per slot)
    if (need > L(c).params_floats) {
        HIPCHK(hipStreamSynchronize(L(c).stream));
        hipFree(L(c).d_params); L(c).d_params = nullptr; L(c).params_floats = 0;
        HIPCHK(hipMalloc(&L(c).d_params, sizeof(float) * need));
        L(c).params_floats = need; L(c).params_serial = 0;
    }
    fl_mwc *rng_pal = c->d_rng + (size_t)c->nslots * c->nw * 64;
    { int rc = wait_other(c, 0); if (rc) return rc; }     // palette RNG states are shared
    launch_interp_palette(L(c).stream, rng_pal, g->d_ptimes, g->d_pals, ts, td / FL_PAL_H, L(c).d_palette);
    launch_interp_params(L(c).stream, L(c).d_params, g->d_times, g->d_knots, g->d_ops, g->nops, g->pstride,
                         c->ntemporal(), ts, td / (float)c->ntemporal(), d, L(c).params_serial != g->serial);
    L(c).params_serial = g->serial;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L(c).ev_interp_done, L(c).stream));
    L(c).interp_rec = true;
    return FL_OK;
}

static int do_clear(fl_ctx *c, const fl_dim &d, bool reset_points)
{
    size_t nbins = (size_t)d.ah * d.astride;
    // cuburn/render.py:321-328
    launch_clear_frame(L(c).stream, L(c).d_front, L(c).d_atom, L(c).d_hot, c->d_counters, c->d_points, (uint32_t)nbins,
                       reset_points ? c->nslots * (uint32_t)c->nw * 64u : 0u);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

// Maximum write-enabled rounds of one binned launch (bounds the sample log: nslots*NT*4 B per round).  The reference's batches grow
// 1024, 1536, 2304, ... rounds (cuburn/render.py:338-369); here they stop growing at 1024 — equal launches overlap best in the
// lane's two-stream pipeline (cfg3, 2731 rounds: 1024 + 1024 + 683 is 1.7 % faster than 1024 + 1536 + 171) — unless following the
// reference's schedule up to 2304 rounds saves a launch, i.e. a flush and a zeroed + added tile per workgroup (cfg5, 4096 rounds:
// 1024 + 1536 + 1536 instead of 4 x 1024, frame 37.0 -> 36.1 ms; profiles/r05_launch_cap.txt).
#define FL_BIN_MAX_ROUNDS 1024u
#define FL_BIN_MAX_ROUNDS_LONG 2304u

static int ensure_binned(fl_ctx *c, const fl_dim &d, uint32_t write_rounds, int buf, uint32_t *tiles_x, uint32_t *nbins,
                         uint32_t *nbatch_total, bool *wide)
{
    const uint32_t nt = (uint32_t)c->nw * 64;
    // 128x64 tiles while their number fits the 11 bits left in a staged record (up to 4K);
    // larger images use 256x64 tiles with separately staged tile numbers
    const uint32_t rows = (d.ah + FL_TILE_H - 1) / FL_TILE_H;
    *wide = ((d.astride + 127) / 128) * rows > FL_MAX_BINS || c->env_bin_wide;
    const uint32_t tw = *wide ? (1u << FL_TILE_W_WIDE_LOG2) : 128u;
    *tiles_x = (d.astride + tw - 1) / tw;
    *nbins = *tiles_x * rows;
    if (*nbins > FL_MAX_BINS_WIDE) return fail(FL_E_UNSUPPORTED, "image too large for the binned accumulate (> 8191 tiles of 256x64)", __FILE__, __LINE__);
    const uint32_t per_slot = (write_rounds + c->bin_rounds - 1) / c->bin_rounds;
    *nbatch_total = per_slot * c->nslots;
    // 32-bit words of the log: a region per batch — bin_rounds * nt records, one per word (256x64 tiles) or three per 64-bit word (flame_device.h)
    const size_t region = !*wide && FL_LOG_PACK3 ? 2 * (size_t)fl_pack3_words(c->bin_rounds * nt) : (size_t)c->bin_rounds * nt;
    size_t lw = (size_t)*nbatch_total * region + 8, dw = (size_t)*nbins * *nbatch_total;
    if (lw > L(c).log_words[buf]) {
        HIPCHK(hipStreamSynchronize(L(c).stream)); HIPCHK(hipStreamSynchronize(L(c).aux));
        hipFree(L(c).d_log[buf]); L(c).d_log[buf] = nullptr; L(c).log_words[buf] = 0;
        HIPCHK(hipMalloc(&L(c).d_log[buf], lw * 4));
        L(c).log_words[buf] = lw;
    }
    if (dw > L(c).dir_words[buf]) {
        HIPCHK(hipStreamSynchronize(L(c).stream)); HIPCHK(hipStreamSynchronize(L(c).aux));
        hipFree(L(c).d_dir[buf]); L(c).d_dir[buf] = nullptr; L(c).dir_words[buf] = 0;
        HIPCHK(hipMalloc(&L(c).d_dir[buf], dw * 4));
        L(c).dir_words[buf] = dw;
    }
    return FL_OK;
}

// One iterate launch and (binned mode) its tile accumulate.  `buf` selects the log / directory set;
// `drain` is the stream the accumulate runs on: the lane's own stream, or its aux stream when the
// launches of a frame are pipelined (then the accumulate waits for this iterate through ev_it[buf]).
static int do_iter_launch(fl_ctx *c, fl_genome *g, const fl_dim &d, uint32_t nrounds, uint32_t fuse, bool count, int acc = 0,
                          int buf = 0, hipStream_t drain = nullptr)
{
    if (!drain) drain = L(c).stream;
    uint32_t tiles_x = 0, nbins = 0, nbatch_total = 0;
    bool wide = false;
    if (acc == FL_ACCUM_BINNED) {
        if (nrounds <= fuse) return fail(FL_E_INVAL, "binned launch needs write-enabled rounds", __FILE__, __LINE__);
        int rc = ensure_binned(c, d, nrounds - fuse, buf, &tiles_x, &nbins, &nbatch_total, &wide);
        if (rc) return rc;
    }
    EvPair *e = ev_pair(c, c->iter_ev);
    const int kacc = acc == FL_ACCUM_BINNED && wide ? 3 : acc;
    // the kernel specialised for this genome's structure (compiled on first use, rtc.hip); the
    // interpreter kernel if hipRTC is unavailable, switched off, or the compile failed
    hipFunction_t fn = nullptr;
    if (c->use_rtc && !g->rtc_failed && kacc != 2) {
        const unsigned ep = rtc_epoch();
        if (g->rtc_epoch != ep) { memset(g->rtc_fn, 0, sizeof g->rtc_fn); g->rtc_epoch = ep; }     // the module cache was flushed
        hipFunction_t &slot = g->rtc_fn[c->sub_log2 ? 2 + c->sub_log2 : c->nw == 16 ? 2 : c->nw == 8][count ? 1 : 0][kacc];
        if (!slot) {
            std::string err;
            if (rtc_iter_kernel(c->device, g->spec, c->nw, c->nslots, count, kacc, &slot, &err, c->sub_log2)) {
                g->rtc_failed = true;
                slot = nullptr;
                fprintf(stderr, "libflame_hip: per-genome kernel not available (%s); using the interpreter kernel\n", err.c_str());
            }
        }
        fn = slot;
    }
    (fn ? c->n_spec_launch : c->n_interp_launch) += 1;
    if (fn)
        launch_iter_fn(L(c).stream, fn, c->nw, kacc, c->nslots, g->d_prog, L(c).d_params, L(c).d_palette, c->d_rng, c->d_points,
                       L(c).d_hot, L(c).d_atom, (float *)L(c).d_front, c->d_counters, d.astride, d.ah, c->round_counter, nrounds, fuse,
                       tiles_x, nbins, c->bin_rounds, nbatch_total, L(c).d_log[buf], L(c).d_dir[buf],
                       e ? e->a : nullptr, e ? e->b : nullptr, c->sub_log2);
    else
    launch_iter(L(c).stream, c->nw, count, kacc, c->nslots, g->d_prog, L(c).d_params, L(c).d_palette, c->d_rng, c->d_points,
                L(c).d_hot, L(c).d_atom, (float *)L(c).d_front, c->d_counters, d.astride, d.ah, c->round_counter, nrounds, fuse,
                tiles_x, nbins, c->bin_rounds, nbatch_total, L(c).d_log[buf], L(c).d_dir[buf],
                e ? e->a : nullptr, e ? e->b : nullptr, c->sub_log2);
    c->round_counter += nrounds;
    HIPCHK(hipGetLastError());
    if (acc == FL_ACCUM_BINNED) {
        if (drain != L(c).stream) {
            HIPCHK(hipEventRecord(L(c).ev_it[buf], L(c).stream));
            HIPCHK(hipStreamWaitEvent(drain, L(c).ev_it[buf], 0));
        }
        EvPair *e2 = ev_begin_on(c, c->accum_ev, drain);
        // workgroups per tile: enough of them to fill the chip several times over (~8192 in all),
        // no more — every workgroup zeroes and drains a whole LDS tile whatever its share of records
        // (256x64 tiles, one workgroup per CU: twice as many — a dense region then spreads over more
        // workgroups; cfg5 8K: 3 per tile 20.7 ms of accumulate per frame, 8 per tile 16.2, 12: 18.2)
        // (round 5: with the ganged tile order of images of more than 512 tiles — launch_accum_tiles — six per tile at 4K and 8K:
        // 507 / 531 / 619 us per 4K launch with 6 / 8 / 12, 2187 / 2410 / 2677 at 8K; profiles/r05_bin_parts.txt)
        uint32_t parts = c->bin_parts ? c->bin_parts : nbins > 512u ? ((wide ? 12800u : 6400u) + nbins / 2u) / nbins : (wide ? 16384u : 8192u) / nbins;
        parts = parts < 1u ? 1u : parts > (c->bin_parts ? 64u : 16u) ? (c->bin_parts ? 64u : 16u) : parts;
        launch_accum_tiles(drain, L(c).d_log[buf], L(c).d_dir[buf], L(c).d_palette, L(c).d_atom, (float *)L(c).d_front, tiles_x, nbins,
                           parts, nbatch_total, c->bin_rounds * (uint32_t)c->nw * 64, c->nslots, d.astride, d.ah, wide);
        ev_end_on(e2, drain);
        HIPCHK(hipGetLastError());
    }
    return FL_OK;
}

static int do_flush(fl_ctx *c, const fl_dim &d, bool use_hot = true, hipStream_t st = nullptr)
{
    if (!st) st = L(c).stream;
    EvPair *e = ev_begin_on(c, c->flush_ev, st);
    launch_flush(st, L(c).d_atom, L(c).d_front, L(c).d_hot, d.ah * d.astride, use_hot);
    ev_end_on(e, st);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_iterate(fl_ctx *c, fl_genome *g, uint32_t w, uint32_t h, double nsamples, uint32_t fuse,
               int accum_mode, uint64_t *nsamples_run)
{
    REQUIRE(c && g, "null argument");
    REQUIRE(accum_mode == FL_ACCUM_ATOMIC || accum_mode == FL_ACCUM_BINNED || accum_mode == 2, "bad accumulation mode");
    REQUIRE(L(c).params_floats >= (size_t)c->ntemporal() * g->pstride, "fl_interp has not run for this genome");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    flush_pending(c);
    if ((rc = wait_other(c, 1))) return rc;                 // walkers / RNG states are shared between lanes
    if ((rc = do_clear(c, d, true))) return rc;
    const uint32_t nt = (uint32_t)c->nw * 64;
    const double per_round = (double)c->nslots * nt;
    uint64_t rounds = (uint64_t)ceil(nsamples / per_round);
    if (rounds == 0) rounds = 1;
    if (nsamples_run) *nsamples_run = (uint64_t)(rounds * per_round);
    // cuburn/render.py:338-369: launch batches grow 4, 6, 9, 13, ... (x 256 rounds), each
    // followed by a flush; the first batch also carries the fuse rounds.  The reference alternates
    // two streams so that flush k overlaps iter k+1 (render.py:358-369).  Here, when a binned frame
    // needs several launches, the iterate kernels stay on the lane's stream and the tile accumulate
    // + flush of each launch go to the lane's aux stream, with two log / directory sets: launch k+1
    // iterates while launch k drains.  (Both kernels want the whole chip, so this buys little —
    // DESIGN.md §4.1 — but it costs nothing and hides the drains' launch gaps.)
    // (workgroups in sub-blocks have a half or a quarter of the walkers of their plain geometry: their rounds count double / fourfold
    // for the same samples per launch, i.e. the same log, flush schedule and number of launches)
    const uint64_t unit = 256ull << c->sub_log2;
    auto launches_with = [rounds, unit](uint64_t cap_) { uint32_t nl = 0; for (uint64_t r = rounds, b = 4; r; b += b / 2) { r -= std::min(std::min(r, b * unit), cap_); ++nl; } return nl; };
    // (16-wave workgroups: the long cap stops at the 1536 rounds that tests/test_gpu_parity.py::test_long_launch_log_beyond_4gb... pins —
    // a 2304-round launch of the 8K geometry is a 14.5 GB log whose record indices pass 2^31)
    const uint64_t cap_short = (uint64_t)FL_BIN_MAX_ROUNDS << c->sub_log2;
    uint64_t cap_long = (uint64_t)(c->nw == 16 ? 1536u : FL_BIN_MAX_ROUNDS_LONG) << c->sub_log2;
    // The long cap saves a launch but sizes the (grow-only) sample log and directory for the longest launch: it is taken only if what
    // would have to be allocated beyond the short cap's buffers fits the device's free memory with a margin — a frame that rendered
    // with 1024-round logs must not start failing on a shared or smaller device because a schedule saves it a flush.
    if (accum_mode == FL_ACCUM_BINNED && !c->launch_rounds && launches_with(cap_long) < launches_with(cap_short)) {
        const uint32_t nt_ = (uint32_t)c->nw * 64;
        const bool wide_ = ((d.astride + 127) / 128) * ((d.ah + FL_TILE_H - 1) / FL_TILE_H) > FL_MAX_BINS || c->env_bin_wide;
        const uint32_t tw_ = wide_ ? (1u << FL_TILE_W_WIDE_LOG2) : 128u;
        const size_t nbins_ = (size_t)((d.astride + tw_ - 1) / tw_) * ((d.ah + FL_TILE_H - 1) / FL_TILE_H);
        const size_t region = !wide_ && FL_LOG_PACK3 ? 2 * (size_t)fl_pack3_words(c->bin_rounds * nt_) : (size_t)c->bin_rounds * nt_;
        const size_t nb_long = (size_t)((std::min(rounds, cap_long) + c->bin_rounds - 1) / c->bin_rounds) * c->nslots;
        const size_t need = (nb_long * region + 8 + nbins_ * nb_long) * 4;      // one log + directory set, bytes
        size_t have = 0, grow = 0;
        for (int b = 0; b < 2; ++b) {                                            // (two sets: launches of a frame are pipelined)
            have = (L(c).log_words[b] + L(c).dir_words[b]) * 4;
            if (need > have) grow += need - have;
        }
        size_t mfree = 0, mtotal = 0;
        if (grow && (hipMemGetInfo(&mfree, &mtotal) != hipSuccess || grow + (size_t(1) << 30) > mfree)) cap_long = cap_short;
    }
    const uint64_t cap = accum_mode != FL_ACCUM_BINNED ? ~0ull : c->launch_rounds ? c->launch_rounds :
                         launches_with(cap_long) < launches_with(cap_short) ? cap_long : cap_short;
    const uint32_t nlaunch = launches_with(cap);
    const bool pipelined = accum_mode == FL_ACCUM_BINNED && nlaunch > 1 && !c->env_no_intra;
    hipStream_t drain = pipelined ? L(c).aux : L(c).stream;
    uint64_t batch = 4;
    uint32_t k = 0;
    while (rounds) {
        const uint64_t n = std::min(std::min(rounds, batch * unit), cap);
        const uint32_t f = k == 0 ? fuse : 0;
        const int buf = pipelined ? (int)(k & 1u) : 0;
        // the drains of launch k-2 read this log / directory set: they must be done before it is rewritten
        if (pipelined && k >= 2) HIPCHK(hipStreamWaitEvent(L(c).stream, L(c).ev_ac[buf], 0));
        if ((rc = do_iter_launch(c, g, d, (uint32_t)n + f, f, false, accum_mode, buf, drain))) return rc;
        if ((rc = do_flush(c, d, accum_mode != FL_ACCUM_BINNED, drain))) return rc;
        if (pipelined) HIPCHK(hipEventRecord(L(c).ev_ac[buf], drain));
        rounds -= n;
        batch += batch / 2;
        ++k;
    }
    // the walkers are free once the last iterate kernel has run (the drain kernels that follow
    // touch only this lane's buffers)
    HIPCHK(hipEventRecord(L(c).ev_iter_done, L(c).stream));
    L(c).iter_rec = true;
    if (pipelined) {                    // whatever comes next on this lane's stream sees the finished accumulator
        HIPCHK(hipStreamWaitEvent(L(c).stream, L(c).ev_ac[(k - 1) & 1u], 0));
        if (k >= 2) HIPCHK(hipStreamWaitEvent(L(c).stream, L(c).ev_ac[k & 1u], 0));
    }
    return FL_OK;
}

static void gauss7(float stdev, float *c)      // cuburn/filters.py:11-16
{
    float s = 0.0f;
    for (int i = 0; i < 7; ++i) { float x = (float)(i - 3); c[i] = expf(x * x / (-2.0f * stdev * stdev)); s += c[i]; }
    for (int i = 0; i < 7; ++i) c[i] /= s;
}

// The filter entry point defers two cheap per-pixel steps so that the NEXT call can take them
// along in one pass (cuburn's default chains are yuv -> bilateral -> logscale -> colorclip /
// smearclip): `yuv` directly in front of `bilateral` becomes part of the DE's preparation pass,
// and the DE's final un-normalising pass takes a following `logscale` and `colorclip` with it.
// Anything else that looks at the buffers (another filter, output, the debug taps, the next
// frame) first runs what is pending, so the observable behaviour is that of the separate kernels
// (the fused kernels run the same per-pixel device functions in the same order).
// The pending end of the DE: un-normalise (+ logscale if one was deferred, + colorclip if `clip` is
// given) into d_front — through the last direction's kernel when that is what is pending.
static void run_de_finish(fl_ctx *c, const float *clip)
{
    Lane &ln = L(c);
    // the eight per-direction kernels of de.hip, all queued here (the tail is known now): the accumulator in d_front goes through
    // d_back and ends in d_front
    DeTail t = {ln.pend_log ? 1 : 0, ln.pend_k1, ln.pend_k2, clip ? 1 : 0, clip ? clip[0] : 0.f, clip ? clip[1] : 0.f,
                clip ? clip[2] : 0.f, clip ? clip[3] : 0.f, clip ? clip[4] : 0.f};
    EvPair *e = ev_begin(c, c->de_ev);                               // the DE proper: fl_timings_detail[4], whoever flushes it
    float4 *Na = ln.d_back, *Nb = ln.d_front;
    launch_de_dir(ln.stream, ln.pend_dim, 0, Na, Nb, ln.pend_k7, ln.pend_dp[0], ln.pend_dp[1], ln.pend_dp[2], ln.pend_dp[3], ln.pend_dp[4], ln.pend_in_mode, nullptr);
    for (int pat = 1; pat < 7; ++pat) {
        launch_de_dir(ln.stream, ln.pend_dim, pat, Nb, Na, ln.pend_k7, ln.pend_dp[0], ln.pend_dp[1], ln.pend_dp[2], ln.pend_dp[3], ln.pend_dp[4]);
        std::swap(Na, Nb);
    }
    launch_de_dir(ln.stream, ln.pend_dim, 7, ln.d_front, Na, ln.pend_k7, ln.pend_dp[0], ln.pend_dp[1], ln.pend_dp[2], ln.pend_dp[3], ln.pend_dp[4], 0, &t);
    ev_end(c, e);
    ln.pend_finish = ln.pend_log = false;
}

static void flush_pending(fl_ctx *c)
{
    Lane &ln = L(c);
    if (ln.pend_yuv) {
        launch_yuv_to_rgb(ln.stream, ln.pend_dim, ln.d_back, ln.d_front);
        std::swap(ln.d_front, ln.d_back);
        ln.pend_yuv = false;
    }
    if (ln.pend_finish) run_de_finish(c, nullptr);
}

int fl_filter(fl_ctx *c, int id, uint32_t w, uint32_t h, const float *p, uint32_t np)
{
    REQUIRE(c, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    hipStream_t st = L(c).stream;
    float k7[7];
    EvPair *e = ev_begin(c, c->filt_ev);
    switch (id) {
    case FL_FILT_YUV:
        flush_pending(c);
        L(c).pend_yuv = true; L(c).pend_dim = d;             // runs with the next call
        break;
    case FL_FILT_BILATERAL: {            // cuburn/filters.py:62-95
        REQUIRE(np >= 5, "bilateral needs sstd,cstd,dstd,dpow,gspeed");
        gauss7(1.0f, k7);
        if (L(c).pend_finish) flush_pending(c);
        // One kernel per direction (de.hip), all eight deferred: the first normalises the accumulator as it stages it (after
        // `yuv`, if that is pending), the last un-normalises and takes a following logscale / colorclip with it — queued
        // together by run_de_finish once the tail is known (or when anything else looks at the buffers)
        L(c).pend_in_mode = L(c).pend_yuv ? 2 : 1;
        L(c).pend_yuv = false;
        L(c).pend_finish = true; L(c).pend_log = false; L(c).pend_dim = d;
        for (int i = 0; i < 5; ++i) L(c).pend_dp[i] = p[i];
        for (int i = 0; i < 7; ++i) L(c).pend_k7[i] = k7[i];
    } break;
    case FL_FILT_LOGSCALE:
        REQUIRE(np >= 2, "logscale needs k1,k2");
        if (L(c).pend_finish && !L(c).pend_log && !L(c).pend_yuv) { L(c).pend_log = true; L(c).pend_k1 = p[0]; L(c).pend_k2 = p[1]; break; }
        flush_pending(c);
        launch_logscale(st, d, L(c).d_front, p[0], p[1]);
        break;
    case FL_FILT_COLORCLIP:
        REQUIRE(np >= 5, "colorclip needs vib,highpow,gam,lin,lingam");
        if (L(c).pend_finish && !L(c).pend_yuv) {
            run_de_finish(c, p);
            break;
        }
        flush_pending(c);
        launch_colorclip(st, d, L(c).d_front, p[0], p[1], p[2], p[3], p[4]);
        break;
    case FL_FILT_SMEARCLIP:              // cuburn/filters.py:142-163
        flush_pending(c);
        REQUIRE(np >= 4, "smearclip needs width,gam_m_1,lin,lingam");
        gauss7(p[0], k7);
        launch_gamma_full_hi(st, d, L(c).d_side, L(c).d_front);
        launch_full_blur(st, d, L(c).d_back, L(c).d_side, 2, 0, k7);
        launch_full_blur(st, d, L(c).d_side, L(c).d_back, 3, 0, k7);
        launch_full_blur(st, d, L(c).d_back, L(c).d_side, 0, 0, k7);
        launch_full_blur(st, d, L(c).d_side, L(c).d_back, 1, 0, k7);
        launch_smearclip(st, d, L(c).d_front, L(c).d_side, p[1], p[2], p[3]);
        break;
    case FL_FILT_HALOCLIP:               // cuburn/filters.py:113-130
        flush_pending(c);
        REQUIRE(np >= 1, "haloclip needs gam_m_1");
        gauss7(1.0f, k7);
        launch_apply_gamma(st, d, L(c).d_blur, L(c).d_front, 0.1f);
        launch_den_blur_1c(st, d, (float *)L(c).d_side, L(c).d_blur, 2, 0, k7);
        launch_den_blur_1c(st, d, L(c).d_blur, (const float *)L(c).d_side, 3, 0, k7);
        launch_haloclip(st, d, L(c).d_front, L(c).d_blur, p[0]);
        break;
    case FL_FILT_PLAINCLIP:
        flush_pending(c);
        REQUIRE(np >= 4, "plainclip needs gam_m_1,lin,lingam,brightness");
        launch_plainclip(st, d, L(c).d_front, p[0], p[1], p[2], p[3]);
        break;
    case FL_FILT_LOGENCODE:
        flush_pending(c);
        REQUIRE(np >= 1, "logencode needs degamma");
        launch_logencode(st, d, L(c).d_back, L(c).d_front, p[0]);
        std::swap(L(c).d_front, L(c).d_back);
        break;
    default:
        return fail(FL_E_UNSUPPORTED, "unknown filter id", __FILE__, __LINE__);
    }
    ev_end(c, e);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

size_t fl_output_bytes(uint32_t w, uint32_t h, int fmt)
{
    const size_t n = (size_t)w * h;
    switch (fmt) {
    case FL_OUT_RGBA8: return 4 * n;
    case FL_OUT_RGBA16: return 8 * n;
    case FL_OUT_YUV444P: return 3 * n;
    case FL_OUT_YUV444P10: case FL_OUT_YUV444P12: return 6 * n;
    case FL_OUT_YUV420P10: return 3 * n;                    // (n + 2 * n/4) * 2 bytes
    default: return 0;
    }
}

int fl_output(fl_ctx *c, uint32_t w, uint32_t h, int fmt, void *host_out, uint64_t dev_out)
{
    REQUIRE(c && fl_output_bytes(w, h, fmt) != 0, "bad argument");
    REQUIRE(fmt != FL_OUT_YUV420P10 || (w % 2 == 0 && h % 2 == 0), "4:2:0 needs even width and height");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    flush_pending(c);                                       // deferred ends of the filter chain
    void *dst = dev_out ? (void *)(uintptr_t)dev_out : L(c).d_outpix;
    { int rc2 = wait_other(c, 2); if (rc2) return rc2; }   // the dither RNG states are shared between lanes
    fl_mwc *rng_out = c->d_rng + (size_t)c->nslots * c->nw * 64 + FL_PAL_H * 256;
    launch_f32_to_rgba(L(c).stream, d, L(c).d_front, rng_out, FL_NOUT, fmt, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L(c).ev_out_done, L(c).stream));
    L(c).out_rec = true;
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, dst, fl_output_bytes(w, h, fmt), hipMemcpyDeviceToHost, L(c).stream));
    if (c->frame_seq) HIPCHK(hipEventRecord(c->ev_end_[(c->frame_seq - 1) % fl_ctx::kFrames], L(c).stream));
    return FL_OK;
}

int fl_sort_u32(fl_ctx *c, uint64_t dst_dev, uint64_t src_dev, uint32_t n, uint32_t lo_bit, uint32_t nbits, int ignore_max,
                uint32_t *nvalid)
{
    REQUIRE(c && dst_dev && src_dev && dst_dev != src_dev, "null or aliased key arrays");
    REQUIRE(nbits >= 1 && nbits <= 10 && lo_bit + nbits <= 32, "a pass sorts 1..10 bits inside the 32-bit key");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) { if (nvalid) *nvalid = 0; return FL_OK; }
    size_t chunk_words = 0;
    const size_t hist_words = sort_scratch_words(n, nbits, &chunk_words), need = hist_words + chunk_words;
    // Every pass runs on lane 0's stream whatever lane the frame loop is on: consecutive passes of a
    // multi-pass sort stay ordered across frame boundaries, and the scratch has ONE user stream.
    hipStream_t sst = c->lanes[0].stream;
    if (need > c->sort_words) {
        sync_all(c);                                        // nothing may still be using the old scratch
        hipFree(c->d_sort); c->d_sort = nullptr; c->sort_words = 0;
        HIPCHK(hipMalloc(&c->d_sort, need * 4));
        c->sort_words = need;
    }
    uint32_t *chunk_tot = c->d_sort + hist_words, *total_dev = chunk_tot + (chunk_words - 1);
    launch_sort_pass(sst, (uint32_t *)(uintptr_t)dst_dev, (const uint32_t *)(uintptr_t)src_dev, n, lo_bit, nbits,
                     ignore_max, c->d_sort, chunk_tot, total_dev);
    HIPCHK(hipGetLastError());
    if (nvalid) {                                           // the reference leaves this count on the device (sort.py:449-452)
        HIPCHK(hipMemcpyAsync(nvalid, total_dev, 4, hipMemcpyDeviceToHost, sst));
        HIPCHK(hipStreamSynchronize(sst));
    }
    return FL_OK;
}

int fl_frame_ms(fl_ctx *c, uint32_t frame_id, float *ms)
{
    REQUIRE(c && ms, "null argument");
    REQUIRE(frame_id < c->frame_seq && c->frame_seq - frame_id <= fl_ctx::kFrames, "frame id no longer tracked");
    const uint32_t k = frame_id % fl_ctx::kFrames;
    HIPCHK(hipEventSynchronize(c->ev_end_[k]));
    HIPCHK(hipEventElapsedTime(ms, c->ev_begin_[k], c->ev_end_[k]));
    return FL_OK;
}

int fl_frame_query(fl_ctx *c, uint32_t frame_id)
{
    REQUIRE(c, "null ctx");
    REQUIRE(frame_id < c->frame_seq && c->frame_seq - frame_id <= fl_ctx::kFrames, "frame id no longer tracked");
    hipError_t e = hipEventQuery(c->ev_end_[frame_id % fl_ctx::kFrames]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(FL_E_HIP, "hipEventQuery", __FILE__, __LINE__, e);
}

void *fl_host_alloc(size_t nbytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, nbytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void fl_host_free(void *p) { if (p) hipHostFree(p); }

static float sum_ms(std::vector<EvPair> &v)
{
    float t = 0.0f;
    for (auto &p : v) { float ms = 0.0f; if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) t += ms; }
    return t;
}

int fl_timings_reset(fl_ctx *c)
{
    REQUIRE(c, "null ctx");
    sync_all(c);
    c->iter_ev.clear(); c->accum_ev.clear(); c->flush_ev.clear(); c->filt_ev.clear(); c->de_ev.clear(); c->pool_used = 0;
    c->n_spec_launch = c->n_interp_launch = 0;
    return FL_OK;
}

int fl_timings_detail(fl_ctx *c, float ms[6])
{
    REQUIRE(c && ms, "null argument");
    sync_all(c);
    ms[0] = sum_ms(c->iter_ev); ms[1] = sum_ms(c->accum_ev); ms[2] = sum_ms(c->flush_ev);
    ms[3] = sum_ms(c->filt_ev); ms[4] = sum_ms(c->de_ev); ms[5] = 0.0f;      // [4]: the DE's eight launches (fused ends included), recorded where they are queued; [5]: unused since round 5
    return FL_OK;
}

int fl_launch_stats(fl_ctx *c, uint32_t out[4])
{
    REQUIRE(c && out, "null argument");
    out[0] = c->n_spec_launch; out[1] = c->n_interp_launch; out[2] = c->nslots; out[3] = (uint32_t)c->nw;
    return FL_OK;
}

int fl_measure_copy(int device, size_t nbytes, int iters, float *ms)
{
    REQUIRE(ms && iters > 0 && nbytes >= 16, "bad argument");
    HIPCHK(hipSetDevice(device));
    if (launch_measure_copy(nbytes, iters, ms)) return fail(FL_E_HIP, "copy measurement (allocation of 2 x nbytes, or the launch)", __FILE__, __LINE__);
    return FL_OK;
}

int fl_timings(fl_ctx *c, float *iter_ms, float *flush_ms, float *filter_ms, uint32_t *nlaunch)
{
    REQUIRE(c, "null ctx");
    sync_all(c);
    if (iter_ms) *iter_ms = sum_ms(c->iter_ev);
    if (flush_ms) *flush_ms = sum_ms(c->accum_ev) + sum_ms(c->flush_ev);
    if (filter_ms) *filter_ms = sum_ms(c->filt_ev);
    if (nlaunch) *nlaunch = (uint32_t)c->iter_ev.size();
    return FL_OK;
}

static int buf_ptr(fl_ctx *c, fl_genome *g, int which, void **p, size_t *cap)
{
    switch (which) {
    case FL_BUF_FRONT: *p = L(c).d_front; *cap = 16 * L(c).nbins; break;
    case FL_BUF_BACK: *p = L(c).d_back; *cap = 16 * L(c).nbins; break;
    case FL_BUF_SIDE: *p = L(c).d_side; *cap = 16 * L(c).nbins; break;
    case FL_BUF_PARAMS: *p = L(c).d_params; *cap = 4 * L(c).params_floats; break;
    case FL_BUF_PALETTE: *p = L(c).d_palette; *cap = 8 * FL_PAL_H * FL_PAL_W; break;
    case FL_BUF_POINTS: *p = c->d_points; *cap = 16 * (size_t)c->nslots * c->nw * 64; break;
    case FL_BUF_SEEDS: *p = c->d_rng; *cap = sizeof(fl_mwc) * (size_t)c->nwalkers; break;
    case FL_BUF_ATOM: *p = L(c).d_atom; *cap = 8 * L(c).nbins; break;
    case FL_BUF_HOT: *p = L(c).d_hot; *cap = 4 * (L(c).nbins / 16); break;
    default: return fail(FL_E_INVAL, "unknown buffer", __FILE__, __LINE__);
    }
    if (!*p) return fail(FL_E_INVAL, "buffer not allocated yet", __FILE__, __LINE__);
    return FL_OK;
}

int fl_read_buffer(fl_ctx *c, fl_genome *g, int which, void *dst, size_t nbytes)
{
    REQUIRE(c && dst, "null argument");
    HIPCHK(hipSetDevice(c->device));
    flush_pending(c);
    void *p; size_t cap;
    int rc = buf_ptr(c, g, which, &p, &cap);
    if (rc) return rc;
    REQUIRE(nbytes <= cap, "read larger than buffer");
    sync_all(c);
    HIPCHK(hipMemcpy(dst, p, nbytes, hipMemcpyDeviceToHost));
    return FL_OK;
}

int fl_buffer_ptr(fl_ctx *c, fl_genome *g, int which, void **dev_ptr, size_t *nbytes)
{
    REQUIRE(c && dev_ptr && nbytes, "null argument");
    HIPCHK(hipSetDevice(c->device));
    flush_pending(c);
    int rc = buf_ptr(c, g, which, dev_ptr, nbytes);
    if (rc) return rc;
    sync_all(c);
    return FL_OK;
}

int fl_buffer_ptr_async(fl_ctx *c, fl_genome *g, int which, void **dev_ptr, size_t *nbytes)
{
    REQUIRE(c && dev_ptr && nbytes, "null argument");
    HIPCHK(hipSetDevice(c->device));
    flush_pending(c);
    return buf_ptr(c, g, which, dev_ptr, nbytes);
}

int fl_reserve(fl_ctx *c, uint32_t w, uint32_t h)
{
    REQUIRE(c && w && h, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    return ensure_fb(c, d);
}

int fl_stream_dependency(fl_ctx *c, void *stream, int ctx_waits)
{
    REQUIRE(c, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t other = (hipStream_t)stream;
    // a small ring of events: an event may be re-recorded once the wait that used it has been queued
    hipEvent_t &ev = c->dep_ev[c->dep_next++ % fl_ctx::kDepEvents];
    if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (ctx_waits) {
        HIPCHK(hipEventRecord(ev, other));
        HIPCHK(hipStreamWaitEvent(L(c).stream, ev, 0));
    } else {
        flush_pending(c);                                    // deferred filter steps belong to "everything queued so far"
        HIPCHK(hipEventRecord(ev, L(c).stream));
        HIPCHK(hipStreamWaitEvent(other, ev, 0));
    }
    return FL_OK;
}

int fl_write_buffer(fl_ctx *c, fl_genome *g, int which, const void *src, size_t nbytes)
{
    REQUIRE(c && src, "null argument");
    HIPCHK(hipSetDevice(c->device));
    flush_pending(c);
    void *p; size_t cap;
    int rc = buf_ptr(c, g, which, &p, &cap);
    if (rc) return rc;
    REQUIRE(nbytes <= cap, "write larger than buffer");
    sync_all(c);
    HIPCHK(hipMemcpy(p, src, nbytes, hipMemcpyHostToDevice));
    if (which == FL_BUF_PARAMS) L(c).params_serial = 0;       // whatever was written, the next fl_interp starts from zeroed blocks
    return FL_OK;
}

int fl_debug_clear(fl_ctx *c, uint32_t w, uint32_t h, int reset_points)
{
    REQUIRE(c, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    flush_pending(c);
    return do_clear(c, d, reset_points != 0);
}

int fl_debug_iter_launch(fl_ctx *c, fl_genome *g, uint32_t w, uint32_t h, uint32_t round0,
                         uint32_t nrounds, uint32_t fuse, int accum_mode)
{
    REQUIRE(c && g && (accum_mode == FL_ACCUM_ATOMIC || accum_mode == FL_ACCUM_BINNED), "bad argument");
    REQUIRE(L(c).params_floats >= (size_t)c->ntemporal() * g->pstride, "fl_interp has not run for this genome");
    HIPCHK(hipSetDevice(c->device));
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    c->round_counter = round0;
    HIPCHK(hipMemsetAsync(c->d_counters, 0, 32, L(c).stream));
    return do_iter_launch(c, g, d, nrounds, fuse, true, accum_mode);
}

int fl_debug_flush(fl_ctx *c, uint32_t w, uint32_t h)
{
    REQUIRE(c, "null ctx");
    fl_dim d; fl_calc_dim(w, h, &d);
    int rc = ensure_fb(c, d);
    if (rc) return rc;
    flush_pending(c);
    return do_flush(c, d);
}

int fl_debug_clear_hot(fl_ctx *c, uint32_t w, uint32_t h)
{
    REQUIRE(c && L(c).d_hot, "null ctx");
    fl_dim d; fl_calc_dim(w, h, &d);
    HIPCHK(hipMemsetAsync(L(c).d_hot, 0, 4 * ((size_t)d.ah * d.astride / 16), L(c).stream));
    return FL_OK;
}

// scratch device memory of a debug tap, freed on every exit path
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n); }
};

int fl_debug_shuffle(fl_ctx *c, uint32_t round, uint32_t *out256)
{
    REQUIRE(c && out256, "null argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf d; const size_t n = (size_t)c->nw * 64;
    HIPCHK(d.alloc(4 * n));
    launch_shuffle_tap(L(c).stream, c->nw, (uint32_t *)d.p, round);
    HIPCHK(hipStreamSynchronize(L(c).stream));
    HIPCHK(hipMemcpy(out256, d.p, 4 * n, hipMemcpyDeviceToHost));
    return FL_OK;
}

int fl_debug_apply_xf(fl_ctx *c, fl_genome *g, uint32_t ts, int xfi, uint32_t n, float *xyzw, fl_mwc *rng)
{
    REQUIRE(c && g && xyzw && rng && n > 0 && ts < c->ntemporal(), "bad argument");
    REQUIRE(xfi >= 0 && xfi < g->prog[1] + g->prog[2], "xform index out of range");
    REQUIRE(L(c).params_floats >= (size_t)c->ntemporal() * g->pstride, "fl_interp has not run for this genome");
    HIPCHK(hipSetDevice(c->device));
    DevBuf dp, dr;
    HIPCHK(dp.alloc(16 * (size_t)n));
    HIPCHK(dr.alloc(sizeof(fl_mwc) * (size_t)n));
    HIPCHK(hipMemcpy(dp.p, xyzw, 16 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr.p, rng, sizeof(fl_mwc) * (size_t)n, hipMemcpyHostToDevice));
    launch_apply_xf_tap(L(c).stream, g->d_prog, L(c).d_params, ts, xfi, n, (float4 *)dp.p, (fl_mwc *)dr.p);
    HIPCHK(hipStreamSynchronize(L(c).stream));
    HIPCHK(hipMemcpy(xyzw, dp.p, 16 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rng, dr.p, sizeof(fl_mwc) * (size_t)n, hipMemcpyDeviceToHost));
    return FL_OK;
}

int fl_rtc_compile_check(const int32_t *prog, uint32_t nprog, const int32_t *ops, uint32_t nops, int nw, int count, int acc,
                         char *log, size_t log_bytes)
{
    REQUIRE(prog && ops && nprog >= FL_PROG_HDR && (nw == 4 || nw == 8 || nw == 16) && acc >= 0 && acc <= 3, "bad argument");
    int rc = check_prog(prog, nprog);
    if (rc) return rc;
    const int xo = prog[5], xs = prog[6], vs = prog[7], nrec = prog[1] + prog[2];
    IterSpec spec;
    spec.nxf = prog[1]; spec.has_final = prog[2]; spec.pstride = prog[3]; spec.cdf_off = prog[4];
    spec.xf_off = xo; spec.xf_stride = xs; spec.var_stride = vs;
    spec.nvar.assign(nrec, 0); spec.post.assign(nrec, 0); spec.vids.assign(nrec, std::vector<int>(16, 0));
    for (uint32_t i = 0; i < nops; ++i) {
        const int32_t *o = ops + 4 * i;
        if (o[0] != FL_OP_CONST) continue;
        const int rel = o[1] - xo, rec = rel / xs, w = rel % xs;
        REQUIRE(rel >= 0 && rec < nrec, "structure word outside the xform records");
        if (w == 14) { spec.nvar[rec] = o[2] & 0xff; spec.post[rec] = (o[2] >> 8) & 1; }
        else { const int j = (w - FL_XF_HDR) / vs; if (j >= (int)spec.vids[rec].size()) spec.vids[rec].resize(j + 1, 0); spec.vids[rec][j] = o[2]; }
    }
    std::vector<char> code;
    std::string err;
    rc = rtc_compile(spec, nw, (count & 1) != 0, acc, &code, &err, nullptr, (count & 2) == 0 ? 0u : nw == 8 ? 1u : nw == 16 ? 2u : 0u);
    if (log && log_bytes) { snprintf(log, log_bytes, "%s", rc ? err.c_str() : "ok"); }
    if (rc) return fail(rtc_available() ? FL_E_HIP : FL_E_UNSUPPORTED, "per-genome kernel did not compile", __FILE__, __LINE__);
    return (int)(code.size() > 0 ? FL_OK : FL_E_HIP);
}

int fl_debug_counters(fl_ctx *c, uint64_t out4[4])
{
    REQUIRE(c && out4, "null argument");
    sync_all(c);
    HIPCHK(hipMemcpy(out4, c->d_counters, 32, hipMemcpyDeviceToHost));
    return FL_OK;
}

} // extern "C"
#pragma GCC visibility pop
