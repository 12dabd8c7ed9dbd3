// rtc.hip — per-genome specialisation of the iterate kernel with hipRTC.
//
// The reference generates and compiles CUDA source for every genome (cuburn/render.py:232-236,
// cuburn/code/iter.py:559-575, Tempita + nvcc through PyCUDA, up to 20 modules cached,
// render.py:229-245).  Here one precompiled kernel interprets any genome; on top of it the STRUCTURE
// of a genome (xform count, variation numbers per xform, post affines, final xform, record strides)
// can be compiled into the same kernel source at run time: iter.hip is rebuilt by hipRTC with the
// structure as constexpr tables ("flame_spec.h", generated below), which removes the variation
// dispatch, the variation loop and the post / final tests from the round loop.  All VALUES (affine
// coefficients, weights, variation parameters) stay data in the parameter block, so an animation
// whose structure does not change compiles once.  Code objects are cached per process by
// (device, structure, walker geometry, accumulate mode).  libhiprtc is loaded with dlopen: without it
// (or with FLAME_RTC=0) the interpreter kernel runs — same results, bit for bit.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "kernels.h"
#include "flame_device.h"      // the compile-time switches the run-time build must share with this library (FL_LOG_PACK3)
#define FL_STR2(x) #x
#define FL_STR(x) FL_STR2(x)
#include "rtc_sources.inc"

namespace {

struct Api {
    bool ok = false;
    hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
    hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
};

Api load_api()
{
    Api a;
    void *h = nullptr;
    for (const char *name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return a;
#define SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym)); if (!a.field) return a
    SYM(create, "hiprtcCreateProgram"); SYM(compile, "hiprtcCompileProgram");
    SYM(log_size, "hiprtcGetProgramLogSize"); SYM(log, "hiprtcGetProgramLog");
    SYM(code_size, "hiprtcGetCodeSize"); SYM(code, "hiprtcGetCode"); SYM(destroy, "hiprtcDestroyProgram");
#undef SYM
    a.ok = true;
    return a;
}

const Api &api() { static Api a = load_api(); return a; }

struct Entry { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
std::mutex g_mu;
std::map<std::string, Entry> g_cache;           // key: device + generated header
unsigned g_epoch = 1;                           // bumped whenever modules are unloaded: handles held by callers expire
const size_t kMaxModules = 64;                  // (the reference keeps 20, render.py:229)

std::string spec_header(const IterSpec &s, int nw, bool count, int acc, uint32_t sub_log2)
{
    const int nrec = s.nxf + s.has_final;
    int maxv = 1;
    for (int i = 0; i < nrec; ++i) maxv = std::max(maxv, s.nvar[i]);
    std::string h = "// generated: structure of one genome (rtc.hip)\n";
    char buf[256];
#define DEF(name, val) snprintf(buf, sizeof buf, "#define %s %d\n", name, (int)(val)); h += buf
    DEF("FL_SPEC_NXF", s.nxf); DEF("FL_SPEC_FINAL", s.has_final); DEF("FL_SPEC_PSTRIDE", s.pstride);
    DEF("FL_SPEC_CDF_OFF", s.cdf_off); DEF("FL_SPEC_XF_OFF", s.xf_off); DEF("FL_SPEC_XF_STRIDE", s.xf_stride);
    DEF("FL_SPEC_VAR_STRIDE", s.var_stride); DEF("FL_SPEC_NW", nw); DEF("FL_SPEC_COUNT", count ? 1 : 0); DEF("FL_SPEC_ACC", acc);
    DEF("FL_SPEC_SUB_LOG2", sub_log2 != 0u && (4 << sub_log2) == nw ? (int)sub_log2 : 0);
#undef DEF
    h += "constexpr int kSpecNvar[] = {";
    for (int i = 0; i < nrec; ++i) h += std::to_string(s.nvar[i]) + ",";
    h += "0};\nconstexpr int kSpecPost[] = {";
    for (int i = 0; i < nrec; ++i) h += std::to_string(s.post[i]) + ",";
    snprintf(buf, sizeof buf, "0};\nconstexpr int kSpecVid[][%d] = {", maxv);
    h += buf;
    for (int i = 0; i < nrec; ++i) {
        h += "{";
        for (int j = 0; j < maxv; ++j) h += std::to_string(j < s.nvar[i] ? s.vids[i][j] : 0) + ",";
        h += "},";
    }
    h += "{0}};\n";
    return h;
}

}  // namespace

bool rtc_available() { return api().ok; }

int rtc_compile(const IterSpec &spec, int nw, bool count, int acc, std::vector<char> *code, std::string *err, const char *extra_opt, uint32_t sub_log2)
{
    const Api &a = api();
    if (!a.ok) { *err = "libhiprtc not found"; return -1; }
    const std::string header = spec_header(spec, nw, count, acc, sub_log2);
    hiprtcProgram prog = nullptr;
    const char *hdr_src[] = {rtc_src_variations, rtc_src_device, rtc_src_abi, header.c_str()};
    const char *hdr_name[] = {"variations.h", "flame_device.h", "flame_hip.h", "flame_spec.h"};
    if (a.create(&prog, rtc_src_iter, "iter.hip", 4, hdr_src, hdr_name) != HIPRTC_SUCCESS) { *err = "hiprtcCreateProgram failed"; return -1; }
    // same code generation options as the ahead-of-time build of iter.hip (csrc/Makefile); the target is the
    // current device's own architecture string (fl_ctx_create has refused anything that is not gfx950)
    std::string arch_opt = "--offload-arch=gfx950";
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
            std::string name(prop.gcnArchName);
            arch_opt = "--offload-arch=" + name.substr(0, name.find(':'));
        } else (void)hipGetLastError();
    }
    const char *opts[] = {arch_opt.c_str(), "-O3", "-std=c++20", "-ffp-contract=off", "-DFL_RTC=1",
                          "-mllvm", "-structurizecfg-skip-uniform-regions=true",
#ifdef FL_SCAN_SERIAL_MAX
                          "-DFL_SCAN_SERIAL_MAX=" FL_STR(FL_SCAN_SERIAL_MAX),
#endif
#ifdef FL_CNT_SETS
                          "-DFL_CNT_SETS=" FL_STR(FL_CNT_SETS),
#endif
#ifdef FL_CNT_SETS_BIG
                          "-DFL_CNT_SETS_BIG=" FL_STR(FL_CNT_SETS_BIG),
#endif
                          "-DFL_LOG_PACK3=" FL_STR(FL_LOG_PACK3),
#ifdef FL_BIN_R_MAX
                          "-DFL_BIN_R_MAX=" FL_STR(FL_BIN_R_MAX),
#endif
#ifdef FL_ITER_ROT3
                          "-DFL_ITER_ROT3=" FL_STR(FL_ITER_ROT3),
#endif
                          "-fno-slp-vectorize",
    };
    // FLAME_RTC_FLAGS="-mllvm -x=y ...": extra options for code-generation experiments (tools/exp_rtc_flags.sh)
    std::vector<const char *> optv(opts, opts + sizeof opts / sizeof *opts);
    if (extra_opt) optv.push_back(extra_opt);
    std::vector<std::string> extra;
    if (const char *e = getenv("FLAME_RTC_FLAGS")) {
        std::string w;
        for (const char *q = e;; ++q) {
            if (*q && *q != ' ') { w += *q; continue; }
            if (!w.empty()) { extra.push_back(w); w.clear(); }
            if (!*q) break;
        }
        for (const std::string &x : extra) optv.push_back(x.c_str());
    }
    const hiprtcResult rc = a.compile(prog, (int)optv.size(), optv.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        a.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) a.log(prog, &log[0]);
        *err = "hiprtc compile failed: " + log.substr(0, 4000);
        a.destroy(&prog);
        return -1;
    }
    size_t n = 0;
    a.code_size(prog, &n);
    code->resize(n);
    a.code(prog, code->data());
    a.destroy(&prog);
    // FLAME_RTC_DUMP=<dir>: keep the generated header and the code object of the last compile, for
    // llvm-objdump -d (tools/dump_spec_kernel.py)
    if (const char *dir = getenv("FLAME_RTC_DUMP")) {
        const std::string d(dir);
        if (FILE *f = fopen((d + "/flame_spec.h").c_str(), "w")) { fwrite(header.data(), 1, header.size(), f); fclose(f); }
        if (FILE *f = fopen((d + "/k_iter_spec.co").c_str(), "wb")) { fwrite(code->data(), 1, code->size(), f); fclose(f); }
    }
    return 0;
}

unsigned rtc_epoch() { std::lock_guard<std::mutex> lock(g_mu); return g_epoch; }

int rtc_iter_kernel(int device, const IterSpec &spec, int nw, uint32_t nslots, bool count, int acc, hipFunction_t *fn, std::string *err, uint32_t sub_log2)
{
    // vector registers a four-wave kernel may use before a workgroup per CU is lost: every slot is resident at once, nslots * 4
    // waves over 1024 SIMDs of 512 registers each — 128 at the 1024 slots of frames of up to 2^28 samples (four waves per SIMD),
    // 80 at 1536 (six; the allocation granule is 8).  Other geometries are bounded by LDS, not registers: no limit.
    const uint32_t wps = (nslots * 4u + 1023u) / 1024u;
    const int reg_limit = nw == 4 && wps >= 1u && wps <= 8u ? (int)(512u / wps / 8u * 8u) : 0;
    const std::string key = std::to_string(device) + "|" + std::to_string(reg_limit) + "|" + spec_header(spec, nw, count, acc, sub_log2);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { *fn = it->second.fn; return 0; }
    // The kernel keeps wave-uniform operands of the round in vector registers (FL_HOIST_BUDGET, iter.hip); where that is what takes a
    // genome's kernel past the geometry's register limit, it is compiled again with a smaller budget — a sixth of the workgroups
    // waiting for a slot costs far more than the few instructions per round the registers save.  (Round 4 compared with 80 whatever
    // the geometry: kernels of 81-128 registers were recompiled up to three times for the 1024-slot geometry, for nothing.)
    Entry e;
    const char *budgets[] = {nullptr, "-DFL_HOIST_BUDGET=7", "-DFL_HOIST_BUDGET=5", "-DFL_HOIST_BUDGET=0"};
    Entry first; bool have_first = false;
    for (int b = 0; b < 4; ++b) {
        std::vector<char> code;
        if (rtc_compile(spec, nw, count, acc, &code, err, budgets[b], sub_log2)) { if (have_first) break; return -1; }
        Entry t;
        if (hipModuleLoadData(&t.mod, code.data()) != hipSuccess) { (void)hipGetLastError(); *err = "hipModuleLoadData failed"; if (have_first) break; return -1; }
        if (hipModuleGetFunction(&t.fn, t.mod, "k_iter_spec") != hipSuccess) {
            (void)hipGetLastError(); (void)hipModuleUnload(t.mod); *err = "k_iter_spec not found in the compiled module"; if (have_first) break; return -1;
        }
        int regs = 0;
        if (hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, t.fn) != hipSuccess) { (void)hipGetLastError(); regs = 0; }
        if (reg_limit == 0 || regs <= reg_limit) { if (have_first) (void)hipModuleUnload(first.mod); have_first = false; e = t; break; }
        if (!have_first) { first = t; have_first = true; } else (void)hipModuleUnload(t.mod);
    }
    if (have_first) e = first;          // over the limit with every budget: the registers are not the hoisted operands — keep the full budget
    err->clear();                       // (a later budget's failure text must not outlive a successful result)
    if (g_cache.size() >= kMaxModules) {
        // simple bound: drop THIS device's modules (the current device is `device`: its queued kernels are
        // waited for first); modules of other devices held by the process stay loaded — their kernels may
        // still be queued and this thread cannot wait for them from here
        (void)hipDeviceSynchronize();
        const std::string prefix = std::to_string(device) + "|";
        for (auto kv = g_cache.begin(); kv != g_cache.end();) {
            if (kv->first.compare(0, prefix.size(), prefix) == 0) { (void)hipModuleUnload(kv->second.mod); kv = g_cache.erase(kv); }
            else ++kv;
        }
        ++g_epoch;
    }
    g_cache[key] = e;
    *fn = e.fn;
    return 0;
}
