// Version / backend / error reporting of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "vtx_common.h"

static thread_local char g_err[512] = "";

void vtx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vtx_version(void) { return 100; }
extern "C" const char* vtx_last_error(void) { return g_err; }
extern "C" const char* vtx_backend(void) {
#ifdef HIPEMU
    return "hipemu";
#else
    return "hip:gfx950";
#endif
}

// Runtime switch used for A/B measurements of the two contraction-kernel generations.
namespace vtxg { int g_vtx_contraction_generation = 2; int g_vtx_ablate = getenv("VIRTEX_AMD_KFLAGS") ? atoi(getenv("VIRTEX_AMD_KFLAGS")) : 0;   /* 16: tile-major split-K block order (A/B) */ int g_vtx_tile_override = -1; thread_local int g_vtx_last_colgroups = 0; thread_local int g_vtx_last_generation = 0; }
extern "C" int vtx_last_contraction_generation(void) { return vtxg::g_vtx_last_generation; }
namespace vtxg { std::atomic<long> g_vtx_generation_count[4]; }
extern "C" int vtx_contraction_generation_counts(long* gen1, long* gen2, int reset) {
    if (gen1) *gen1 = vtxg::g_vtx_generation_count[1].load();
    if (gen2) *gen2 = vtxg::g_vtx_generation_count[2].load() + vtxg::g_vtx_generation_count[3].load();   // the LDS-DMA generations
    if (reset) { vtxg::g_vtx_generation_count[1] = 0; vtxg::g_vtx_generation_count[2] = 0; vtxg::g_vtx_generation_count[3] = 0; }
    return VTX_OK;
}
extern "C" int vtx_set_contraction_generation(int gen) {
    VTX_CHECK(gen == 1 || gen == 2, VTX_ERR_ARG, "contraction generation must be 1 or 2");
    vtxg::g_vtx_contraction_generation = gen;
    return VTX_OK;
}
const uint32_t* g_vtx_dropout_epoch = nullptr;
// Device word mixed into every dropout seed by the kernels themselves (Dropout::resolved): lets a replayed hipGraph of the step draw fresh masks.  nullptr = off.
extern "C" int vtx_set_dropout_epoch(const void* dev_u32) { g_vtx_dropout_epoch = (const uint32_t*)dev_u32; return VTX_OK; }
namespace vtxg { unsigned long long* g_vtx_dbg = nullptr; }
extern "C" int vtx_set_debug_buffer(void* p) { vtxg::g_vtx_dbg = (unsigned long long*)p; return VTX_OK; }   // measurement builds (-DVTX_ABLATE) only
extern "C" int vtx_set_ablation(int bits) { vtxg::g_vtx_ablate = bits; return VTX_OK; }
// Run-time switches of the specialised kernels (each starts from its VIRTEX_AMD_* environment variable): what the
// interleaved step-level A/B of tools/ab_step.py flips between rounds inside one process.
int g_vtx_sw_wgrad3x3 = getenv("VIRTEX_AMD_WGRAD3X3") ? atoi(getenv("VIRTEX_AMD_WGRAD3X3")) : 1;         // conv3x3_wgrad.hip: 0 off, 1 by image size, 2 always
int g_vtx_sw_stem_stream = getenv("VIRTEX_AMD_STEM_STREAM") ? atoi(getenv("VIRTEX_AMD_STEM_STREAM")) : 1;  // stem.hip
int g_vtx_sw_expand1x1 = getenv("VIRTEX_AMD_EXPAND1X1") ? atoi(getenv("VIRTEX_AMD_EXPAND1X1")) : 1;        // expand1x1.hip
int g_vtx_sw_conv3_bwd = getenv("VIRTEX_AMD_CONV3_BWD") ? atoi(getenv("VIRTEX_AMD_CONV3_BWD")) : 1;        // conv3_bwd.hip: fused bn3-backward + conv3 input / weight gradient of the stage-1 Bottlenecks
int g_vtx_sw_splitk_blocks = getenv("VIRTEX_AMD_SPLITK_BLOCKS") ? atoi(getenv("VIRTEX_AMD_SPLITK_BLOCKS")) : 512;   // split-K block target of the weight gradients
int g_vtx_sw_bn_fin2 = getenv("VIRTEX_AMD_BN_FIN2") ? atoi(getenv("VIRTEX_AMD_BN_FIN2")) : 0;   // BatchNorm strips > 512: compaction + finalize in one launch (bn_fin2_kernel).  Measured: with a release fence per block 24.66 vs 24.40 ms/step (profiles/r04_ab_bn_fin2.txt), with write-through stores 24.34 vs 24.31 (r04_ab_bn_fin2_write_through.txt): a dependent 5-us launch costs what it runs, the boundary itself ~2 us -> off (two launches keep the summation order the tests were calibrated on)
int g_vtx_sw_bn_fin_wide = getenv("VIRTEX_AMD_BN_FIN_WIDE") ? atoi(getenv("VIRTEX_AMD_BN_FIN_WIDE")) : 0;   // 1024-thread BatchNorm finalize / compaction blocks
// tile rule of the convolutions with BatchNorm epilogues (launch_auto): 0 = the plain picker, 1 = 8-wave 128x128 tiles for every
// statistics epilogue on large M (rounds 1-2), 2 = only for the forward statistics, 3 = only for the fused backward
namespace vtxg { int g_vtx_sw_stats_tile = getenv("VIRTEX_AMD_STATS_TILE") ? atoi(getenv("VIRTEX_AMD_STATS_TILE")) : 4; }
int g_vtx_sw_bn_red_adj = getenv("VIRTEX_AMD_BN_RED_ADJ") ? atoi(getenv("VIRTEX_AMD_BN_RED_ADJ")) : 0;   // stand-alone BatchNorm reductions in interleaved trips: measured neutral (step 23.96 vs 23.94, kernel time 25.78 vs 25.71 ms: profiles/r04_ab_bn_red_adj.txt) -> off, the summation order of rounds 1-3 stays
int g_vtx_sw_bn_adj = getenv("VIRTEX_AMD_BN_ADJ") ? atoi(getenv("VIRTEX_AMD_BN_ADJ")) : 1;      // flat BatchNorm apply kernels: adjacent vectors per trip
int g_vtx_sw_bn_grid = getenv("VIRTEX_AMD_BN_GRID") ? atoi(getenv("VIRTEX_AMD_BN_GRID")) : 8192;  // ... and their grid cap
int g_vtx_sw_pool_xcd = getenv("VIRTEX_AMD_POOL_XCD") ? atoi(getenv("VIRTEX_AMD_POOL_XCD")) : 0;   // the stem's pooling tails walk their pixels in XCD-major block order (vtx_xcd_major_block): overlapping windows of neighbouring blocks meet in ONE L2.  Measured (profiles/r06_pool_xcd_order.txt): fetch 621 -> 412 MB (forward tail, = its input exactly) and 728 -> 569 MB (backward apply) per launch, and NOT faster -- 242-276 vs 220-234 us per forward call, step 22.85 vs 22.80 ms (consistent with the re-fetched windows having been Infinity-Cache hits).  Off
namespace vtxg { int g_vtx_sw_conv3x3_shared = getenv("VIRTEX_AMD_CONV3X3_SHARED") ? atoi(getenv("VIRTEX_AMD_CONV3X3_SHARED")) : 1; }   // conv3x3_kernel.h
namespace vtxg { int g_vtx_sw_tile64x256 = getenv("VIRTEX_AMD_TILE64X256") ? atoi(getenv("VIRTEX_AMD_TILE64X256")) : 1; }   // launch_auto: the stem's weight gradient on one 64x256 tile
namespace vtxg { int g_vtx_sw_gen3 = getenv("VIRTEX_AMD_GEN3") ? atoi(getenv("VIRTEX_AMD_GEN3")) : 80; }   // generation-3 contraction kernels (gemm_v3.h): 0 forced only, n >= 2: taken when the cost model predicts n % of the generation-2 class rate (step A/B: 80 -> 24.26, 100 -> 24.46, off 24.65 ms/step)
namespace vtxg { int g_vtx_sw_gen3_pers = getenv("VIRTEX_AMD_GEN3_PERS") ? atoi(getenv("VIRTEX_AMD_GEN3_PERS")) : 0; }   // persistent 256x256 generation-3 blocks (gemm_v3.h): n blocks walk all tiles, the next tile's first K tile staged in front of the epilogue.  Measured neutral (text GEMMs 1 144.5 vs 1 147.7 us summed, step 24.40 vs 24.34 ms: profiles/r04_gen3_persistent.txt) -- gfx9 has ONE vmcnt for DMA loads and stores, so the next tile's first wait sits behind the epilogue's stores: off
namespace vtxg { int g_vtx_sw_gen3_s2 = 0; }   // generation 3 for the stride-2 input-gradient classes (loses per shape: off)
namespace vtxg { int g_vtx_sw_gen3_mc = getenv("VIRTEX_AMD_GEN3_MC") ? atoi(getenv("VIRTEX_AMD_GEN3_MC")) : 800; }   // generation 3 for the weight gradients (gemm_v3mc.h): 0 forced only, n: taken from M N / (M + N) >= n (MFMA-leaning shapes); step A/B (profiles/r04_ab_gen3_mc_threshold.txt): off 23.78, 200 23.71, 400 23.68, 800 23.64, 1500 23.78 ms
namespace vtxg { int g_vtx_sw_mc_eff128 = getenv("VIRTEX_AMD_MC_EFF128") ? atoi(getenv("VIRTEX_AMD_MC_EFF128")) : 70; }   // tile picker: 128x128 efficiency (%) for k-major operands (round 3: 84; with generation 3 taking the large text gradients 70 measures 23.66 / 23.88 against 23.73 / 24.04 ms per step on two boxes: profiles/r04_ab_mc_eff128.txt)
extern "C" int vtx_set_switch(const char* name, int value) {
    VTX_CHECK(name, VTX_ERR_ARG, "vtx_set_switch: null name");
    if (!strcmp(name, "wgrad3x3")) g_vtx_sw_wgrad3x3 = value;
    else if (!strcmp(name, "stem_stream")) g_vtx_sw_stem_stream = value;
    else if (!strcmp(name, "expand1x1")) g_vtx_sw_expand1x1 = value;
    else if (!strcmp(name, "conv3_bwd")) g_vtx_sw_conv3_bwd = value;
    else if (!strcmp(name, "bn_fin_wide")) g_vtx_sw_bn_fin_wide = value;
    else if (!strcmp(name, "bn_fin2")) g_vtx_sw_bn_fin2 = value;
    else if (!strcmp(name, "stats_tile")) vtxg::g_vtx_sw_stats_tile = value;
    else if (!strcmp(name, "epi_regs")) vtxg::g_vtx_ablate = (vtxg::g_vtx_ablate & ~256) | (value ? 256 : 0);   // 1: generation-3 plain epilogue by register transposition (v_permlane swaps) instead of LDS strips: measured slower, off
    else if (!strcmp(name, "tile_order")) vtxg::g_vtx_ablate = (vtxg::g_vtx_ablate & ~32) | (value ? 32 : 0);   // 1: plain block -> tile order (A/B)
    else if (!strcmp(name, "bn_adj")) g_vtx_sw_bn_adj = value;
    else if (!strcmp(name, "bn_red_adj")) g_vtx_sw_bn_red_adj = value;
    else if (!strcmp(name, "bn_grid")) g_vtx_sw_bn_grid = value > 0 ? value : 8192;
    else if (!strcmp(name, "pool_xcd")) g_vtx_sw_pool_xcd = value;
    else if (!strcmp(name, "gen3")) vtxg::g_vtx_sw_gen3 = value;
    else if (!strcmp(name, "gen3_mc")) vtxg::g_vtx_sw_gen3_mc = value;
    else if (!strcmp(name, "gen3_s2")) vtxg::g_vtx_sw_gen3_s2 = value;
    else if (!strcmp(name, "gen3_pers")) vtxg::g_vtx_sw_gen3_pers = value < 0 ? 0 : (value & ~7);
    else if (!strcmp(name, "conv3x3_shared")) vtxg::g_vtx_sw_conv3x3_shared = value;
    else if (!strcmp(name, "tile64x256")) vtxg::g_vtx_sw_tile64x256 = value;
    else if (!strcmp(name, "mc_eff128")) vtxg::g_vtx_sw_mc_eff128 = value > 0 ? value : 70;
    else if (!strcmp(name, "splitk_blocks")) g_vtx_sw_splitk_blocks = value > 0 ? value : 512;
    else VTX_CHECK(false, VTX_ERR_ARG, "vtx_set_switch: unknown switch '%s'", name);
    return VTX_OK;
}
// tests: force tile candidate 0..5 = 256x256, 256x128, 128x128, 128x64, 64x128, 64x64 (-1: automatic)
extern "C" int vtx_set_tile_override(int c) { vtxg::g_vtx_tile_override = c; return VTX_OK; }

// ---------------------------------------------------------------------------------------
// Per-launch timing of the contraction kernels (bench.py's roofline leg): while profiling is on, every
// launch carries a start and a stop HIP event (hipExtLaunchKernel: the dispatch's own begin / end
// timestamps); vtx_profile_stop synchronises the device and sums launches / seconds / algorithmic FLOPs /
// algorithmic bytes per kernel instantiation.
// ---------------------------------------------------------------------------------------
namespace vtxg {
int g_vtx_prof_on = 0;
int g_vtx_prof_only = -1;     // >= 0: only this class is timed (fewer events perturb the step less)
namespace {
struct ProfClass { std::string name; long launches = 0; double seconds = 0, flops = 0, bytes = 0; };
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
std::mutex g_prof_mu;
std::vector<ProfClass> g_prof_classes;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    // device-scope release: a default event ends the kernel with a system-scope L2 write-back, which is exactly the
    // perturbation a per-launch timer must not add
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventReleaseToDevice); return e;
}
}  // namespace
int vtx_prof_register(const char* pretty) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // "... [BM = 256, BN = 128, ..., AL = vtxg::PlainKC<unsigned short, 2>, ...]" -> keep the bracket part
    const char* br = strchr(pretty, '[');
    ProfClass c; c.name = br ? br : pretty;
    g_prof_classes.push_back(c);
    return (int)g_prof_classes.size() - 1;
}
int vtx_prof_register_family(const char* family, const char* kernel_expr, const char* enclosing) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const char* br = enclosing ? strchr(enclosing, '[') : nullptr;      // "[T = unsigned short, UNR = 2]" of a templated launcher
    ProfClass c; c.name = std::string(family) + "|" + (kernel_expr ? kernel_expr : "") + "|" + (br ? br : "");
    g_prof_classes.push_back(c);
    return (int)g_prof_classes.size() - 1;
}
void vtx_prof_events(int cls, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{prof_event(), prof_event(), cls, flops, bytes};
    g_prof_recs.push_back(r);
    *start = r.a; *stop = r.b;
}
}  // namespace vtxg

extern "C" int vtx_profile_select(int cls) { vtxg::g_vtx_prof_only = cls; return VTX_OK; }
extern "C" int vtx_profile_start(void) {
    using namespace vtxg;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& c : g_prof_classes) { c.launches = 0; c.seconds = c.flops = c.bytes = 0; }
    for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof_recs.clear();
    g_vtx_prof_on = 1;
    return VTX_OK;
}
extern "C" int vtx_profile_stop(void) {
    using namespace vtxg;
    g_vtx_prof_on = 0;
    VTX_CHECK(hipDeviceSynchronize() == hipSuccess, VTX_ERR_LAUNCH, "profile_stop: device synchronize failed");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ProfClass& c = g_prof_classes[r.cls];
            c.launches += 1; c.seconds += ms * 1e-3; c.flops += r.flops; c.bytes += r.bytes;
        }
        g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return (int)g_prof_classes.size();
}
extern "C" int vtx_profile_get(int cls, char* name, int name_len, long* launches, double* seconds, double* flops,
                               double* bytes) {
    using namespace vtxg;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    VTX_CHECK(cls >= 0 && cls < (int)g_prof_classes.size(), VTX_ERR_ARG, "profile_get: no class %d", cls);
    const ProfClass& c = g_prof_classes[cls];
    if (name && name_len > 0) { strncpy(name, c.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (launches) *launches = c.launches;
    if (seconds) *seconds = c.seconds;
    if (flops) *flops = c.flops;
    if (bytes) *bytes = c.bytes;
    return VTX_OK;
}
