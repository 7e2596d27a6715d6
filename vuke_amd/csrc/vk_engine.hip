// vk_engine.hip -- host side of the C ABI in include/vuke_hip.h: device memory, launches,
// error reporting.  No CPU compute path exists here: without a gfx950 device every entry
// point returns VK_E_NO_DEVICE (the oracle under oracle/ is test infrastructure and is never
// linked into this library).
#include "../../include/vuke_hip.h"
#include "vk_aux_kernels.cuh"
#include "vk_sweep_launch.h"
#include "vk_electrum.cuh"
#include "vk_multibit.cuh"
#include "vk_chain_keys.cuh"
#include "vk_search.cuh"
#include "vk_lines.cuh"
#include "vk_records.cuh"
#include "vk_scan.cuh"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <cctype>
#include <chrono>
#include <new>
#include <string>
#include <vector>

static_assert(sizeof(vk_hit) == sizeof(vk_hit_dev), "vk_hit layout");
static_assert(sizeof(vk_hit) == 72, "vk_hit is 72 bytes");

static thread_local std::string g_create_error;

enum { VK_SCR_STAGED = 10, VK_SCR_SLOTS = 11 };

struct vk_engine {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cu_count = 0;
    u32 W = 24, nwin = 11; // default: 11 windows x 2^24 entries x 64 B = 11.8 GB of the 288 GB HBM
    u32 layout = 0;        // 0 uniform unsigned W-bit windows, 1 nine signed windows (VK_LAYOUT_SIGNED9)
    vk_win_desc wins[VK_MAX_WIN];
    u64 table_entries = 0;
    u64 *d_wordlist = nullptr;     // BIP-39 list, 2048 x 8 bytes (vk_bip39_wordlist_set), then the same words as an open-addressed set
    u64 *d_mb_salt = nullptr;      // PBKDF2 salt message of the multibit chain, ready-padded blocks (vk_multibit_passphrase_set; "" by default)
    u32 mb_salt_nblk = 0;
    size_t mb_salt_cap = 0; // words of d_mb_salt
    int ring_waves_per_simd = 0;   // resident worker waves per SIMD of the chain ring (occupancy query, once)
    int chain_stop = 99;           // vk_debug_chain_stop (test-hook build only; 99 = run every stage)
    u32 dbg_ring_chunk = 0, dbg_ring_wps = 0, dbg_ring_workers = 0; // vk_debug_ring_knobs (test-hook build only; 0 = library's choice)
    double clk_cycles = 0.0, clk_ticks = 0.0; // shader-clock samples of the sweep launches of the call in progress
    u64 rescan_items = 0;          // vk_rescan_strings: keys of the last vk_scan_strings call still in scratch slots 1 / 4 (0 = gone)
    u32 rescan_slots = 0;
    u64 rescan_inputs = 0, rescan_skipped = 0;
    vk_win_desc *d_wins = nullptr; // device copy of wins (kernels that take the layout by pointer)
    double table_build_seconds = 0.0;
    double runtime_init_seconds = 0.0, alloc_seconds = 0.0, table_wall_seconds = 0.0; // where vk_engine_create's wall time went
    u32 batch = 128;
    u32 blocks_per_cu = 0;   // workgroups per CU the grids and the workspace are sized for
    bool bpc_auto = true;    // not given by the caller: every launch runs the form the automatic choice names (vk_sweep_design_form)
    uint4 *table = nullptr;
    uint4 *tgt = nullptr;
    u32 *filt = nullptr;     // pre-filter bitmap of the target table (vk_kernels.cuh: probe_filter_word)
    u32 filt_mask = 0;       // its bits - 1
    u32 last_form = 0, last_grid = 0, last_R = 0; // the last sweep launch (vk_debug_last_launch)
    u32 tgt_cap = 0;
    size_t tgt_count = 0;
    vk_hit_dev *d_hits = nullptr;
    size_t d_hits_cap = 0;
    u64 *d_stats = nullptr;
    uint4 *work = nullptr;
    size_t work_bytes = 0;
    // device scratch for the batch entry points (inputs / outputs of one call), kept between calls: a hipMalloc +
    // hipFree pair per buffer per call cost more than the kernel for batches of a few hundred keys
    // slots 0-5: inputs / outputs; 6-9: the key-derivation chains (staging slices, chain state, validity, ring);
    // VK_SCR_STAGED: the staged keys of run_range_staged -- owned by run_range, so no caller of run_range may hold it
    // (every other slot a caller took BEFORE run_range stays valid across it: run_range touches no other slot)
    void *scr[VK_SCR_SLOTS] = {};
    size_t scr_cap[VK_SCR_SLOTS] = {};
    std::mutex mu;
    std::string err;
    std::string warn; // what a successful call wants the caller to know (vk_last_warning); cleared on entry of every call that can set it
};

// The lock every entry point takes.  It also reads away whatever error an EARLIER runtime call of this process left behind (a failed
// hipMalloc of another engine, another library in the process): the checks behind the kernel launches read hipGetLastError(), and the
// runtime keeps an error until somebody reads it -- found by four ranks sharing one card in round 5: three 120 GB tables did not fit,
// and the fall-back engine of the same process then reported the first attempt's "out of memory" at its first launch.
// The engine is serialised by its mutex.  Entry points that launch kernels check their launches with hipGetLastError(), which reports
// the calling THREAD's last non-sticky HIP error whoever caused it: they read a pending one away on entry (a foreign error -- another
// library sharing the thread -- is discarded rather than blamed on this launch; documented in include/vuke_hip.h).  The engine's own
// failures are read away where they happen (HIPCHK, scratch(), the workspace allocation of run_range), and entry points that launch
// nothing (vk_last_warning, the debug setters; vk_engine_info takes no lock at all) leave the thread's error state alone.
struct vk_api_lock {
    std::lock_guard<std::mutex> lk;
    explicit vk_api_lock(std::mutex &m, bool launches = true) : lk(m)
    {
        if (launches) (void)hipGetLastError();
    }
};

// slot `i` of the engine's scratch, at least `bytes` long (contents undefined); nullptr + e->err on failure
static void *scratch(vk_engine *e, int i, size_t bytes)
{
    if (i == 1 || i == 4) e->rescan_items = 0; // the keys / validity flags of the last vk_scan_strings call are about to be overwritten
    if (bytes == 0) bytes = 1;
    if (e->scr_cap[i] >= bytes) return e->scr[i];
    if (e->scr[i]) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipFree(e->scr[i]);
        e->scr[i] = nullptr;
        e->scr_cap[i] = 0;
    }
    const size_t cap = bytes + bytes / 4; // grow with slack so a slowly growing caller does not reallocate every time
    hipError_t c = hipMalloc(&e->scr[i], cap);
    if (c != hipSuccess) {
        e->scr[i] = nullptr;
        e->err = std::string("hipMalloc of ") + std::to_string(cap) + " scratch bytes failed: " + hipGetErrorString(c);
        (void)hipGetLastError(); // read it away: see HIPCHK
        return nullptr;
    }
    e->scr_cap[i] = cap;
    return e->scr[i];
}
#define SCRATCH(e, var, type, slot, bytes)                 \
    type var = (type)scratch((e), (slot), (bytes));        \
    if (!var) return VK_E_HIP

#define HIPCHK(e, call)                                                                             \
    do {                                                                                            \
        hipError_t _r = (call);                                                                     \
        if (_r != hipSuccess) {                                                                     \
            char _b[512];                                                                           \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
            (e)->err = _b;                                                                          \
            (void)hipGetLastError(); /* the runtime keeps the error until it is read: a later launch check must not find it */ \
            return _r == hipErrorOutOfMemory ? VK_E_NOMEM : VK_E_HIP; /* the header's code for "device memory did not suffice" */ \
        }                                                                                           \
    } while (0)

// exclusive prefix sum of n u32 on the engine's stream (vk_scan.cuh); temp = scan_temp_bytes(n) bytes of device memory
static size_t scan_temp_bytes(u64 n) { return (size_t)(1 + (n + VK_SCAN_TILE - 1) / VK_SCAN_TILE) * sizeof(unsigned long long); }
static hipError_t exclusive_scan(vk_engine *e, void *temp, const u32 *in, u32 *out, u64 n)
{
    if (n == 0) return hipSuccess;
    hipError_t c = hipMemsetAsync(temp, 0, scan_temp_bytes(n), e->stream);
    if (c != hipSuccess) return c;
    hipLaunchKernelGGL(vk_scan_kernel, dim3((u32)((n + VK_SCAN_TILE - 1) / VK_SCAN_TILE)), dim3(256), 0, e->stream, in, out, (u32)n, (unsigned long long *)temp);
    return hipGetLastError();
}


static int fail(vk_engine *e, int code, const char *msg)
{
    if (e) e->err = msg;
    else g_create_error = msg;
    return code;
}

// ---- nothing throws across the C ABI (include/vuke_hip.h; the reference's convention: errors are values, src/gpu/error.rs:7-20) ----
// Every extern "C" function below that can allocate on the host (std::vector staging buffers, std::string messages) or take the
// engine's mutex is a function-try-block ending in VK_API_CATCH: std::bad_alloc / std::length_error become VK_E_NOMEM, anything else
// VK_E_HIP, with vk_last_error set from storage that was reserved when the engine was created (the handler itself must not allocate:
// an exception leaving an extern "C" frame into a Rust caller is an abort).
static int vk_caught(vk_engine *e) noexcept
{
    int code = VK_E_HIP;
    char msg[200];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = VK_E_NOMEM;
        snprintf(msg, sizeof msg, "out of host memory (std::bad_alloc inside the library)");
    } catch (const std::length_error &x) {
        code = VK_E_NOMEM;
        snprintf(msg, sizeof msg, "host buffer too large (std::length_error: %.120s)", x.what());
    } catch (const std::exception &x) {
        snprintf(msg, sizeof msg, "unexpected C++ exception inside the library: %.140s", x.what());
    } catch (...) {
        snprintf(msg, sizeof msg, "unexpected C++ exception inside the library");
    }
    try {
        if (e) e->err.assign(msg); // capacity reserved in engine_create: no allocation
        else g_create_error.assign(msg);
    } catch (...) {
    }
    (void)hipGetLastError();
    return code;
}
#define VK_API_CATCH(e)     \
    catch (...)             \
    {                       \
        return vk_caught(e); \
    }

extern "C" int vk_abi_version(void) { return VK_ABI_VERSION; }

extern "C" int vk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" uint32_t vk_slots_per_input(const vk_transform_spec *t)
{
    if (!t) return 0;
    switch (t->kind) {
    case VK_T_SHA256:
    case VK_T_DOUBLE_SHA256:
    case VK_T_DIRECT: return 3;
    case VK_T_MILKSAD:
    case VK_T_MD5:
    case VK_T_MT64: return 1;
    case VK_T_ARMORY: return 2;
    case VK_T_LCG:
    case VK_T_XORSHIFT:
        if (t->variant != VK_VARIANT_ALL && t->variant > 3) return 0;
        return t->variant == VK_VARIANT_ALL ? 4u : 1u;
    case VK_T_SHA256_CHAIN:
        if (t->variant != VK_VARIANT_ALL && t->variant > 3) return 0;
        if (t->chain_depth == 0 || t->chain_depth > 4096) return 0;
        return 2u * (t->variant == VK_VARIANT_ALL ? 4u : 1u) * t->chain_depth;
    case VK_T_ELECTRUM:
        if (t->variant > 1 || t->chain_depth == 0 || t->chain_depth > 1000) return 0;
        return t->chain_depth;
    case VK_T_MULTIBIT:
        if (t->chain_depth == 0 || t->chain_depth > 1000) return 0;
        return t->chain_depth;
    default: return 0;
    }
}

extern "C" uint32_t vk_slots_per_string(const vk_transform_spec *t)
{
    if (!t) return 0;
    switch (t->kind) {
    case VK_T_SHA256:
    case VK_T_DOUBLE_SHA256:
    case VK_T_MD5:
    case VK_T_DIRECT:
    case VK_T_ARMORY: return 1;
    case VK_T_SHA256_CHAIN: return vk_slots_per_input(t) / 2; // the string form only (src/transform/sha256_chain.rs:60-72)
    case VK_T_ELECTRUM:
    case VK_T_MULTIBIT: return vk_slots_per_input(t);
    default: return 0; // the PRNG transforms need Input::u64_val: nothing is emitted for a string
    }
}

static int validate_spec(vk_engine *e, const vk_transform_spec *t, bool strings = false)
{
    if (!t) return fail(e, VK_E_INVALID_ARG, "transform spec is NULL");
    if (vk_slots_per_input(t) == 0) return fail(e, VK_E_INVALID_ARG, "invalid transform spec (kind / variant / chain_depth)");
    const bool string_kind = t->kind == VK_T_ELECTRUM || t->kind == VK_T_MULTIBIT;
    if (!strings && string_kind)
        return fail(e, VK_E_UNSUPPORTED, "electrum / multibit take string inputs: use vk_transform_strings + vk_scan_keys");
    if (strings && vk_slots_per_string(t) == 0) return fail(e, VK_E_UNSUPPORTED, "this transform emits nothing for string inputs (it needs a numeric seed)");
    if (strings && t->kind == VK_T_MULTIBIT && !e->d_wordlist)
        return fail(e, VK_E_INVALID_ARG, "multibit needs the BIP-39 word list: call vk_bip39_wordlist_set first");
    if (t->kind == VK_T_LCG && t->endian > 1) return fail(e, VK_E_INVALID_ARG, "lcg endian must be 0 (big) or 1 (little)");
    return VK_OK;
}

extern "C" const char *vk_last_error(const vk_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

extern "C" void vk_engine_destroy(vk_engine *e)
{
    if (!e) return;
    if (e->device >= 0) (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void *p : e->scr)
        if (p) (void)hipFree(p);
    if (e->table) (void)hipFree(e->table);
    if (e->tgt) (void)hipFree(e->tgt);
    if (e->filt) (void)hipFree(e->filt);
    if (e->d_hits) (void)hipFree(e->d_hits);
    if (e->d_stats) (void)hipFree(e->d_stats);
    if (e->work) (void)hipFree(e->work);
    if (e->d_wins) (void)hipFree(e->d_wins);
    if (e->d_wordlist) (void)hipFree(e->d_wordlist);
    if (e->d_mb_salt) (void)hipFree(e->d_mb_salt);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static double wall_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int engine_init(vk_engine *e, int device, const vk_engine_opts *opts)
{
    const double t_start = wall_now();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(e, VK_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(e, VK_E_INVALID_ARG, "device index out of range");
    HIPCHK(e, hipSetDevice(device));
    (void)hipGetLastError(); // start from a clean "last error" (vk_api_lock)
    e->device = device;
    hipDeviceProp_t prop;
    HIPCHK(e, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        e->err = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only";
        return VK_E_NO_DEVICE;
    }
    e->cu_count = prop.multiProcessorCount;
    if (opts && opts->layout > VK_LAYOUT_SIGNED9) return fail(e, VK_E_INVALID_ARG, "unknown table layout");
    e->layout = opts ? opts->layout : 0;
    if (opts && opts->window_bits) {
        const u32 lim = e->layout == VK_LAYOUT_SIGNED9 ? 29u : 28u;
        if (opts->window_bits < (e->layout == VK_LAYOUT_SIGNED9 ? 8u : 4u) || opts->window_bits > lim)
            return fail(e, VK_E_INVALID_ARG, e->layout == VK_LAYOUT_SIGNED9 ? "window_bits must be in 8..29 for the signed layout" : "window_bits must be in 4..28");
        e->W = opts->window_bits;
    } else if (e->layout == VK_LAYOUT_SIGNED9) {
        e->W = 29;
    }
    memset(e->wins, 0, sizeof e->wins);
    if (e->layout == VK_LAYOUT_SIGNED9) {
        // Signed-digit windows: ceil(256 / W) windows of floor / ceil(256 / windows) bits, all but the last signed (2^(bits-1) + 1
        // entries); the last one unsigned with the inclusive top entry (it absorbs the final carry).  W = 29 (the default of this
        // layout): 4 x 29 + 5 x 28 bits = nine windows, (4*2^28 + 4*2^27 + 2^28 + 9) * 64 B = 120 GB, 8 additions per key.  Since
        // round 4 any W in 8..29: W = 26 -> ten windows (6 x 26 + 4 x 25 bits), 18.3 GB, 9 additions -- what the unsigned W = 26
        // table does with 43 GB; W = 24 -> eleven windows, 4.0 GB, 10 additions like the default 11.8 GB table.
        e->nwin = (256 + e->W - 1) / e->W;
        const u32 lo_bits = 256 / e->nwin, extra = 256 % e->nwin;
        u32 off = 0;
        u64 base = 0;
        for (u32 i = 0; i < e->nwin; i++) {
            vk_win_desc &w = e->wins[i];
            w.bits = lo_bits + (i < extra ? 1u : 0u);
            w.sgn = i + 1 < e->nwin ? 1 : 0;
            w.tb = w.bits - w.sgn;
            w.off = off;
            w.base = base;
            off += w.bits;
            base += (1ull << w.tb) + 1;
        }
        e->table_entries = base;
        // several requests name the same geometry (27 and 28 give the ten windows of 26, 25 the eleven of 24): the engine
        // reports the width its widest window really has, so window_bits x windows always describes the table that was built
        e->W = lo_bits + (extra ? 1u : 0u);
    } else {
        e->nwin = (256 + e->W - 1) / e->W;
        for (u32 i = 0; i < e->nwin; i++) {
            vk_win_desc &w = e->wins[i];
            w.bits = w.tb = e->W;
            w.sgn = 0;
            w.off = i * e->W;
            w.base = (u64)i << e->W;
        }
        e->table_entries = (u64)e->nwin << e->W;
    }
    if (opts && opts->algo)
        return fail(e, VK_E_UNSUPPORTED, "vk_engine_opts.algo: only VK_ALGO_CHAIN (0) is built; the tree summation of rounds 1-2 was removed in round 3 "
                                         "(0.84-0.90 x the chain kernel in every form measured, DESIGN.md section 8)");
    if (opts && opts->batch) {
        if (opts->batch > 1024) return fail(e, VK_E_INVALID_ARG, "batch must be <= 1024");
        e->batch = opts->batch;
    }
    if (opts && opts->blocks_per_cu) {
        if (opts->blocks_per_cu > 8) return fail(e, VK_E_INVALID_ARG, "blocks_per_cu must be <= 8");
        e->blocks_per_cu = opts->blocks_per_cu;
        e->bpc_auto = false;
    }
    if (opts && opts->stream) {
        e->stream = (hipStream_t)opts->stream;
    } else {
        HIPCHK(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    HIPCHK(e, hipEventCreate(&e->ev0));
    HIPCHK(e, hipEventCreate(&e->ev1));
    HIPCHK(e, hipMalloc(&e->d_stats, 12 * sizeof(u64)));
    if (e->blocks_per_cu == 0) e->blocks_per_cu = 3; // the widest grid an instantiation is built for; launches that want two use two
    HIPCHK(e, hipMalloc(&e->d_wins, sizeof e->wins));
    HIPCHK(e, hipMemcpy(e->d_wins, e->wins, sizeof e->wins, hipMemcpyHostToDevice));
    const double t_runtime = wall_now(); // the HIP runtime is up (first engine of a process: ~0.25 s), the small buffers exist
    e->runtime_init_seconds = t_runtime - t_start;
    // batch-inversion workspace for the full grid, once (a multi-GB hipMalloc inside a sweep call costs seconds);
    // run_range only grows it for specs whose slots per input exceed the batch
    e->work_bytes = (size_t)e->cu_count * e->blocks_per_cu * VK_BLOCK * e->batch * 8 * sizeof(uint4);
    HIPCHK(e, hipMalloc(&e->work, e->work_bytes));
    // the k*G table: affine points (64 B each), built on the device window by window in two phases
    HIPCHK(e, hipMalloc(&e->table, e->table_entries * 64));
    const double t_alloc = wall_now(); // the two large allocations: waits for the driver's scrub of VRAM another process has just freed
    e->alloc_seconds = t_alloc - t_runtime;
    // one seed launch and one combine launch for all windows (vk_aux_kernels.cuh)
    vk_tbl_jobs jobs;
    memset(&jobs, 0, sizeof jobs);
    jobs.nwin = e->nwin;
    jobs.with_max = e->layout == VK_LAYOUT_SIGNED9 ? 1u : 0u;
    u32 min_nhi = 0;
    u64 max_seeds = 0;
    for (u32 i = 0; i < e->nwin; i++) {
        const vk_win_desc &w = e->wins[i];
        vk_tbl_win &j = jobs.w[i];
        j.base = w.base;
        j.off = w.off;
        j.tb = w.tb;
        j.h = w.tb < 12 ? w.tb : w.tb / 2; // a combined window gives every wave 64 low digits: h >= 6; smaller windows are seeded whole
        max_seeds = std::max<u64>(max_seeds, (1ull << j.h) + (1ull << (j.tb - j.h)) + jobs.with_max);
        if (j.h < j.tb) min_nhi = min_nhi ? std::min(min_nhi, 1u << (j.tb - j.h)) : 1u << (j.tb - j.h);
    }
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(vk_table_seed_kernel, dim3((u32)((max_seeds + VK_BLOCK - 1) / VK_BLOCK), e->nwin), dim3(VK_BLOCK), 0, e->stream, e->table, jobs);
    HIPCHK(e, hipGetLastError());
    if (min_nhi) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vk_table_combine_kernel, VK_BLOCK, 0) != hipSuccess || occ < 1) occ = 1;
        const u32 blocks = (u32)e->cu_count * (u32)std::min(occ, 4);
        const u64 row_bytes = (u64)blocks * (VK_BLOCK / 64) * 128 * sizeof(uint4); // one step of every wave of the grid
        u32 K = 256; // run length: the inversion costs ~270 / K multiplications per entry
        while (K > min_nhi) K >>= 1;
        uint4 *ws = e->work;
        void *own_ws = nullptr;
        if (row_bytes * K > e->work_bytes) { // an engine created with a small batch: its workspace is too small for a useful run
            K = std::min(K, 64u);
            HIPCHK(e, hipMalloc(&own_ws, row_bytes * K));
            ws = (uint4 *)own_ws;
        }
        jobs.K = K;
        u64 items = 0;
        for (u32 i = 0; i < e->nwin; i++) {
            vk_tbl_win &j = jobs.w[i];
            j.first_item = (u32)items;
            if (j.h < j.tb) items += (u64)((1u << j.h) >> 6) * ((1u << (j.tb - j.h)) / K);
        }
        jobs.items = (u32)items;
        hipLaunchKernelGGL(vk_table_combine_kernel, dim3(blocks), dim3(VK_BLOCK), 0, e->stream, e->table, jobs, ws);
        hipError_t lc = hipGetLastError();
        if (own_ws) {
            (void)hipStreamSynchronize(e->stream);
            (void)hipFree(own_ws);
        }
        HIPCHK(e, lc);
    }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->table_build_seconds = (double)ms * 1e-3;
        e->table_wall_seconds = wall_now() - t_alloc;
    }
    return VK_OK;
}

extern "C" int vk_engine_create_ex(int device, const vk_engine_opts *opts, vk_engine **out) try
{
    if (!out) return fail(nullptr, VK_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    vk_engine *e = new (std::nothrow) vk_engine();
    if (!e) return fail(nullptr, VK_E_NOMEM, "out of host memory");
    int rc;
    try {
        e->err.reserve(512); // vk_caught writes its message into this without allocating
        rc = engine_init(e, device, opts);
    } catch (...) {
        rc = vk_caught(e);
    }
    if (rc != VK_OK) {
        g_create_error = e->err;
        vk_engine_destroy(e);
        return rc;
    }
    *out = e;
    return VK_OK;
}
VK_API_CATCH(nullptr)

extern "C" int vk_engine_create(int device, vk_engine **out) { return vk_engine_create_ex(device, nullptr, out); }

extern "C" int vk_engine_info(const vk_engine *e, vk_engine_info_t *out)
{
    if (!e || !out) return VK_E_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->device = e->device;
    out->cu_count = (uint32_t)e->cu_count;
    out->window_bits = e->W;
    out->windows = e->nwin;
    out->batch = e->batch;
    out->blocks_per_cu = e->blocks_per_cu;
    out->table_bytes = e->table_entries * 64;
    out->layout = e->layout;
    out->table_build_seconds = e->table_build_seconds;
    out->runtime_init_seconds = e->runtime_init_seconds;
    out->alloc_seconds = e->alloc_seconds;
    out->table_wall_seconds = e->table_wall_seconds;
    return VK_OK;
}

// ------------------------------------------------------------------------- targets
extern "C" int vk_targets_set(vk_engine *e, const uint8_t *h160, const uint8_t *kind, size_t n) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    HIPCHK(e, hipSetDevice(e->device));
    if (e->tgt) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipFree(e->tgt));
        e->tgt = nullptr;
        e->tgt_cap = 0;
        e->tgt_count = 0;
        if (e->filt) HIPCHK(e, hipFree(e->filt));
        e->filt = nullptr;
        e->filt_mask = 0;
    }
    if (n == 0) return VK_OK;
    if (!h160 || !kind) return fail(e, VK_E_INVALID_ARG, "h160 / kind is NULL");
    if (n > (1ull << 30)) return fail(e, VK_E_INVALID_ARG, "too many targets (max 2^30)");
    size_t cap = 1024;
    while (cap < 2 * n) cap <<= 1; // load factor <= 1/2
    std::vector<u32> tab(cap * 8, 0u);
    const u32 mask = (u32)(cap - 1);
    // the pre-filter: one bit per target among >= 128 bits per target (at most 2^32 bits = 512 MB: the index is one hash160 word)
    u64 fbits = 1ull << 16;
    while (fbits < ((u64)n << 7) && fbits < (1ull << 32)) fbits <<= 1;
    const u32 fmask = (u32)(fbits - 1);
    std::vector<u32> filt((size_t)(fbits >> 5), 0u);
    for (size_t i = 0; i < n; i++) {
        if (kind[i] > 1) return fail(e, VK_E_INVALID_ARG, "target kind must be 0 (p2pkh) or 1 (p2wpkh)");
        u32 h[5];
        memcpy(h, h160 + 20 * i, 20); // little-endian host == RIPEMD-160 state words
        u32 bit = 1u << kind[i];
        filt[(h[1] & fmask) >> 5] |= 1u << (h[1] & 31u);
        u32 s = h[0] & mask;
        for (;;) {
            u32 *slot = &tab[(size_t)s * 8];
            if (slot[5] == 0) {
                memcpy(slot, h, 20);
                slot[5] = bit;
                break;
            }
            if (memcmp(slot, h, 20) == 0) { // same hash160 listed under both kinds
                slot[5] |= bit;
                break;
            }
            s = (s + 1) & mask;
        }
    }
    HIPCHK(e, hipMalloc(&e->tgt, cap * 32));
    HIPCHK(e, hipMemcpy(e->tgt, tab.data(), cap * 32, hipMemcpyHostToDevice));
    HIPCHK(e, hipMalloc(&e->filt, filt.size() * sizeof(u32)));
    HIPCHK(e, hipMemcpy(e->filt, filt.data(), filt.size() * sizeof(u32), hipMemcpyHostToDevice));
    e->filt_mask = fmask;
    e->tgt_cap = (u32)cap;
    e->tgt_count = n;
    return VK_OK;
}
VK_API_CATCH(e)

// ------------------------------------------------------------------------- launch helper
struct launch_io {
    vk_hit *hits = nullptr;
    size_t hit_cap = 0;
    const u32 *in_keys = nullptr; // device
    const unsigned char *in_valid = nullptr; // device, optional
    u32 *out_priv = nullptr;      // device
    u32 *out_h160c = nullptr;
    u32 *out_h160u = nullptr;
    unsigned char *out_valid = nullptr;
    u32 *out_pub = nullptr;
};


static int run_range(vk_engine *e, const vk_spec_dev &spec, u32 kind, u32 slots, u64 lo, u64 span, u32 flags, const launch_io &io, vk_sweep_stats *st);

// armory's four chained steps over `items` keys that vk_armory_begin_kernel has prepared (keys, chain codes, validity flags):
// public keys of the current keys from the sweep kernel (raw keys in, affine X || Y out), then the tweak
static int armory_steps(vk_engine *e, u64 items, u32 *d_keys, unsigned char *d_valid, u32 *d_cc, u32 *d_pub, double *seconds, u32 *launches)
{
    const vk_spec_dev rawspec = {VK_T_RAWKEYS, 0, 0, 0};
    for (int step = 0; step < 4; step++) {
        launch_io pio;
        pio.in_keys = d_keys;
        pio.in_valid = d_valid;
        pio.out_pub = d_pub;
        vk_sweep_stats tmp;
        memset(&tmp, 0, sizeof tmp);
        const int prc = run_range(e, rawspec, VK_T_RAWKEYS, 1, 0, items - 1, VK_F_COMPRESSED | VK_F_EMIT, pio, &tmp);
        if (prc) return prc;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        hipLaunchKernelGGL(vk_armory_step_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), dim3(VK_BLOCK), 0, e->stream, items, d_pub, d_keys, d_cc, d_valid);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        *seconds += tmp.kernel_seconds + (double)ms * 1e-3;
        *launches += tmp.launches + 1;
    }
    return VK_OK;
}

// The chained transforms (sha256_chain: up to `depth` dependent hashes per key; armory: four chained k*G per key) are swept in
// two kernels: their keys go to HBM first (64 bytes of traffic per key next to ~900 for the sweep itself), then the raw-key
// instantiation of the sweep kernel reads them back.  Fused, their state sat on top of the sweep kernel's 250 registers
// (6 and 127 spilled registers, 48 and 400 bytes of scratch per lane) and every key of a chain recomputed its predecessors.
static int run_range_staged(vk_engine *e, const vk_spec_dev &spec, u32 kind, u32 slots, u64 lo, u64 span, u32 flags, const launch_io &io,
                            vk_sweep_stats *st)
{
    // 2 GB of keys per stage; armory also parks a chain code and a public key per item (128 B + 1): 2^24 items = 2.2 GB
    const u64 max_items = kind == VK_T_ARMORY ? 1ull << 24 : 1ull << 26;
    const u64 step_inputs = std::max<u64>(1, max_items / slots);
    u64 done = 0; // inputs
    for (;;) {
        const u64 left = span - done; // inclusive
        const u64 n = std::min<u64>(left, step_inputs - 1) + 1, items = n * slots;
        const size_t extra = kind == VK_T_ARMORY ? items * (32 + 64) : 0;
        SCRATCH(e, d_keys, u32 *, VK_SCR_STAGED, items * 32 + items + 64 + extra);
        unsigned char *d_valid = (unsigned char *)d_keys + items * 32;
        double pre_seconds = 0.0; // kernel time of the staging work that is not a run_range of the caller's statistics
        u32 pre_launches = 0;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        if (kind == VK_T_SHA256_CHAIN) {
            const u64 lanes = n * 2 * (spec.variant == VK_VARIANT_ALL ? 4 : 1);
            hipLaunchKernelGGL(vk_sha256_chain_range_kernel, dim3((u32)((lanes + VK_BLOCK - 1) / VK_BLOCK)), dim3(VK_BLOCK), 0, e->stream, spec, lo + done, n,
                               d_keys, d_valid);
        } else {
            hipLaunchKernelGGL(vk_armory_begin_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), dim3(VK_BLOCK), 0, e->stream, lo + done, items,
                               (const u32 *)nullptr, d_keys, (u32 *)((unsigned char *)d_keys + ((items * 33 + 63) / 64) * 64), d_valid);
        }
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        pre_seconds += (double)ms * 1e-3;
        pre_launches += 1;
        if (kind == VK_T_ARMORY) {
            u32 *d_cc = (u32 *)((unsigned char *)d_keys + ((items * 33 + 63) / 64) * 64), *d_pub = d_cc + items * 8;
            const int prc = armory_steps(e, items, d_keys, d_valid, d_cc, d_pub, &pre_seconds, &pre_launches);
            if (prc) return prc;
        }
        launch_io sub = io;
        sub.in_keys = d_keys;
        sub.in_valid = d_valid;
        const u64 item0 = done * slots; // generate-mode outputs are indexed by item
        if (sub.out_priv) sub.out_priv += item0 * 8;
        if (sub.out_h160c) sub.out_h160c += item0 * 5;
        if (sub.out_h160u) sub.out_h160u += item0 * 5;
        if (sub.out_valid) sub.out_valid += item0;
        if (sub.out_pub) sub.out_pub += item0 * 16;
        const u64 had = std::min<u64>(st->matches, io.hit_cap);
        const vk_spec_dev raw = {VK_T_RAWKEYS, 0, 0, 0};
        const int rc = run_range(e, raw, VK_T_RAWKEYS, 1, 0, items - 1, flags, sub, st);
        if (rc) return rc;
        st->kernel_seconds += pre_seconds;
        st->launches += pre_launches;
        const u64 got = std::min<u64>(st->matches, io.hit_cap);
        for (u64 i = had; io.hits && i < got; i++) { // key index of this stage -> (input, slot)
            const u64 idx = io.hits[i].seed;
            io.hits[i].seed = lo + done + idx / slots;
            io.hits[i].sub_index = (u32)(idx % slots);
        }
        if (n - 1 == left) return VK_OK;
        done += n;
    }
}

// run [lo, lo+span] (span+1 inputs) under `spec`; accumulates into *st
static int run_range(vk_engine *e, const vk_spec_dev &spec, u32 kind, u32 slots, u64 lo, u64 span, u32 flags,
                     const launch_io &io, vk_sweep_stats *st)
{
    if (kind == VK_T_SHA256_CHAIN || kind == VK_T_ARMORY) return run_range_staged(e, spec, kind, slots, lo, span, flags, io, st);
    HIPCHK(e, hipSetDevice(e->device));
    // Tile = VK_BLOCK lanes x R inputs per lane (R * slots keys share one field inversion).  R is the largest batch
    // that fits e->batch, then shrunk so that every workgroup of the grid gets the same number of full tiles
    // (a 2^28-input block over 768 workgroups would otherwise end with a third of them idle for one tile).
    // Which instantiation (vk_sweep_tu.hip): by default what measured fastest for the kind and the key forms of the launch -- three waves
    // per SIMD as three 256-lane workgroups per CU (compressed only) or one 768-lane workgroup per CU (uncompressed keys hashed), two
    // waves for the compressed-only launches of the sha256 family.  A caller's blocks_per_cu below three asks for the two-wave form.
    const u32 design = vk_sweep_design_form(kind, flags);
    const u32 form = (e->bpc_auto || e->blocks_per_cu >= 3) ? design : VK_FORM_TWO;
    const u32 bpc = e->bpc_auto ? (form == VK_FORM_TWO ? 2u : 3u) : e->blocks_per_cu;
    const bool wide = form == VK_FORM_WIDE;
    const u32 tile_lanes = wide ? 64u : (u32)VK_BLOCK;                 // lanes of a tile = of a workspace slice
    const u32 units_per_wg = wide ? (u32)VK_WIDE_BLOCK / 64u : 1u;     // tile owners per workgroup: waves (wide) or the workgroup itself
    const u64 max_wgs = wide ? (u64)e->cu_count * std::max(1u, bpc / 3u) : (u64)e->cu_count * bpc;
    const u64 max_units = max_wgs * units_per_wg;
    u32 R = std::max(1u, e->batch / slots);
    {
        const u64 per_lane = span / (max_units * tile_lanes) + 1; // inputs per lane if the grid shared them evenly
        const u64 rounds = (per_lane + R - 1) / R;                // tiles per owner
        R = (u32)std::max<u64>(1, (per_lane + rounds - 1) / rounds);
    }
    const u32 B = R * slots;
    const u64 tile_inputs = (u64)tile_lanes * R;
    const u64 ntiles = span / tile_inputs + 1;
    u64 units = std::min<u64>(ntiles, max_units);
    // The workspace is one slice of B items per tile owner.  B >= slots, and slots grows with the chain depth of
    // sha256_chain (up to 8 * 4096): keep the allocation under 16 GiB by launching fewer workgroups (the tile counter hands
    // the same tiles to a smaller grid), and say what is wrong if even one workgroup does not fit.
    const size_t per_unit = (size_t)tile_lanes * B * 8 * sizeof(uint4);
    const size_t per_wg = per_unit * units_per_wg;
    u32 grid = (u32)((units + units_per_wg - 1) / units_per_wg);
    const size_t work_cap = (size_t)16 << 30;
    if ((size_t)grid * per_wg > work_cap && (size_t)grid * per_wg > e->work_bytes) {
        const size_t fit = std::max(work_cap, e->work_bytes) / per_wg;
        if (fit == 0) {
            char msg[256];
            snprintf(msg, sizeof msg, "batch-inversion workspace for %u keys per input needs %.1f GiB per workgroup (chain depth too large for one device pass)",
                     slots, (double)per_wg / (double)((size_t)1 << 30));
            return fail(e, VK_E_NOMEM, msg);
        }
        grid = (u32)std::min<size_t>(grid, fit);
    }
    const size_t need = (size_t)grid * per_wg;
    if (need > e->work_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->work) HIPCHK(e, hipFree(e->work));
        e->work = nullptr;
        e->work_bytes = 0;
        hipError_t mc = hipMalloc(&e->work, need);
        if (mc != hipSuccess) {
            char msg[256];
            snprintf(msg, sizeof msg, "hipMalloc of the %.2f GiB batch-inversion workspace (%u workgroups x %u keys per lane) failed: %s", (double)need / (double)((size_t)1 << 30),
                     grid, B, hipGetErrorString(mc));
            e->work = nullptr;
            (void)hipGetLastError(); // read it away: see HIPCHK
            return fail(e, VK_E_NOMEM, msg);
        }
        e->work_bytes = need;
    }
    if (io.hit_cap > e->d_hits_cap) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->d_hits) HIPCHK(e, hipFree(e->d_hits));
        e->d_hits = nullptr;
        e->d_hits_cap = 0;
        HIPCHK(e, hipMalloc(&e->d_hits, io.hit_cap * sizeof(vk_hit_dev)));
        e->d_hits_cap = io.hit_cap;
    }
    HIPCHK(e, hipMemsetAsync(e->d_stats, 0, 12 * sizeof(u64), e->stream));

    vk_sweep_params p;
    memset(&p, 0, sizeof p);
    p.spec = spec;
    p.lo = lo;
    p.span = span;
    p.ntiles = ntiles;
    p.slots = slots;
    p.R = R;
    p.flags = flags;
    p.nwin = e->nwin;
    memcpy(p.wins, e->wins, sizeof p.wins);
    p.table = e->table;
    p.tgt = e->tgt;
    p.tgt_mask = e->tgt_cap ? e->tgt_cap - 1 : 0;
    p.filt = e->filt;
    p.filt_mask = e->filt_mask;
    p.hits = e->d_hits;
    p.hit_cap = (u32)std::min<size_t>(io.hit_cap, 0xFFFFFFFFu);
    p.stats = e->d_stats;
    p.work = e->work;
    p.in_keys = io.in_keys;
    p.in_valid = io.in_valid;
    p.out_priv = io.out_priv;
    p.out_h160c = io.out_h160c;
    p.out_h160u = io.out_h160u;
    p.out_valid = io.out_valid;
    p.out_pub = io.out_pub;

    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    e->last_form = form;
    e->last_grid = grid;
    e->last_R = R;
    hipError_t lrc = hipSuccess;
    if (!(vk_launch_sweep_g0(kind, p, grid, e->stream, form, &lrc) || vk_launch_sweep_g1(kind, p, grid, e->stream, form, &lrc) ||
          vk_launch_sweep_g2(kind, p, grid, e->stream, form, &lrc)))
        return fail(e, VK_E_UNSUPPORTED, "transform kind not supported on the device");
    HIPCHK(e, lrc);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    u64 hs[12];
    HIPCHK(e, hipMemcpyAsync(hs, e->d_stats, sizeof hs, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    const u64 found = hs[2];
    const u64 had = st->matches;
    if (io.hits && found) {
        // append after hits already delivered by earlier launches of this call
        const u64 room = had < io.hit_cap ? io.hit_cap - had : 0;
        const u64 take = std::min<u64>(std::min<u64>(found, p.hit_cap), room);
        if (take) HIPCHK(e, hipMemcpy(io.hits + had, e->d_hits, take * sizeof(vk_hit), hipMemcpyDeviceToHost));
    }
    st->keys += hs[0];
    st->matches += hs[1];
    st->digest[0] ^= hs[3];
    st->digest[1] ^= hs[4];
    st->digest[2] ^= hs[5];
    st->digest[3] += hs[6];
    if (hs[9]) { // shader clock over the sweep launches of this call: total shader cycles / total s_memrealtime ticks (100 MHz) of
                 // workgroup 0 -- launches without a sample (other kernels of the call) carry no weight
        if (st->launches == 0 || st->shader_clock_ghz == 0.0) e->clk_cycles = e->clk_ticks = 0.0;
        e->clk_cycles += (double)hs[8];
        e->clk_ticks += (double)hs[9];
        st->shader_clock_ghz = e->clk_cycles / (e->clk_ticks / 100e6) / 1e9;
    }
    st->kernel_seconds += (double)ms * 1e-3;
    st->launches += 1;
    return VK_OK;
}

static vk_spec_dev to_dev(const vk_transform_spec *t)
{
    vk_spec_dev s;
    s.kind = t->kind;
    s.variant = t->variant;
    s.endian = t->endian;
    s.chain_depth = t->chain_depth;
    return s;
}

// inputs per launch: keeps one launch in the 0.1 - 1 s range and the counters well inside u64
static const u64 VK_LAUNCH_INPUTS = 1ull << 30;

extern "C" int vk_sweep(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, uint32_t flags, vk_hit *hits,
                        size_t hit_cap, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    int rc = validate_spec(e, t);
    if (rc) return rc;
    if (!out) return fail(e, VK_E_INVALID_ARG, "stats out pointer is NULL");
    if (hi < lo) return fail(e, VK_E_INVALID_ARG, "seed_hi < seed_lo");
    if (!(flags & (VK_F_COMPRESSED | VK_F_UNCOMPRESSED)) || (flags & ~(VK_F_COMPRESSED | VK_F_UNCOMPRESSED)))
        return fail(e, VK_E_INVALID_ARG, "flags must be a non-empty subset of VK_F_COMPRESSED | VK_F_UNCOMPRESSED");
    if (hi - lo == ~0ull) return fail(e, VK_E_INVALID_ARG, "range too large (max 2^64 - 1 inputs per call)");
    memset(out, 0, sizeof *out);
    const u32 slots = vk_slots_per_input(t);
    const vk_spec_dev spec = to_dev(t);
    launch_io io;
    io.hits = hits;
    io.hit_cap = hits ? hit_cap : 0;
    const u64 per = std::max<u64>(1, VK_LAUNCH_INPUTS / slots);
    u64 cur = lo;
    for (;;) {
        const u64 left = hi - cur; // inclusive span remaining
        const u64 span = std::min<u64>(left, per - 1);
        rc = run_range(e, spec, t->kind, slots, cur, span, flags, io, out);
        if (rc) return rc;
        if (span == left) break;
        cur += span + 1;
    }
    out->inputs = hi - lo + 1;
    out->hits_dropped = out->matches > io.hit_cap ? out->matches - io.hit_cap : 0;
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_generate(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, uint32_t flags,
                           uint8_t *privkeys, uint8_t *h160_c, uint8_t *h160_u, uint8_t *valid, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    int rc = validate_spec(e, t);
    if (rc) return rc;
    if (hi < lo) return fail(e, VK_E_INVALID_ARG, "seed_hi < seed_lo");
    if (!(flags & (VK_F_COMPRESSED | VK_F_UNCOMPRESSED)) || (flags & ~(VK_F_COMPRESSED | VK_F_UNCOMPRESSED)))
        return fail(e, VK_E_INVALID_ARG, "flags must be a non-empty subset of VK_F_COMPRESSED | VK_F_UNCOMPRESSED");
    const u32 slots = vk_slots_per_input(t);
    if (hi - lo >= (1ull << 40) / slots) return fail(e, VK_E_INVALID_ARG, "generate range too large for one call");
    vk_sweep_stats local;
    memset(&local, 0, sizeof local);
    HIPCHK(e, hipSetDevice(e->device));
    const vk_spec_dev spec = to_dev(t);
    const u64 chunk_inputs = std::min<u64>(std::max<u64>(1, (1ull << 24) / slots), hi - lo + 1);
    const u64 chunk_items = chunk_inputs * slots;
    SCRATCH(e, d_priv, u32 *, 1, chunk_items * 32);
    SCRATCH(e, d_hc, u32 *, 2, chunk_items * 20);
    SCRATCH(e, d_hu, u32 *, 3, chunk_items * 20);
    SCRATCH(e, d_valid, unsigned char *, 4, chunk_items);
    auto cleanup = [&]() {};
    u64 cur = lo, done_items = 0;
    for (;;) {
        const u64 left = hi - cur;
        const u64 span = std::min<u64>(left, chunk_inputs - 1);
        const u64 items = (span + 1) * slots;
        hipError_t r1 = hipMemsetAsync(d_valid, 0, items, e->stream);
        hipError_t r2 = hipMemsetAsync(d_priv, 0, items * 32, e->stream);
        hipError_t r3 = hipMemsetAsync(d_hc, 0, items * 20, e->stream);
        hipError_t r4 = hipMemsetAsync(d_hu, 0, items * 20, e->stream);
        if (r1 != hipSuccess || r2 != hipSuccess || r3 != hipSuccess || r4 != hipSuccess) {
            cleanup();
            return fail(e, VK_E_HIP, "hipMemsetAsync failed in vk_generate");
        }
        launch_io io;
        io.out_priv = d_priv;
        io.out_h160c = d_hc;
        io.out_h160u = d_hu;
        io.out_valid = d_valid;
        rc = run_range(e, spec, t->kind, slots, cur, span, flags | VK_F_EMIT, io, &local);
        if (rc) {
            cleanup();
            return rc;
        }
        hipError_t c = hipSuccess;
        if (privkeys && c == hipSuccess) c = hipMemcpy(privkeys + done_items * 32, d_priv, items * 32, hipMemcpyDeviceToHost);
        if (h160_c && c == hipSuccess) c = hipMemcpy(h160_c + done_items * 20, d_hc, items * 20, hipMemcpyDeviceToHost);
        if (h160_u && c == hipSuccess) c = hipMemcpy(h160_u + done_items * 20, d_hu, items * 20, hipMemcpyDeviceToHost);
        if (valid && c == hipSuccess) c = hipMemcpy(valid + done_items, d_valid, items, hipMemcpyDeviceToHost);
        if (c != hipSuccess) {
            cleanup();
            return fail(e, VK_E_HIP, "hipMemcpy D2H failed in vk_generate");
        }
        done_items += items;
        if (span == left) break;
        cur += span + 1;
    }
    cleanup();
    local.inputs = hi - lo + 1;
    if (out) *out = local;
    return VK_OK;
}
VK_API_CATCH(e)

// RangeSource::process in generate mode with ConsoleOutput's compact or verbose format, text made on the device (vk_lines.cuh,
// vk_verbose_kernel in vk_records.cuh).
static int generate_text(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, const char *transform_name, const char *const *slot_suffix,
                         uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out, bool verbose)
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    const u32 stride = verbose ? VK_VERBOSE_STRIDE : VK_LINE_STRIDE;
    int rc = validate_spec(e, t);
    if (rc) return rc;
    if (hi < lo) return fail(e, VK_E_INVALID_ARG, "seed_hi < seed_lo");
    if (!transform_name || !text || !text_len) return fail(e, VK_E_INVALID_ARG, "transform_name / text / text_len is NULL");
    const u32 slots = vk_slots_per_input(t);
    if (slots == 0) return fail(e, VK_E_UNSUPPORTED, "the transform emits nothing for numeric inputs");
    if (hi - lo >= VK_LINES_MAX_ITEMS / slots) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines: at most 2^21 (input, slot) pairs per call");
    const u64 items = (hi - lo + 1) * slots;
    if (text_cap < items * stride)
        return fail(e, VK_E_INVALID_ARG, verbose ? "vk_generate_verbose: text_cap must be at least 512 bytes per (input, slot) pair"
                                                 : "vk_generate_lines: text_cap must be at least 192 bytes per (input, slot) pair");
    // label suffix of every slot and the transform name: one length byte + up to 31 characters each
    std::vector<unsigned char> texts((size_t)(slots + 1) * VK_LINE_SUFFIX_BYTES, 0);
    for (u32 i = 0; i <= slots; i++) {
        const char *src = i < slots ? (slot_suffix ? slot_suffix[i] : "") : transform_name;
        const size_t n = src ? strlen(src) : 0;
        if (n >= VK_LINE_SUFFIX_BYTES) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines: label suffix / transform name longer than 31 bytes");
        texts[(size_t)i * VK_LINE_SUFFIX_BYTES] = (unsigned char)n;
        if (n) memcpy(&texts[(size_t)i * VK_LINE_SUFFIX_BYTES + 1], src, n);
    }
    HIPCHK(e, hipSetDevice(e->device));
    SCRATCH(e, d_texts, unsigned char *, 0, texts.size());
    SCRATCH(e, d_priv, u32 *, 1, items * 32);
    SCRATCH(e, d_hc, u32 *, 2, items * 20);
    SCRATCH(e, d_hu, u32 *, 3, verbose ? items * 20 : 4);
    SCRATCH(e, d_valid, unsigned char *, 4, items);
    SCRATCH(e, d_slots, u64 *, 6, items * stride);
    const size_t temp_bytes = scan_temp_bytes(items);
    const size_t off_lens = 0, off_offs = (items + 1) * 4, off_temp = ((off_offs + (items + 1) * 4 + 255) / 256) * 256;
    SCRATCH(e, d_scan, unsigned char *, 7, off_temp + temp_bytes);
    u32 *d_lens = (u32 *)(d_scan + off_lens), *d_offs32 = (u32 *)(d_scan + off_offs);
    SCRATCH(e, d_text, unsigned char *, 8, items * stride);
    HIPCHK(e, hipMemcpy(d_texts, texts.data(), texts.size(), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemsetAsync(d_valid, 0, items, e->stream));
    vk_sweep_stats local;
    memset(&local, 0, sizeof local);
    launch_io io;
    io.out_priv = d_priv;
    io.out_h160c = d_hc;
    io.out_valid = d_valid;
    if (verbose) io.out_h160u = d_hu;
    rc = run_range(e, to_dev(t), t->kind, slots, lo, hi - lo, VK_F_COMPRESSED | (verbose ? VK_F_UNCOMPRESSED : 0u) | VK_F_EMIT, io, &local);
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    const u32 grid = (u32)((items + 255) / 256);
    if (verbose)
        hipLaunchKernelGGL(vk_verbose_kernel, dim3(grid), dim3(256), 0, e->stream, (u64)lo, items, slots, d_priv, d_hc, d_hu, d_valid, d_texts, d_slots, d_lens);
    else
        hipLaunchKernelGGL(vk_lines_kernel, dim3(grid), dim3(256), 0, e->stream, (u64)lo, items, slots, d_priv, d_hc, d_valid, d_texts, (const u32 *)nullptr,
                           d_slots, d_lens);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, exclusive_scan(e, d_scan + off_temp, d_lens, d_offs32, items));
    hipLaunchKernelGGL(vk_lines_compact_kernel, dim3((u32)((items + 63) / 64 * 64 + 255) / 256), dim3(256), 0, e->stream, items, slots, stride, d_slots, d_lens, d_offs32,
                       (const unsigned char *)nullptr, (const u32 *)nullptr, d_text);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    u32 tail[2] = {0, 0};
    HIPCHK(e, hipMemcpyAsync(&tail[0], d_offs32 + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(&tail[1], d_lens + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    const size_t total = (size_t)tail[0] + tail[1];
    if (total) HIPCHK(e, hipMemcpy(text, d_text, total, hipMemcpyDeviceToHost));
    *text_len = total;
    local.inputs = hi - lo + 1;
    local.kernel_seconds += (double)ms * 1e-3;
    local.launches += 2;
    if (out) *out = local;
    return VK_OK;
}

extern "C" int vk_generate_lines(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, const char *transform_name,
                                 const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out) try
{
    return generate_text(e, t, lo, hi, transform_name, slot_suffix, text, text_cap, text_len, out, false);
}
VK_API_CATCH(e)

extern "C" int vk_generate_verbose(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, const char *transform_name,
                                   const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out) try
{
    return generate_text(e, t, lo, hi, transform_name, slot_suffix, text, text_cap, text_len, out, true);
}
VK_API_CATCH(e)

extern "C" size_t vk_record_column_stride(int column) { return column >= 0 && column < VK_RC_COLUMNS ? vk_rc_stride(column) : 0; }

// the device side of the record columns after the sweep has left keys, hash160s and public keys in the scratch slots 1-5:
// slots per column -> lengths -> offsets -> packed pages -> the caller's buffers
static int records_pack(vk_engine *e, u64 items, u32 slots, u64 lo, const unsigned char *d_texts /* nullptr: no source column */, vk_record_columns *cols,
                        double *seconds)
{
    const u32 *d_priv = (const u32 *)e->scr[1], *d_hc = (const u32 *)e->scr[2], *d_hu = (const u32 *)e->scr[3], *d_pub = (const u32 *)e->scr[5];
    const unsigned char *d_valid = (const unsigned char *)e->scr[4];
    size_t base[VK_RC_COLUMNS + 1];
    base[0] = 0;
    for (int c = 0; c < VK_RC_COLUMNS; c++) base[c + 1] = base[c] + (size_t)items * vk_rc_stride(c);
    SCRATCH(e, d_slots, unsigned char *, 6, base[VK_RC_COLUMNS]);
    const size_t temp_bytes = scan_temp_bytes(items);
    // per column: items lengths; then one array of offsets (reused column after column) and the scan's tile states
    const size_t off_offs = (size_t)VK_RC_COLUMNS * items * 4, off_temp = ((off_offs + (items + 1) * 4 + 255) / 256) * 256;
    SCRATCH(e, d_scan, unsigned char *, 7, off_temp + temp_bytes);
    SCRATCH(e, d_packed, unsigned char *, 8, base[VK_RC_COLUMNS]);
    u32 *d_offs32 = (u32 *)(d_scan + off_offs);
    vk_record_slots rs;
    for (int c = 0; c < VK_RC_COLUMNS; c++) {
        rs.slot[c] = (u64 *)(d_slots + base[c]);
        rs.lens[c] = (u32 *)d_scan + (size_t)c * items;
    }
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    const u32 grid = (u32)((items + 255) / 256);
    hipLaunchKernelGGL(vk_records_key_kernel, dim3(grid), dim3(256), 0, e->stream, lo, items, slots, d_priv, d_valid, d_texts, rs);
    HIPCHK(e, hipGetLastError());
    hipLaunchKernelGGL(vk_records_addr_kernel, dim3(grid), dim3(256), 0, e->stream, items, d_hc, d_hu, d_pub, d_valid, rs);
    HIPCHK(e, hipGetLastError());
    u32 tails[2 * VK_RC_COLUMNS]; // per column: offset and length of the last item
    for (int c = 0; c < VK_RC_COLUMNS; c++) {
        HIPCHK(e, exclusive_scan(e, d_scan + off_temp, rs.lens[c], d_offs32, items));
        hipLaunchKernelGGL(vk_lines_compact_kernel, dim3((u32)((items + 63) / 64 * 64 + 255) / 256), dim3(256), 0, e->stream, items, slots, vk_rc_stride(c),
                           rs.slot[c], rs.lens[c], d_offs32, (const unsigned char *)nullptr, (const u32 *)nullptr, d_packed + base[c]);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipMemcpyAsync(&tails[2 * c], d_offs32 + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(&tails[2 * c + 1], rs.lens[c] + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    for (int c = 0; c < VK_RC_COLUMNS; c++) {
        const size_t total = (size_t)tails[2 * c] + tails[2 * c + 1];
        if (total) HIPCHK(e, hipMemcpyAsync(cols->data[c], d_packed + base[c], total, hipMemcpyDeviceToHost, e->stream));
        cols->len[c] = total;
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    cols->rows = cols->len[VK_RC_KEY_RAW] / 32;
    *seconds = (double)ms * 1e-3;
    return VK_OK;
}

static int check_columns(vk_engine *e, vk_record_columns *cols, u64 items, bool with_source, const char *who)
{
    if (!cols) return fail(e, VK_E_INVALID_ARG, (std::string(who) + ": cols is NULL").c_str());
    for (int c = 0; c < VK_RC_COLUMNS; c++) {
        cols->len[c] = 0;
        if (c == VK_RC_SOURCE && !with_source) continue;
        if (!cols->data[c] || cols->cap[c] < items * vk_rc_stride(c))
            return fail(e, VK_E_INVALID_ARG, (std::string(who) + ": every column buffer must hold vk_record_column_stride(column) bytes per row").c_str());
    }
    cols->rows = 0;
    return VK_OK;
}

// RangeSource::process in generate mode into a sink that stores the full record (StorageOutput): the value columns of the record
// made on the device (vk_records.cuh), each as the bytes of a Parquet PLAIN page.
extern "C" int vk_generate_columns(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, const char *const *slot_suffix,
                                   vk_record_columns *cols, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    int rc = validate_spec(e, t);
    if (rc) return rc;
    if (hi < lo) return fail(e, VK_E_INVALID_ARG, "seed_hi < seed_lo");
    const u32 slots = vk_slots_per_input(t);
    if (slots == 0) return fail(e, VK_E_UNSUPPORTED, "the transform emits nothing for numeric inputs");
    if (hi - lo >= VK_RECORDS_MAX_ITEMS / slots) return fail(e, VK_E_INVALID_ARG, "vk_generate_columns: at most 2^20 (input, slot) pairs per call");
    const u64 items = (hi - lo + 1) * slots;
    rc = check_columns(e, cols, items, true, "vk_generate_columns");
    if (rc) return rc;
    std::vector<unsigned char> texts((size_t)slots * VK_LINE_SUFFIX_BYTES, 0);
    for (u32 i = 0; i < slots; i++) {
        const char *src = slot_suffix ? slot_suffix[i] : "";
        const size_t n = src ? strlen(src) : 0;
        if (n >= VK_LINE_SUFFIX_BYTES) return fail(e, VK_E_INVALID_ARG, "vk_generate_columns: label suffix longer than 31 bytes");
        texts[(size_t)i * VK_LINE_SUFFIX_BYTES] = (unsigned char)n;
        if (n) memcpy(&texts[(size_t)i * VK_LINE_SUFFIX_BYTES + 1], src, n);
    }
    HIPCHK(e, hipSetDevice(e->device));
    SCRATCH(e, d_texts, unsigned char *, 0, texts.size());
    SCRATCH(e, d_priv, u32 *, 1, items * 32);
    SCRATCH(e, d_hc, u32 *, 2, items * 20);
    SCRATCH(e, d_hu, u32 *, 3, items * 20);
    SCRATCH(e, d_valid, unsigned char *, 4, items);
    SCRATCH(e, d_pub, u32 *, 5, items * 64);
    HIPCHK(e, hipMemcpy(d_texts, texts.data(), texts.size(), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemsetAsync(d_valid, 0, items, e->stream));
    vk_sweep_stats local;
    memset(&local, 0, sizeof local);
    launch_io io;
    io.out_priv = d_priv;
    io.out_h160c = d_hc;
    io.out_h160u = d_hu;
    io.out_pub = d_pub;
    io.out_valid = d_valid;
    rc = run_range(e, to_dev(t), t->kind, slots, lo, hi - lo, VK_F_COMPRESSED | VK_F_UNCOMPRESSED | VK_F_EMIT, io, &local);
    if (rc) return rc;
    double sec = 0.0;
    rc = records_pack(e, items, slots, lo, d_texts, cols, &sec);
    if (rc) return rc;
    local.inputs = hi - lo + 1;
    local.kernel_seconds += sec;
    local.launches += 2 + 2 * VK_RC_COLUMNS;
    if (out) *out = local;
    return VK_OK;
}
VK_API_CATCH(e)

// KeyDeriver::derive (src/derive.rs:78-165) for n raw keys, the whole record as columns: what a source that already holds keys
// (wordlist, stdin, files) hands to a storing sink.  The source column is the caller's (len[VK_RC_SOURCE] = 0).
extern "C" int vk_derive_columns(vk_engine *e, const uint8_t *keys, size_t n, vk_record_columns *cols) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    int rc = check_columns(e, cols, n, false, "vk_derive_columns");
    if (rc) return rc;
    if (n == 0) return VK_OK;
    if (!keys) return fail(e, VK_E_INVALID_ARG, "keys is NULL");
    if (n > VK_RECORDS_MAX_ITEMS) return fail(e, VK_E_INVALID_ARG, "vk_derive_columns: at most 2^20 keys per call");
    HIPCHK(e, hipSetDevice(e->device));
    SCRATCH(e, d_in, u32 *, 0, n * 32);
    SCRATCH(e, d_priv, u32 *, 1, n * 32);
    SCRATCH(e, d_hc, u32 *, 2, n * 20);
    SCRATCH(e, d_hu, u32 *, 3, n * 20);
    SCRATCH(e, d_valid, unsigned char *, 4, n);
    SCRATCH(e, d_pub, u32 *, 5, n * 64);
    HIPCHK(e, hipMemcpy(d_in, keys, n * 32, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemsetAsync(d_valid, 1, n, e->stream)); // KeyDeriver::derive has a record for every key (0 -> 1, k >= n -> k mod n)
    vk_sweep_stats local;
    memset(&local, 0, sizeof local);
    launch_io io;
    io.in_keys = d_in;
    io.out_priv = d_priv;
    io.out_h160c = d_hc;
    io.out_h160u = d_hu;
    io.out_pub = d_pub;
    const vk_spec_dev raw = {VK_T_RAWKEYS, 0, 0, 0};
    rc = run_range(e, raw, VK_T_RAWKEYS, 1, 0, n - 1, VK_F_COMPRESSED | VK_F_UNCOMPRESSED | VK_F_EMIT, io, &local);
    if (rc) return rc;
    double sec = 0.0;
    return records_pack(e, n, 1, 0, nullptr, cols, &sec);
}
VK_API_CATCH(e)

extern "C" int vk_derive_batch_pub(vk_engine *e, const uint8_t *keys, size_t n, uint8_t *normalised, uint8_t *h160_c, uint8_t *h160_u,
                                   uint8_t *pub_xy) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (n == 0) return VK_OK;
    if (!keys) return fail(e, VK_E_INVALID_ARG, "keys is NULL");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t chunk = 1u << 22;
    const size_t cn = std::min(n, chunk);
    SCRATCH(e, d_in, u32 *, 0, cn * 32);
    SCRATCH(e, d_priv, u32 *, 1, cn * 32);
    SCRATCH(e, d_hc, u32 *, 2, cn * 20);
    SCRATCH(e, d_hu, u32 *, 3, cn * 20);
    u32 *d_pub = nullptr;
    if (pub_xy) {
        d_pub = (u32 *)scratch(e, 5, cn * 64);
        if (!d_pub) return VK_E_HIP;
    }
    auto cleanup = [&]() {};
    vk_spec_dev spec = {VK_T_RAWKEYS, 0, 0, 0};
    const u32 flags = (h160_u ? VK_F_UNCOMPRESSED : 0u) | ((h160_c || !h160_u) ? VK_F_COMPRESSED : 0u) | VK_F_EMIT;
    for (size_t done = 0; done < n; done += chunk) {
        const size_t m = std::min(chunk, n - done);
        hipError_t c = hipMemcpy(d_in, keys + done * 32, m * 32, hipMemcpyHostToDevice);
        if (c != hipSuccess) {
            cleanup();
            return fail(e, VK_E_HIP, "hipMemcpy H2D failed in vk_derive_batch");
        }
        launch_io io;
        io.in_keys = d_in;
        io.out_priv = d_priv;
        io.out_h160c = d_hc;
        io.out_h160u = d_hu;
        io.out_pub = d_pub;
        vk_sweep_stats st;
        memset(&st, 0, sizeof st);
        int rc = run_range(e, spec, VK_T_RAWKEYS, 1, 0, m - 1, flags, io, &st);
        if (rc) {
            cleanup();
            return rc;
        }
        if (normalised && c == hipSuccess) c = hipMemcpy(normalised + done * 32, d_priv, m * 32, hipMemcpyDeviceToHost);
        if (h160_c && c == hipSuccess) c = hipMemcpy(h160_c + done * 20, d_hc, m * 20, hipMemcpyDeviceToHost);
        if (h160_u && c == hipSuccess) c = hipMemcpy(h160_u + done * 20, d_hu, m * 20, hipMemcpyDeviceToHost);
        if (pub_xy && c == hipSuccess) c = hipMemcpy(pub_xy + done * 64, d_pub, m * 64, hipMemcpyDeviceToHost);
        if (c != hipSuccess) {
            cleanup();
            return fail(e, VK_E_HIP, "hipMemcpy D2H failed in vk_derive_batch");
        }
    }
    cleanup();
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_derive_batch(vk_engine *e, const uint8_t *keys, size_t n, uint8_t *normalised, uint8_t *h160_c, uint8_t *h160_u) try
{
    return vk_derive_batch_pub(e, keys, n, normalised, h160_c, h160_u, nullptr);
}
VK_API_CATCH(e)

template <u32 KIND>
static void launch_transform(const vk_spec_dev &s, const u64 *seeds, u64 n, u32 slots, u32 *keys, unsigned char *valid, hipStream_t st,
                             const vk_ec_ctx &ec = vk_ec_ctx{nullptr, nullptr, 0})
{
    const u64 items = n * slots;
    hipLaunchKernelGGL(vk_transform_kernel<KIND>, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), dim3(VK_BLOCK), 0, st, s, seeds, (u64)0, n,
                       slots, keys, valid, ec);
}

extern "C" int vk_transform_batch(vk_engine *e, const vk_transform_spec *t, const uint64_t *seeds, size_t n, uint8_t *keys_out,
                                  uint8_t *valid) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    int rc = validate_spec(e, t);
    if (rc) return rc;
    if (n == 0) return VK_OK;
    if (!seeds || !keys_out || !valid) return fail(e, VK_E_INVALID_ARG, "seeds / keys_out / valid is NULL");
    const u32 slots = vk_slots_per_input(t);
    if ((u64)n * slots > (1ull << 31)) return fail(e, VK_E_INVALID_ARG, "too many keys for one vk_transform_batch call (max 2^31)");
    HIPCHK(e, hipSetDevice(e->device));
    const vk_spec_dev spec = to_dev(t);
    const u64 items = (u64)n * slots;
    SCRATCH(e, d_seeds, u64 *, 0, n * 8);
    SCRATCH(e, d_keys, u32 *, 1, items * 32);
    SCRATCH(e, d_valid, unsigned char *, 4, items);
    auto cleanup = [&]() {};
    hipError_t c = hipMemcpy(d_seeds, seeds, n * 8, hipMemcpyHostToDevice);
    if (c == hipSuccess) {
        switch (t->kind) {
        case VK_T_SHA256: launch_transform<VK_T_SHA256>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_MILKSAD: launch_transform<VK_T_MILKSAD>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_MT64: launch_transform<VK_T_MT64>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_LCG: launch_transform<VK_T_LCG>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_XORSHIFT: launch_transform<VK_T_XORSHIFT>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_SHA256_CHAIN: launch_transform<VK_T_SHA256_CHAIN>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_DOUBLE_SHA256: launch_transform<VK_T_DOUBLE_SHA256>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_MD5: launch_transform<VK_T_MD5>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_DIRECT: launch_transform<VK_T_DIRECT>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream); break;
        case VK_T_ARMORY: launch_transform<VK_T_ARMORY>(spec, d_seeds, n, slots, d_keys, d_valid, e->stream, vk_ec_ctx{e->table, e->d_wins, e->nwin}); break;
        default: cleanup(); return fail(e, VK_E_UNSUPPORTED, "transform kind not supported on the device");
        }
        c = hipGetLastError();
    }
    if (c == hipSuccess) c = hipStreamSynchronize(e->stream);
    if (c == hipSuccess) c = hipMemcpy(keys_out, d_keys, items * 32, hipMemcpyDeviceToHost);
    if (c == hipSuccess) c = hipMemcpy(valid, d_valid, items, hipMemcpyDeviceToHost);
    cleanup();
    if (c != hipSuccess) {
        e->err = std::string("vk_transform_batch: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_bip39_wordlist_set(vk_engine *e, const uint8_t *words) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (!words) return fail(e, VK_E_INVALID_ARG, "words is NULL");
    for (int i = 0; i < 2048; i++) {
        const uint8_t *w = words + 8 * i;
        if (w[0] == 0) return fail(e, VK_E_INVALID_ARG, "empty word in the BIP-39 list");
        bool ended = false;
        for (int j = 0; j < 8; j++) {
            if (w[j] == 0) ended = true;
            else if (ended || w[j] == ' ' || w[j] == '\t' || w[j] == '\n' || w[j] == '\r' || w[j] == 0x0b || w[j] == 0x0c)
                return fail(e, VK_E_INVALID_ARG, "BIP-39 words must be zero padded and free of whitespace");
        }
    }
    // the same words as an open-addressed set for validate_mnemonic (vk_multibit.cuh: wordset_has)
    std::vector<u64> set(VK_WORDSET_SLOTS, 0);
    for (int i = 0; i < 2048; i++) {
        u64 w;
        memcpy(&w, words + 8 * i, 8);
        u32 h = (u32)((w * 0x9E3779B97F4A7C15ull) >> 52);
        while (set[h] != 0 && set[h] != w) h = (h + 1) & (VK_WORDSET_SLOTS - 1);
        set[h] = w;
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->d_wordlist) HIPCHK(e, hipMalloc(&e->d_wordlist, (2048 + VK_WORDSET_SLOTS) * 8));
    HIPCHK(e, hipMemcpy(e->d_wordlist, words, 2048 * 8, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->d_wordlist + 2048, set.data(), VK_WORDSET_SLOTS * 8, hipMemcpyHostToDevice));
    return VK_OK;
}
VK_API_CATCH(e)

static int set_mb_salt(vk_engine *e, const uint8_t *pass, size_t len)
{
    std::vector<u64> w((size_t)vk_salt_nblk(len) * 16);
    const u32 nblk = vk_salt_blocks(w.data(), pass, len);
    HIPCHK(e, hipSetDevice(e->device));
    if (e->d_mb_salt && e->mb_salt_cap < w.size()) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipFree(e->d_mb_salt));
        e->d_mb_salt = nullptr;
    }
    if (!e->d_mb_salt) {
        e->mb_salt_cap = std::max<size_t>(w.size(), 48);
        HIPCHK(e, hipMalloc(&e->d_mb_salt, e->mb_salt_cap * 8));
    }
    HIPCHK(e, hipMemcpy(e->d_mb_salt, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    e->mb_salt_nblk = nblk;
    return VK_OK;
}

extern "C" int vk_multibit_passphrase_set(vk_engine *e, const uint8_t *passphrase, size_t len) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (len && !passphrase) return fail(e, VK_E_INVALID_ARG, "passphrase is NULL");
    return set_mb_salt(e, passphrase, len);
}
VK_API_CATCH(e)

// One stretch of a key-derivation chain through the ring of vk_chain.cuh: `groups` groups of 64 inputs, `iters` iterations
// in chunks of `chunk_iters`.  Workers <= groups: with no more workers than groups a pop never waits.
template <class Body>
static int run_ring(vk_engine *e, u32 groups, u32 iters, u32 chunk_iters, typename Body::args a)
{
    if (groups == 0 || iters == 0) return VK_OK;
    if (e->dbg_ring_chunk) chunk_iters = e->dbg_ring_chunk;
    const u32 chunks = (iters + chunk_iters - 1) / chunk_iters;
    if ((u64)groups * chunks >= (1ull << 32)) return fail(e, VK_E_INVALID_ARG, "too many chain chunks for one call");
    u32 nslots = 1;
    while (nslots < 2 * groups) nslots <<= 1;
    const size_t bytes = sizeof(vk_ring) + (size_t)nslots * 8 + (size_t)groups * 4;
    SCRATCH(e, ring_mem, unsigned char *, 9, bytes);
    vk_ring *ring = (vk_ring *)ring_mem;
    unsigned long long *slots = (unsigned long long *)(ring_mem + sizeof(vk_ring));
    u32 *chunk_done = (u32 *)(ring_mem + sizeof(vk_ring) + (size_t)nslots * 8);
    if (!e->ring_waves_per_simd) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, vk_ring_kernel<vk_pbkdf2_body>, 256, 0) != hipSuccess || nb < 1) nb = 1;
        int ne = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ne, vk_ring_kernel<vk_electrum_body>, 256, 0) != hipSuccess || ne < 1) ne = 1;
        e->ring_waves_per_simd = std::max(1, std::min(nb, ne)); // a 256-thread block is one wave on each SIMD of its CU
    }
    // Workers: as many waves per SIMD as keep EVERY worker busy (workers <= groups) and every SIMD equally loaded -- a whole
    // number of waves per SIMD.  Measured (tools/probe_ring.py, profiles/archive/r03_ring_probe.txt): at the config's 100 000 inputs
    // = 1 563 groups, 1 024 workers (one per SIMD) finish in 84 ms / 609 ms (MultiBit / Electrum); 1 563 workers take 108 /
    // 737 ms and 2 048 workers 93 / 639 ms (a quarter of them spin for a group at any time).  Why even ONE doubled CU costs 5 %
    // (profiles/r06_ring_worker_probe.txt): the ring is FIFO, every group gets one chunk per trip through the queue, a group that
    // meets a two-wave worker falls most of a chunk time behind for good, and the call ends with the most delayed group.
    int wps = e->ring_waves_per_simd;
    if (e->dbg_ring_wps) wps = std::min(wps, (int)e->dbg_ring_wps);
    const u32 simds = (u32)e->cu_count * 4u;
    const u32 per_simd = std::max(1u, std::min((u32)wps, groups / simds));
    u32 workers = groups < simds ? groups : simds * per_simd;
    const u32 resident = simds * (u32)wps;
    if (e->dbg_ring_workers) workers = std::min(e->dbg_ring_workers, resident); // experiments: may exceed groups
    a.iters = iters;
    a.chunk_iters = chunk_iters;
    hipLaunchKernelGGL(vk_ring_init_kernel, dim3((nslots + 255) / 256), dim3(256), 0, e->stream, ring, slots, chunk_done, groups, chunks, nslots);
    hipLaunchKernelGGL(vk_ring_kernel<Body>, dim3((workers + 3) / 4), dim3(256), 0, e->stream, ring, slots, chunk_done, chunks, a);
    HIPCHK(e, hipGetLastError());
    vk_ring h;
    HIPCHK(e, hipMemcpyAsync(&h, ring, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (h.error || h.head < h.total) return fail(e, VK_E_HIP, "the chain ring stalled (a bounded wait ran out): internal error");
    return VK_OK;
}

// Device half of Transform::apply_batch for string inputs: uploads the strings, runs the transform kernel on the engine's
// stream and leaves keys (items x 32 B) and validity flags (items) in scratch slots 1 and 4.  Strings longer than the
// device limit of the two KDF chains get no keys; *overlong counts the ones the reference would not have dropped too.
static int strings_to_device_keys(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n, u32 slots,
                                  u32 **d_keys_out, unsigned char **d_valid_out, size_t *overlong, double *kernel_seconds)
{
    const bool kdf = t->kind == VK_T_ELECTRUM || t->kind == VK_T_MULTIBIT;
    // No transform limits the length of a string: Electrum hashes the whole seed (src/electrum.rs:57-78; beyond the
    // register-resident 64 bytes a seed runs on the one-lane path that reads the caller's bytes), MultiBit's password is the line
    // as given (src/multibit.rs:87-101; beyond the 448 bytes of the staging slice its HMAC key is hashed from the caller's bytes).
    // `overlong` stays in the interface and is always 0 since ABI 5.
    *overlong = 0;
    size_t n_long = 0; // electrum seeds beyond the register-resident fast path
    u32 max_long = 0;  // the longest of them
    for (size_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(e, VK_E_INVALID_ARG, "offsets must be non-decreasing");
        const u32 len = offsets[i + 1] - offsets[i];
        if (t->kind == VK_T_ELECTRUM && len > VK_ELECTRUM_FAST_SEED) {
            n_long++;
            max_long = std::max(max_long, len);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bytes = offsets[n], items = n * slots;
    SCRATCH(e, d_data, unsigned char *, 0, bytes + VK_STRING_SLACK); // aligned dword reads may reach a few bytes beyond the last string
    SCRATCH(e, d_offs, u32 *, 2, (n + 1) * 4);
    SCRATCH(e, d_keys, u32 *, 1, items * 32);
    SCRATCH(e, d_valid, unsigned char *, 4, items);
    hipError_t c = hipSuccess;
    if (bytes) c = hipMemcpy(d_data, data, bytes, hipMemcpyHostToDevice);
    if (c == hipSuccess) c = hipMemcpy(d_offs, offsets, (n + 1) * 4, hipMemcpyHostToDevice);
    if (c == hipSuccess) c = hipMemsetAsync(d_keys, 0, items * 32, e->stream);
    if (c == hipSuccess) c = hipMemsetAsync(d_valid, 0, items, e->stream);
    if (c == hipSuccess) c = hipEventRecord(e->ev0, e->stream);
    if (c != hipSuccess) {
        e->err = std::string("string transform on the device: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    const dim3 grid((u32)((n + VK_BLOCK - 1) / VK_BLOCK)), block(VK_BLOCK);
    const u32 groups = (u32)((n + 63) / 64);
    double extra_seconds = 0.0; // kernel time of sub-steps that record the engine's events themselves
    if (t->kind == VK_T_ARMORY) {
        // root seed = SHA256(string) from the strings kernel, then the batched chain of the range sweep (run_range_staged)
        vk_spec_dev sha = to_dev(t);
        sha.kind = VK_T_SHA256;
        SCRATCH(e, d_cc, u32 *, 5, items * (32 + 64));
        hipLaunchKernelGGL(vk_strings_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), block, 0, e->stream, sha, d_data, d_offs, (u64)n, slots, d_keys,
                           d_valid, vk_ec_ctx{e->table, e->d_wins, e->nwin});
        hipLaunchKernelGGL(vk_armory_begin_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), block, 0, e->stream, (u64)0, (u64)items, (const u32 *)d_keys,
                           d_keys, d_cc, d_valid);
        HIPCHK(e, hipGetLastError());
        u32 nl = 0;
        const int rc = armory_steps(e, items, d_keys, d_valid, d_cc, d_cc + items * 8, &extra_seconds, &nl); // records its own events
        if (rc) return rc;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        // a key the chain rejected (a tweak >= n or zero) is invalid and must read as zeros like every other invalid slot
        hipLaunchKernelGGL(vk_zero_invalid_keys_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), block, 0, e->stream, (u64)items, d_keys, d_valid);
    } else if (!kdf) {
        hipLaunchKernelGGL(vk_strings_kernel, dim3((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), block, 0, e->stream, to_dev(t), d_data, d_offs, (u64)n, slots,
                           d_keys, d_valid, vk_ec_ctx{e->table, e->d_wins, e->nwin});
    } else if (t->kind == VK_T_MULTIBIT) {
        // prepare -> 2047 HMAC iterations -> 48 words, prepare again -> 2047 iterations -> master key and children
        if (!e->d_mb_salt) {
            const int rc = set_mb_salt(e, nullptr, 0);
            if (rc) return rc;
        }
        SCRATCH(e, d_stage, u64 *, 6, n * VK_STAGE_BYTES);
        SCRATCH(e, d_state, u64 *, 7, (size_t)groups * VK_PB_WORDS * 64 * 8);
        SCRATCH(e, d_ok, unsigned char *, 8, n);
        const u64 *wordset = e->d_wordlist + 2048;
        hipLaunchKernelGGL(vk_mb_prepare_kernel, grid, block, 0, e->stream, d_data, d_offs, (u64)n, wordset, e->d_mb_salt, e->mb_salt_nblk, d_stage, d_state, d_ok);
        vk_pbkdf2_body::args a = {d_state, 0, 0};
        const int stop = e->chain_stop; // test hook (vk_debug_chain_stop): leave the chain after this many of its five kernels
        if (stop <= 1) goto chain_done;
        int rc;
        rc = run_ring<vk_pbkdf2_body>(e, groups, 2047, 64, a);
        if (rc) return rc;
        if (stop <= 2) goto chain_done;
        hipLaunchKernelGGL(vk_mb_middle_kernel, grid, block, 0, e->stream, (u64)n, e->d_wordlist, e->d_mb_salt, e->mb_salt_nblk, d_stage, d_state, d_ok);
        if (stop <= 3) goto chain_done;
        rc = run_ring<vk_pbkdf2_body>(e, groups, 2047, 64, a);
        if (rc) return rc;
        if (stop <= 4) goto chain_done;
        hipLaunchKernelGGL(vk_mb_final_kernel, grid, block, 0, e->stream, (u64)n, t->chain_depth, d_state, d_ok, e->table, e->d_wins, e->nwin, d_keys, d_valid);
    } else {
        SCRATCH(e, d_stage, u64 *, 6, n * VK_STAGE_BYTES);
        SCRATCH(e, d_state, u32 *, 7, (size_t)groups * VK_EL_WORDS * 64 * 4);
        SCRATCH(e, d_ok, unsigned char *, 8, n);
        hipLaunchKernelGGL(vk_electrum_prepare_kernel, grid, block, 0, e->stream, d_data, d_offs, (u64)n, d_stage, d_state, d_ok);
        vk_electrum_body::args a = {d_state, 0, 0};
        const int rc = run_ring<vk_electrum_body>(e, groups, 100000u - 1u, 500, a);
        if (rc) return rc;
        if (n_long) {
            // seeds of more than 64 bytes: rare, stretched one lane each outside the ring.  The rounds are cut so that no launch runs
            // more than VK_EL_LONG_BUDGET compressions per lane: a megabyte of hex digits on one line is 1.6 x 10^9 serial
            // compressions (the reference spends a minute on it, one lane of a GPU more than an hour) -- slow, but never ONE
            // dispatch that cannot be told from a hung GPU, and a killed process gives the card back at the next boundary
            const u32 nblk_max = (32 + max_long + 9 + 63) / 64, rounds = 100000u;
            const u32 per = std::max(1u, VK_EL_LONG_BUDGET / nblk_max);
            for (u32 r = 0; r < rounds; r += per)
                hipLaunchKernelGGL(vk_electrum_long_kernel, grid, block, 0, e->stream, d_data, d_offs, (u64)n, r, std::min(rounds, r + per), rounds, d_state, d_ok);
        }
        hipLaunchKernelGGL(vk_electrum_final_kernel, grid, block, 0, e->stream, (u64)n, t->variant, t->chain_depth, d_state, d_ok, d_stage, e->table, e->d_wins,
                           e->nwin, d_keys, d_valid);
    }
chain_done:
    c = hipGetLastError();
    if (c == hipSuccess) c = hipEventRecord(e->ev1, e->stream);
    if (c == hipSuccess) c = hipStreamSynchronize(e->stream);
    float ms = 0.f;
    if (c == hipSuccess) c = hipEventElapsedTime(&ms, e->ev0, e->ev1);
    if (c != hipSuccess) {
        e->err = std::string("string transform on the device: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    if (kernel_seconds) *kernel_seconds = (double)ms * 1e-3 + extra_seconds;
    *d_keys_out = d_keys;
    *d_valid_out = d_valid;
    return VK_OK;
}

static void note_overlong(vk_engine *e, const vk_transform_spec *t, size_t overlong)
{
    (void)t;
    if (overlong) e->warn = "warning: " + std::to_string(overlong) + " input string(s) produced no keys"; // unreachable since ABI 5: no length limits
}

extern "C" size_t vk_last_warning(vk_engine *e, char *buf, size_t cap) try
{
    if (!e) return 0;
    vk_api_lock lk(e->mu, false);
    const size_t n = e->warn.size();
    if (buf && cap) {
        const size_t c = std::min(n, cap - 1);
        memcpy(buf, e->warn.data(), c);
        buf[c] = 0;
    }
    return n;
}
catch (...)
{
    return 0;
}

extern "C" int vk_transform_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n,
                                    uint8_t *keys_out, uint8_t *valid) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    e->warn.clear();
    int rc = validate_spec(e, t, true);
    if (rc) return rc;
    if (n == 0) return VK_OK;
    if (!data || !offsets || !keys_out || !valid) return fail(e, VK_E_INVALID_ARG, "data / offsets / keys_out / valid is NULL");
    const u32 slots = vk_slots_per_string(t);
    if ((u64)n * slots > (1ull << 28)) return fail(e, VK_E_INVALID_ARG, "too many keys for one vk_transform_strings call (max 2^28)");
    u32 *d_keys = nullptr;
    unsigned char *d_valid = nullptr;
    size_t overlong = 0;
    rc = strings_to_device_keys(e, t, data, offsets, n, slots, &d_keys, &d_valid, &overlong, nullptr);
    if (rc) return rc;
    const size_t items = n * slots;
    hipError_t c = hipMemcpy(keys_out, d_keys, items * 32, hipMemcpyDeviceToHost);
    if (c == hipSuccess) c = hipMemcpy(valid, d_valid, items, hipMemcpyDeviceToHost);
    if (c != hipSuccess) {
        e->err = std::string("vk_transform_strings: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    note_overlong(e, t, overlong);
    return VK_OK;
}
VK_API_CATCH(e)

// Source::process for the string-input sources in scan mode (reference src/source/wordlist.rs, src/source/stdin.rs:
// apply_batch -> derive -> Matcher::check per key): the transform's keys never leave HBM -- the sweep kernel reads them
// back as VK_T_RAWKEYS with the validity flags as its mask.  hit.seed = index of the input string, hit.sub_index = slot.
extern "C" int vk_scan_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n, uint32_t flags,
                               vk_hit *hits, size_t hit_cap, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    e->warn.clear();
    int rc = validate_spec(e, t, true);
    if (rc) return rc;
    if (!out) return fail(e, VK_E_INVALID_ARG, "stats out pointer is NULL");
    if (!(flags & (VK_F_COMPRESSED | VK_F_UNCOMPRESSED)) || (flags & ~(VK_F_COMPRESSED | VK_F_UNCOMPRESSED)))
        return fail(e, VK_E_INVALID_ARG, "flags must be a non-empty subset of VK_F_COMPRESSED | VK_F_UNCOMPRESSED");
    memset(out, 0, sizeof *out);
    if (n == 0) return VK_OK;
    if (!data || !offsets) return fail(e, VK_E_INVALID_ARG, "data / offsets is NULL");
    const u32 slots = vk_slots_per_string(t);
    if ((u64)n * slots > (1ull << 28)) return fail(e, VK_E_INVALID_ARG, "too many keys for one vk_scan_strings call (max 2^28)");
    u32 *d_keys = nullptr;
    unsigned char *d_valid = nullptr;
    size_t overlong = 0;
    double tsec = 0.0;
    rc = strings_to_device_keys(e, t, data, offsets, n, slots, &d_keys, &d_valid, &overlong, &tsec);
    if (rc) return rc;
    launch_io io;
    io.in_keys = d_keys;
    io.in_valid = d_valid;
    io.hits = hits;
    io.hit_cap = hits ? hit_cap : 0;
    vk_spec_dev spec = {VK_T_RAWKEYS, 0, 0, 0};
    rc = run_range(e, spec, VK_T_RAWKEYS, 1, 0, (u64)n * slots - 1, flags, io, out);
    if (rc) return rc;
    out->inputs = n;
    out->kernel_seconds += tsec;
    out->launches += 1;
    out->hits_dropped = out->matches > (hits ? hit_cap : 0) ? out->matches - (hits ? hit_cap : 0) : 0;
    const size_t got = (size_t)std::min<u64>(out->matches, hits ? hit_cap : 0);
    for (size_t i = 0; i < got; i++) { // key index -> (input, slot)
        const u64 idx = hits[i].seed;
        hits[i].seed = idx / slots;
        hits[i].sub_index = (u32)(idx % slots);
    }
    note_overlong(e, t, overlong);
    out->inputs_skipped = (uint32_t)std::min<size_t>(overlong, 0xFFFFFFFFu);
    e->rescan_items = (u64)n * slots;
    e->rescan_slots = slots;
    e->rescan_inputs = n;
    e->rescan_skipped = out->inputs_skipped;
    return VK_OK;
}
VK_API_CATCH(e)

// vk_scan_strings again over the keys its last call left on the device: only the derive + probe sweep runs, not the transform
// (for a caller whose hit buffer was too small: the KDF chains cost seconds per batch, the sweep milliseconds).
extern "C" int vk_rescan_strings(vk_engine *e, uint32_t flags, vk_hit *hits, size_t hit_cap, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (!out) return fail(e, VK_E_INVALID_ARG, "stats out pointer is NULL");
    if (!(flags & (VK_F_COMPRESSED | VK_F_UNCOMPRESSED)) || (flags & ~(VK_F_COMPRESSED | VK_F_UNCOMPRESSED)))
        return fail(e, VK_E_INVALID_ARG, "flags must be a non-empty subset of VK_F_COMPRESSED | VK_F_UNCOMPRESSED");
    if (!e->rescan_items || !e->scr[1] || !e->scr[4])
        return fail(e, VK_E_INVALID_ARG, "vk_rescan_strings: the keys of the last vk_scan_strings call are no longer on the device (another call has run since)");
    memset(out, 0, sizeof *out);
    launch_io io;
    io.in_keys = (const u32 *)e->scr[1];
    io.in_valid = (const unsigned char *)e->scr[4];
    io.hits = hits;
    io.hit_cap = hits ? hit_cap : 0;
    vk_spec_dev spec = {VK_T_RAWKEYS, 0, 0, 0};
    const int rc = run_range(e, spec, VK_T_RAWKEYS, 1, 0, e->rescan_items - 1, flags, io, out);
    if (rc) return rc;
    out->inputs = e->rescan_inputs;
    out->inputs_skipped = (uint32_t)e->rescan_skipped;
    out->hits_dropped = out->matches > io.hit_cap ? out->matches - io.hit_cap : 0;
    const size_t got = (size_t)std::min<u64>(out->matches, io.hit_cap);
    for (size_t i = 0; i < got; i++) { // key index -> (input, slot)
        const u64 idx = hits[i].seed;
        hits[i].seed = idx / e->rescan_slots;
        hits[i].sub_index = (u32)(idx % e->rescan_slots);
    }
    return VK_OK;
}
VK_API_CATCH(e)

// WordlistSource / StdinSource in generate mode with ConsoleOutput's compact format: the label of a line is the input string (plus
// the slot's suffix), which the compaction kernel takes straight from the input buffer.
extern "C" int vk_generate_lines_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n,
                                         const uint8_t *label_data, const uint32_t *label_offsets, const char *transform_name,
                                         const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    e->warn.clear();
    int rc = validate_spec(e, t, true);
    if (rc) return rc;
    if ((label_data == nullptr) != (label_offsets == nullptr)) return fail(e, VK_E_INVALID_ARG, "label_data and label_offsets go together");
    if (!transform_name || !text || !text_len) return fail(e, VK_E_INVALID_ARG, "transform_name / text / text_len is NULL");
    *text_len = 0;
    if (out) memset(out, 0, sizeof *out);
    if (n == 0) return VK_OK;
    if (!data || !offsets) return fail(e, VK_E_INVALID_ARG, "data / offsets is NULL");
    const u32 slots = vk_slots_per_string(t);
    if (slots == 0) return fail(e, VK_E_UNSUPPORTED, "the transform emits nothing for string inputs");
    if ((u64)n * slots > VK_LINES_MAX_ITEMS) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines_strings: at most 2^21 (input, slot) pairs per call");
    const u64 items = (u64)n * slots;
    const uint32_t *lab_offs = label_offsets ? label_offsets : offsets; // what a line starts with: the caller's label or the input itself
    // both offset arrays are checked before anything is computed from them: a decreasing entry would wrap a length to ~4 G on the
    // device (vk_lines_kernel), and lab_offs[n] sizes the text below
    if (label_offsets && label_offsets[0] != 0) return fail(e, VK_E_INVALID_ARG, "label_offsets[0] must be 0");
    for (size_t i = 0; i < n; i++)
        if (lab_offs[i + 1] < lab_offs[i]) return fail(e, VK_E_INVALID_ARG, label_offsets ? "label_offsets must be non-decreasing" : "offsets must be non-decreasing");
    const u64 need = (u64)lab_offs[n] * slots + items * VK_LINE_STRIDE;
    if (need >= (1ull << 32)) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines_strings: more than 4 GB of text in one call; pass fewer lines");
    if (text_cap < need) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines_strings: text_cap must be at least slots x the label bytes + 192 bytes per (input, slot) pair");
    std::vector<unsigned char> texts((size_t)(slots + 1) * VK_LINE_SUFFIX_BYTES, 0);
    for (u32 i = 0; i <= slots; i++) {
        const char *src = i < slots ? (slot_suffix ? slot_suffix[i] : "") : transform_name;
        const size_t len = src ? strlen(src) : 0;
        if (len >= VK_LINE_SUFFIX_BYTES) return fail(e, VK_E_INVALID_ARG, "vk_generate_lines_strings: label suffix / transform name longer than 31 bytes");
        texts[(size_t)i * VK_LINE_SUFFIX_BYTES] = (unsigned char)len;
        if (len) memcpy(&texts[(size_t)i * VK_LINE_SUFFIX_BYTES + 1], src, len);
    }
    u32 *d_keys = nullptr;
    unsigned char *d_valid = nullptr;
    size_t overlong = 0;
    double tsec = 0.0;
    rc = strings_to_device_keys(e, t, data, offsets, n, slots, &d_keys, &d_valid, &overlong, &tsec); // strings -> slot 0, offsets -> slot 2
    if (rc) return rc;
    if (label_data) { // the inputs have done their part: their two slots take the labels
        SCRATCH(e, d_lab, unsigned char *, 0, std::max<size_t>(label_offsets[n], 1));
        SCRATCH(e, d_lo, u32 *, 2, (n + 1) * 4);
        if (label_offsets[n]) HIPCHK(e, hipMemcpy(d_lab, label_data, label_offsets[n], hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(d_lo, label_offsets, (n + 1) * 4, hipMemcpyHostToDevice));
    }
    const unsigned char *d_data = (const unsigned char *)e->scr[0];
    const u32 *d_offs = (const u32 *)e->scr[2];
    SCRATCH(e, d_texts, unsigned char *, 9, texts.size());
    SCRATCH(e, d_priv, u32 *, 3, items * 32);
    SCRATCH(e, d_hc, u32 *, 5, items * 20);
    SCRATCH(e, d_slots, u64 *, 6, items * VK_LINE_STRIDE);
    const size_t temp_bytes = scan_temp_bytes(items);
    const size_t off_offs = (items + 1) * 4, off_temp = ((off_offs + (items + 1) * 4 + 255) / 256) * 256;
    SCRATCH(e, d_scan, unsigned char *, 7, off_temp + temp_bytes);
    u32 *d_lens = (u32 *)d_scan, *d_offs32 = (u32 *)(d_scan + off_offs);
    SCRATCH(e, d_text, unsigned char *, 8, (size_t)need);
    HIPCHK(e, hipMemcpy(d_texts, texts.data(), texts.size(), hipMemcpyHostToDevice));
    vk_sweep_stats local;
    memset(&local, 0, sizeof local);
    launch_io io;
    io.in_keys = d_keys;
    io.in_valid = d_valid;
    io.out_priv = d_priv;
    io.out_h160c = d_hc;
    const vk_spec_dev raw = {VK_T_RAWKEYS, 0, 0, 0};
    rc = run_range(e, raw, VK_T_RAWKEYS, 1, 0, items - 1, VK_F_COMPRESSED | VK_F_EMIT, io, &local);
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(vk_lines_kernel, dim3((u32)((items + 255) / 256)), dim3(256), 0, e->stream, (u64)0, items, slots, d_priv, d_hc, d_valid, d_texts, d_offs, d_slots,
                       d_lens);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, exclusive_scan(e, d_scan + off_temp, d_lens, d_offs32, items));
    hipLaunchKernelGGL(vk_lines_compact_kernel, dim3((u32)((items + 63) / 64 * 64 + 255) / 256), dim3(256), 0, e->stream, items, slots, VK_LINE_STRIDE, d_slots, d_lens, d_offs32, d_data,
                       d_offs, d_text);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    u32 tail[2] = {0, 0};
    HIPCHK(e, hipMemcpyAsync(&tail[0], d_offs32 + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(&tail[1], d_lens + (items - 1), 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    const size_t total = (size_t)tail[0] + tail[1];
    if (total) HIPCHK(e, hipMemcpy(text, d_text, total, hipMemcpyDeviceToHost));
    *text_len = total;
    local.inputs = n;
    local.kernel_seconds += tsec + (double)ms * 1e-3;
    local.launches += 3;
    note_overlong(e, t, overlong);
    local.inputs_skipped = (uint32_t)std::min<size_t>(overlong, 0xFFFFFFFFu);
    if (out) *out = local;
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_bitimage_keys(vk_engine *e, const uint8_t *entropy, size_t n_files, const uint8_t *pass_data, const uint32_t *pass_offsets,
                                size_t n_pass, const uint32_t *path, uint32_t depth, uint32_t count, uint8_t *keys_out, uint8_t *valid) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (!e->d_wordlist) return fail(e, VK_E_INVALID_ARG, "bitimage needs the BIP-39 word list: call vk_bip39_wordlist_set first");
    if (n_files == 0 || count == 0) return VK_OK;
    if (!entropy || !pass_offsets || !keys_out || !valid || n_pass == 0) return fail(e, VK_E_INVALID_ARG, "entropy / pass_offsets / keys_out / valid is NULL or n_pass is 0");
    if (depth > (1u << 20)) return fail(e, VK_E_INVALID_ARG, "derivation path of more than 2^20 levels");
    if (depth && !path) return fail(e, VK_E_INVALID_ARG, "path is NULL");
    for (size_t i = 0; i < n_pass; i++) {
        if (pass_offsets[i + 1] < pass_offsets[i]) return fail(e, VK_E_INVALID_ARG, "pass_offsets must be non-decreasing");
    }
    const u64 items = (u64)n_files * n_pass;
    if (items > (1ull << 28) || items * count > (1ull << 28)) return fail(e, VK_E_INVALID_ARG, "too many keys for one vk_bitimage_keys call (max 2^28)");
    vk_bip32_path pd = {};
    pd.depth = depth;
    // salt message of every passphrase, ready-padded (pass_data may be NULL when every passphrase is empty)
    std::vector<u32> salt_off(n_pass + 1, 0); // in words: passphrases of any length (src/bitimage.rs:38-74 takes the string as given)
    for (size_t j = 0; j < n_pass; j++) {
        const size_t pl = pass_offsets[j + 1] - pass_offsets[j];
        if (pl && !pass_data) return fail(e, VK_E_INVALID_ARG, "pass_data is NULL");
        const u64 next = (u64)salt_off[j] + (u64)vk_salt_nblk(pl) * 16;
        if (next > 0xFFFFFFFFull) return fail(e, VK_E_INVALID_ARG, "the passphrases of one vk_bitimage_keys call add up to more than 32 GiB of salt blocks");
        salt_off[j + 1] = (u32)next;
    }
    std::vector<u64> salts(salt_off[n_pass]);
    for (size_t j = 0; j < n_pass; j++) {
        const size_t pl = pass_offsets[j + 1] - pass_offsets[j];
        vk_salt_blocks(&salts[salt_off[j]], pl ? pass_data + pass_offsets[j] : nullptr, pl);
    }
    HIPCHK(e, hipSetDevice(e->device));
    const size_t outs = (size_t)(items * count); // the passphrases travel as their salt blocks (d_salts below)
    SCRATCH(e, d_ent, unsigned char *, 3, n_files * 32);
    SCRATCH(e, d_keys, u32 *, 1, outs * 32);
    SCRATCH(e, d_valid, unsigned char *, 4, outs);
    hipError_t c = hipMemcpy(d_ent, entropy, n_files * 32, hipMemcpyHostToDevice);
    if (c == hipSuccess) c = hipMemsetAsync(d_keys, 0, outs * 32, e->stream);
    if (c == hipSuccess) c = hipMemsetAsync(d_valid, 0, outs, e->stream);
    if (c != hipSuccess) {
        e->err = std::string("vk_bitimage_keys: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    {
        const dim3 grid((u32)((items + VK_BLOCK - 1) / VK_BLOCK)), block(VK_BLOCK);
        const u32 groups = (u32)((items + 63) / 64);
        SCRATCH(e, d_stage, u64 *, 6, (size_t)items * VK_STAGE_BYTES);
        SCRATCH(e, d_state, u64 *, 7, (size_t)groups * VK_PB_WORDS * 64 * 8);
        SCRATCH(e, d_salts, u64 *, 8, salts.size() * 8 + (n_pass + 1) * 4 + (size_t)depth * 4);
        u32 *d_salt_off = (u32 *)(d_salts + salts.size());
        u32 *d_comp = d_salt_off + n_pass + 1;
        if (depth) HIPCHK(e, hipMemcpy(d_comp, path, (size_t)depth * 4, hipMemcpyHostToDevice));
        pd.comp = d_comp;
        HIPCHK(e, hipMemcpy(d_salts, salts.data(), salts.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(d_salt_off, salt_off.data(), (n_pass + 1) * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(vk_bitimage_prepare_kernel, grid, block, 0, e->stream, d_ent, items, (u32)n_pass, e->d_wordlist, d_salts, d_salt_off, d_stage, d_state);
        vk_pbkdf2_body::args a = {d_state, 0, 0};
        const int rc = run_ring<vk_pbkdf2_body>(e, groups, 2047, 64, a);
        if (rc) return rc;
        hipLaunchKernelGGL(vk_bitimage_final_kernel, grid, block, 0, e->stream, items, pd, count, d_state, e->table, e->d_wins, e->nwin, d_keys, d_valid);
        c = hipGetLastError();
    }
    if (c == hipSuccess) c = hipStreamSynchronize(e->stream);
    if (c == hipSuccess) c = hipMemcpy(keys_out, d_keys, outs * 32, hipMemcpyDeviceToHost);
    if (c == hipSuccess) c = hipMemcpy(valid, d_valid, outs, hipMemcpyDeviceToHost);
    if (c != hipSuccess) {
        e->err = std::string("vk_bitimage_keys: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_scan_keys(vk_engine *e, const uint8_t *keys, size_t n, uint32_t flags, vk_hit *hits, size_t hit_cap, vk_sweep_stats *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (!out) return fail(e, VK_E_INVALID_ARG, "stats out pointer is NULL");
    if (!(flags & (VK_F_COMPRESSED | VK_F_UNCOMPRESSED)) || (flags & ~(VK_F_COMPRESSED | VK_F_UNCOMPRESSED)))
        return fail(e, VK_E_INVALID_ARG, "flags must be a non-empty subset of VK_F_COMPRESSED | VK_F_UNCOMPRESSED");
    memset(out, 0, sizeof *out);
    if (n == 0) return VK_OK;
    if (!keys) return fail(e, VK_E_INVALID_ARG, "keys is NULL");
    HIPCHK(e, hipSetDevice(e->device));
    SCRATCH(e, d_in, u32 *, 0, n * 32);
    hipError_t c = hipMemcpy(d_in, keys, n * 32, hipMemcpyHostToDevice);
    int rc = VK_OK;
    if (c != hipSuccess) rc = fail(e, VK_E_HIP, "hipMemcpy H2D failed in vk_scan_keys");
    if (rc == VK_OK) {
        launch_io io;
        io.in_keys = d_in;
        io.hits = hits;
        io.hit_cap = hits ? hit_cap : 0;
        vk_spec_dev spec = {VK_T_RAWKEYS, 0, 0, 0};
        rc = run_range(e, spec, VK_T_RAWKEYS, 1, 0, n - 1, flags, io, out);
    }
    if (rc) return rc;
    out->inputs = n;
    out->hits_dropped = out->matches > (hits ? hit_cap : 0) ? out->matches - (hits ? hit_cap : 0) : 0;
    return VK_OK;
}
VK_API_CATCH(e)

// ------------------------------------------------------------------------- analyzer search
template <u32 KIND, int ND>
static void launch_search(const vk_search_params &p, u32 grid, hipStream_t s)
{
    hipLaunchKernelGGL((vk_search_kernel<KIND, ND>), dim3(grid), dim3(VK_BLOCK), 0, s, p);
}
template <u32 KIND, int ND>
static void launch_emit(const vk_spec_dev &spec, u64 seed, u32 *out, hipStream_t s)
{
    hipLaunchKernelGGL((vk_search_emit_kernel<KIND, ND>), dim3(1), dim3(64), 0, s, spec, seed, out);
}

extern "C" int vk_search(vk_engine *e, const vk_transform_spec *t, uint64_t lo, uint64_t hi, const vk_search_target *tg, vk_search_result *out) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (!t || !tg || !out) return fail(e, VK_E_INVALID_ARG, "spec / target / out is NULL");
    if (hi < lo) return fail(e, VK_E_INVALID_ARG, "seed_hi < seed_lo");
    const bool slot_kind = t->kind == VK_T_SHA256 || t->kind == VK_T_DOUBLE_SHA256 || t->kind == VK_T_MD5 || t->kind == VK_T_DIRECT;
    if (t->kind != VK_T_MILKSAD && t->kind != VK_T_MT64 && t->kind != VK_T_LCG && t->kind != VK_T_XORSHIFT && t->kind != VK_T_SHA256_CHAIN && !slot_kind)
        return fail(e, VK_E_UNSUPPORTED, "vk_search is built for milksad, mt64, lcg, xorshift, sha256_chain and the hash transforms");
    if (slot_kind && tg->mode == VK_SEARCH_CASCADE) return fail(e, VK_E_UNSUPPORTED, "the hash transforms are not generators: no cascade mode");
    if (t->kind == VK_T_LCG && (t->variant > 3 || t->endian > 1)) return fail(e, VK_E_INVALID_ARG, "lcg search takes one variant (0..3) and an endian");
    if (t->kind == VK_T_XORSHIFT) {
        if (t->variant > 3) return fail(e, VK_E_INVALID_ARG, "xorshift search takes one variant (0..3)");
        if (tg->mode != VK_SEARCH_CASCADE) return fail(e, VK_E_UNSUPPORTED, "the xorshift analyzer has cascade mode only (64-bit seed space)");
    }
    if (t->kind == VK_T_SHA256_CHAIN) {
        if (t->variant != VK_VARIANT_ALL && t->variant > 3) return fail(e, VK_E_INVALID_ARG, "sha256_chain variant must be 0..3 or VK_VARIANT_ALL");
        if (tg->mode != VK_SEARCH_CASCADE && (t->chain_depth == 0 || t->chain_depth > 4096)) return fail(e, VK_E_INVALID_ARG, "chain_depth must be 1..4096");
        if (lo > 0xFFFFFFFFull) return fail(e, VK_E_INVALID_ARG, "sha256_chain search takes u32 seeds");
        if (hi > 0xFFFFFFFFull) hi = 0xFFFFFFFFull;
    }
    if (tg->mode > VK_SEARCH_CASCADE) return fail(e, VK_E_INVALID_ARG, "unknown search mode");
    if (tg->mode == VK_SEARCH_MASKED && (tg->mask_bits < 1 || tg->mask_bits > 64)) return fail(e, VK_E_INVALID_ARG, "mask_bits must be 1..64");
    if (tg->mode == VK_SEARCH_CASCADE) {
        if (tg->n_cascade < 1 || tg->n_cascade > VK_SEARCH_MAX_DRAWS) return fail(e, VK_E_INVALID_ARG, "n_cascade must be 1..8");
        if (t->kind == VK_T_LCG) return fail(e, VK_E_UNSUPPORTED, "the lcg analyzer has no cascade mode");
        for (u32 j = 0; j < tg->n_cascade; j++)
            if (tg->bits[j] < 1 || tg->bits[j] > 64) return fail(e, VK_E_INVALID_ARG, "cascade bits must be 1..64");
    }
    memset(out, 0, sizeof *out);
    HIPCHK(e, hipSetDevice(e->device));
    vk_search_params p;
    memset(&p, 0, sizeof p);
    p.spec = to_dev(t);
    p.mode = tg->mode;
    p.ndraws = tg->mode == VK_SEARCH_CASCADE ? tg->n_cascade : 1;
    for (int i = 0; i < 8; i++)
        p.key[i] = ((u32)tg->key[4 * i] << 24) | ((u32)tg->key[4 * i + 1] << 16) | ((u32)tg->key[4 * i + 2] << 8) | tg->key[4 * i + 3];
    if (tg->mode == VK_SEARCH_MASKED) {
        p.bits[0] = tg->mask_bits;
        p.targets[0] = ((u64)p.key[6] << 32) | p.key[7];
    } else if (tg->mode == VK_SEARCH_CASCADE) {
        for (u32 j = 0; j < tg->n_cascade; j++) {
            p.bits[j] = tg->bits[j];
            p.targets[j] = tg->targets[j];
        }
    }
    u64 *d_best = e->d_stats + 7; // spare counter slot
    p.best = d_best;
    const bool wide = tg->mode == VK_SEARCH_CASCADE && tg->n_cascade > 1;
    const u64 per = 1ull << 30;
    const u32 grid = (u32)e->cu_count * 8;
    u64 cur = lo, best = ~0ull, checked = 0;
    for (;;) {
        const u64 left = hi - cur, span = std::min<u64>(left, per - 1);
        const u64 init = ~0ull;
        HIPCHK(e, hipMemcpyAsync(d_best, &init, 8, hipMemcpyHostToDevice, e->stream));
        p.lo = cur;
        p.span = span;
        HIPCHK(e, hipEventRecord(e->ev0, e->stream));
        if (t->kind == VK_T_MILKSAD) wide ? launch_search<VK_T_MILKSAD, 8>(p, grid, e->stream) : launch_search<VK_T_MILKSAD, 1>(p, grid, e->stream);
        else if (t->kind == VK_T_MT64) wide ? launch_search<VK_T_MT64, 8>(p, grid, e->stream) : launch_search<VK_T_MT64, 1>(p, grid, e->stream);
        else if (t->kind == VK_T_XORSHIFT) wide ? launch_search<VK_T_XORSHIFT, 8>(p, grid, e->stream) : launch_search<VK_T_XORSHIFT, 1>(p, grid, e->stream);
        else if (t->kind == VK_T_SHA256_CHAIN) launch_search<VK_T_SHA256_CHAIN, 1>(p, grid, e->stream);
        else if (t->kind == VK_T_SHA256) hipLaunchKernelGGL(vk_search_slots_kernel<VK_T_SHA256>, dim3(grid), dim3(VK_BLOCK), 0, e->stream, p);
        else if (t->kind == VK_T_DOUBLE_SHA256) hipLaunchKernelGGL(vk_search_slots_kernel<VK_T_DOUBLE_SHA256>, dim3(grid), dim3(VK_BLOCK), 0, e->stream, p);
        else if (t->kind == VK_T_MD5) hipLaunchKernelGGL(vk_search_slots_kernel<VK_T_MD5>, dim3(grid), dim3(VK_BLOCK), 0, e->stream, p);
        else if (t->kind == VK_T_DIRECT) hipLaunchKernelGGL(vk_search_slots_kernel<VK_T_DIRECT>, dim3(grid), dim3(VK_BLOCK), 0, e->stream, p);
        else launch_search<VK_T_LCG, 1>(p, grid, e->stream);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipEventRecord(e->ev1, e->stream));
        HIPCHK(e, hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        out->kernel_seconds += (double)ms * 1e-3;
        checked += span + 1;
        if (best != ~0ull || span == left) break;
        cur += span + 1;
    }
    out->seeds_checked = checked;
    if (best == ~0ull) return VK_OK;
    out->found = 1;
    out->seed = best;
    out->n_keys = p.ndraws;
    u32 *d_out = nullptr;
    u32 h_out[66];
    HIPCHK(e, hipMalloc(&d_out, sizeof h_out));
    hipError_t c = hipMemsetAsync(d_out, 0, sizeof h_out, e->stream);
    if (c == hipSuccess) {
        if (t->kind == VK_T_MILKSAD) launch_emit<VK_T_MILKSAD, 8>(p.spec, best, d_out, e->stream);
        else if (t->kind == VK_T_MT64) launch_emit<VK_T_MT64, 8>(p.spec, best, d_out, e->stream);
        else if (t->kind == VK_T_XORSHIFT) launch_emit<VK_T_XORSHIFT, 8>(p.spec, best, d_out, e->stream);
        else if (t->kind == VK_T_SHA256_CHAIN) hipLaunchKernelGGL(vk_search_emit_chain_kernel, dim3(1), dim3(64), 0, e->stream, p, best, d_out);
        else if (t->kind == VK_T_SHA256) hipLaunchKernelGGL(vk_search_emit_slots_kernel<VK_T_SHA256>, dim3(1), dim3(64), 0, e->stream, p, best, d_out);
        else if (t->kind == VK_T_DOUBLE_SHA256) hipLaunchKernelGGL(vk_search_emit_slots_kernel<VK_T_DOUBLE_SHA256>, dim3(1), dim3(64), 0, e->stream, p, best, d_out);
        else if (t->kind == VK_T_MD5) hipLaunchKernelGGL(vk_search_emit_slots_kernel<VK_T_MD5>, dim3(1), dim3(64), 0, e->stream, p, best, d_out);
        else if (t->kind == VK_T_DIRECT) hipLaunchKernelGGL(vk_search_emit_slots_kernel<VK_T_DIRECT>, dim3(1), dim3(64), 0, e->stream, p, best, d_out);
        else launch_emit<VK_T_LCG, 1>(p.spec, best, d_out, e->stream);
        c = hipGetLastError();
    }
    if (c == hipSuccess) c = hipStreamSynchronize(e->stream);
    if (c == hipSuccess) c = hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    memcpy(out->keys, h_out, 8 * 32);
    out->variant = t->kind == VK_T_SHA256_CHAIN ? h_out[64] : (t->variant == VK_VARIANT_ALL ? 0u : t->variant);
    out->index = (t->kind == VK_T_SHA256_CHAIN || slot_kind) ? h_out[65] : 0u;
    if (slot_kind) out->n_keys = 1;
    if (t->kind == VK_T_SHA256_CHAIN && tg->mode != VK_SEARCH_CASCADE) out->n_keys = 1;
    if (c != hipSuccess) {
        e->err = std::string("vk_search: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    return VK_OK;
}
VK_API_CATCH(e)

// ------------------------------------------------------------------------- test hooks
// Only in libvuke_hip_hooks.so (build.sh compiles this file a second time with -DVK_TEST_HOOKS); declared in
// include/vuke_hip_test_hooks.h.  The shipped library exports none of them.
#ifdef VK_TEST_HOOKS
#include "../../include/vuke_hip_test_hooks.h"
extern "C" int vk_debug_table_entry(vk_engine *e, uint32_t window, uint32_t digit, uint8_t xy[64]) try
{
    if (!e || !xy) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (window >= e->nwin) return fail(e, VK_E_INVALID_ARG, "table index out of range");
    {
        const vk_win_desc &w = e->wins[window];
        const u64 count = (window + 1 < e->nwin ? e->wins[window + 1].base : e->table_entries) - w.base;
        if (digit >= count) return fail(e, VK_E_INVALID_ARG, "table index out of range");
    }
    HIPCHK(e, hipSetDevice(e->device));
    u32 raw[16];
    HIPCHK(e, hipMemcpy(raw, e->table + (e->wins[window].base + digit) * 4, 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) {
        u32 xv = raw[7 - i], yv = raw[8 + 7 - i];
        xy[4 * i] = (uint8_t)(xv >> 24); xy[4 * i + 1] = (uint8_t)(xv >> 16); xy[4 * i + 2] = (uint8_t)(xv >> 8); xy[4 * i + 3] = (uint8_t)xv;
        xy[32 + 4 * i] = (uint8_t)(yv >> 24); xy[32 + 4 * i + 1] = (uint8_t)(yv >> 16); xy[32 + 4 * i + 2] = (uint8_t)(yv >> 8); xy[32 + 4 * i + 3] = (uint8_t)yv;
    }
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_debug_chain_stop(vk_engine *e, int stages) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu, false);
    e->chain_stop = stages <= 0 ? 99 : stages;
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_debug_scratch(vk_engine *e, int slot, size_t offset, uint8_t *out, size_t n) try
{
    if (!e || !out) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    if (slot < 0 || slot >= VK_SCR_SLOTS || !e->scr[slot] || offset + n > e->scr_cap[slot]) return fail(e, VK_E_INVALID_ARG, "vk_debug_scratch: slot / range");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpy(out, (const unsigned char *)e->scr[slot] + offset, n, hipMemcpyDeviceToHost));
    return VK_OK;
}
VK_API_CATCH(e)

extern "C" int vk_debug_field(vk_engine *e, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) try
{
    if (!e || !a || !out || n == 0) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    HIPCHK(e, hipSetDevice(e->device));
    u32 *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(e, hipMalloc(&da, n * 32));
    HIPCHK(e, hipMalloc(&dout, n * 32));
    hipError_t c = hipMemcpy(da, a, n * 32, hipMemcpyHostToDevice);
    if (b && c == hipSuccess) {
        c = hipMalloc(&db, n * 32);
        if (c == hipSuccess) c = hipMemcpy(db, b, n * 32, hipMemcpyHostToDevice);
    }
    if (c == hipSuccess) {
        hipLaunchKernelGGL(vk_field_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, e->stream, op, da, db, (u64)n, dout);
        c = hipGetLastError();
    }
    if (c == hipSuccess) c = hipStreamSynchronize(e->stream);
    if (c == hipSuccess) c = hipMemcpy(out, dout, n * 32, hipMemcpyDeviceToHost);
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    if (c != hipSuccess) {
        e->err = std::string("vk_debug_field: ") + hipGetErrorString(c);
        return VK_E_HIP;
    }
    return VK_OK;
}
VK_API_CATCH(e)

// chunk length (iterations), resident waves per SIMD and worker waves of the chain ring for the following chain calls; 0 = the
// library's own choice.  (Rounds 2-3 read VUKE_RING_* from the environment on every call, in the shipped library: ADVICE r3.)
extern "C" int vk_debug_ring_knobs(vk_engine *e, uint32_t chunk_iters, uint32_t waves_per_simd, uint32_t workers) try
{
    if (!e) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu, false);
    if (chunk_iters > 100000 || waves_per_simd > 16 || workers > 65536) return fail(e, VK_E_INVALID_ARG, "vk_debug_ring_knobs: value out of range");
    e->dbg_ring_chunk = chunk_iters;
    e->dbg_ring_wps = waves_per_simd;
    e->dbg_ring_workers = workers;
    return VK_OK;
}
VK_API_CATCH(e)

// exclusive prefix sum of n u32 through vk_scan_kernel (the offsets of generate mode's compacted text / columns)
extern "C" int vk_debug_scan(vk_engine *e, const uint32_t *in, size_t n, uint32_t *out) try
{
    if (!e || !in || !out || n == 0 || n > (1u << 26)) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t off_out = ((n * 4 + 255) / 256) * 256, off_temp = 2 * off_out;
    SCRATCH(e, d, unsigned char *, 7, off_temp + scan_temp_bytes(n));
    HIPCHK(e, hipMemcpy(d, in, n * 4, hipMemcpyHostToDevice));
    HIPCHK(e, exclusive_scan(e, d + off_temp, (const u32 *)d, (u32 *)(d + off_out), n));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, d + off_out, n * 4, hipMemcpyDeviceToHost));
    return VK_OK;
}
VK_API_CATCH(e)
// Allocation-failure injection (tests/test_abi.py, tests/test_gpu_parity.py): the hooks build replaces the global operator new for the
// library's OWN references (build.sh links libvuke_hip_hooks.so with -Bsymbolic-functions; malloc / free underneath, so memory may
// cross to libstdc++'s operator delete and back).  vk_debug_fail_alloc(k) arms the calling thread: its k-th next allocation from
// inside the library (0 = the very next one) throws std::bad_alloc, once; k < 0 disarms.  Returns the allocations the thread has made
// through the library since the previous call.  What the tests hold: every armed call returns VK_E_NOMEM (or succeeds when the
// k-th allocation does not exist), none terminates the process, and the engine works afterwards.
#if !defined(__HIP_DEVICE_COMPILE__)
static thread_local long long g_fail_after = -1, g_allocs = 0;
static void *vk_hook_alloc(size_t n, bool may_throw)
{
    g_allocs++;
    if (g_fail_after >= 0 && g_fail_after-- == 0) {
        if (may_throw) throw std::bad_alloc();
        return nullptr;
    }
    void *p = malloc(n ? n : 1);
    if (!p && may_throw) throw std::bad_alloc();
    return p;
}
void *operator new(size_t n) { return vk_hook_alloc(n, true); }
void *operator new[](size_t n) { return vk_hook_alloc(n, true); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { return vk_hook_alloc(n, false); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return vk_hook_alloc(n, false); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }
#endif
extern "C" long long vk_debug_fail_alloc(long long after)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    const long long seen = g_allocs;
    g_allocs = 0;
    g_fail_after = after;
    return seen;
#else
    return 0;
#endif
}

extern "C" int vk_debug_last_launch(vk_engine *e, uint32_t out[4]) try
{
    if (!e || !out) return VK_E_INVALID_ARG;
    vk_api_lock lk(e->mu, false);
    out[0] = e->last_form;
    out[1] = e->last_grid;
    out[2] = e->last_form == VK_FORM_WIDE ? (u32)VK_WIDE_BLOCK : (u32)VK_BLOCK;
    out[3] = e->last_R;
    return VK_OK;
}
VK_API_CATCH(e)
#endif // VK_TEST_HOOKS
