// ccd_api.cpp - C ABI of libccd.so (include/ccd.h): batch bookkeeping, device memory, stage launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "ccd_device.hpp"
#include "ccd_format.hpp"

namespace ccd {
// kernels (ccd_entropy.hip, ccd_float.hip)
size_t entropy_lds_bytes(int dim, int arm_len);
hipError_t launch_entropy(const EntropyParams* d_slots, int n_slots, size_t lds_bytes, hipStream_t stream);
size_t entropy_pipe_lds_bytes(int dim, int n_layers, int ring_rows, int mfma);
int entropy_pipe_ring_rows(int max_grid_w);
bool entropy_pipe_supports_mfma(int dim, int n_layers, int n_ifce_out, int narrow, int max_grid_w, long long max_abs_weight);
bool entropy_pipe_supports(int dim, int n_layers, int narrow, int max_grid_w);
hipError_t launch_entropy_pipe(const EntropyParams* d_slots, int n_slots, int nv, int mfma, int dyn, int shape, size_t lds_bytes, hipStream_t stream);
int entropy_pipe_fixed_shape(int dim, int n_layers, int n_spatial);
hipError_t launch_laplace_bounds(const int32_t* mu_idx, const int32_t* scale_idx, const int32_t* sym,
                                 const float* scale_table, int64_t n, uint32_t* left, uint32_t* right, hipStream_t stream);
hipError_t launch_laplace_sweep_pipe(const float* scale_table, const double* rcp_table, int scale_first, int n_scales, uint32_t* out, hipStream_t stream);
hipError_t launch_laplace_sweep_generic(const float* scale_table, int scale_first, int n_scales, uint32_t* out, hipStream_t stream);
hipError_t launch_upsample_step(const UpsampleLevel* d_levels, const uint32_t* d_zmap, int n_z, int max_w, int max_h, hipStream_t stream);
hipError_t launch_i8_to_f32(const int8_t* in, float* out, size_t n, hipStream_t stream);
hipError_t launch_syn_layer(const float* in, const float* in2, const float* wt, const float* bias, float* out, int c_in,
                            int c_out, int k, int residual, int relu, int h, int w, hipStream_t stream);
bool syn_fused_supports(int c_in, int c, int halo);
void syn_fused_tiles(int h, int w, int halo, int* tiles_x, int* tiles_y);
hipError_t launch_syn_fused(const SynthFused* d_frames, int n_frames, int c_in, int c, int max_tiles_x, int max_tiles_y,
                            hipStream_t stream);
bool fused_dec_supports(int c_in, int c);
size_t fused_dec_lds_bytes(int n_lv, int c, int n_conv, int n_params, int pre);
void fused_dec_param_shape(int c_in, int c, int* nwv, int* nws, int* nwc, int* nwo);
hipError_t launch_fused_dec(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, int pre, size_t lds_bytes, hipStream_t stream);
hipError_t launch_fused_pyramid(const FusedDec* d_frames, const void* d_work, int n_work, int levels, size_t lds_bytes, hipStream_t stream);
size_t fused_pyr_lds_bytes(int n_lv);
bool fused_dec_cr_supports(int c_in, int c);
hipError_t launch_fused_dec_cr(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, size_t lds_bytes, hipStream_t stream);
int fused_dec_profile(unsigned long long* out16, int reset);
hipError_t launch_resize_nearest(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out,
                                 hipStream_t stream);
hipError_t launch_resize_interp(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out, int cubic,
                                float scale_y, float scale_x, hipStream_t stream);
hipError_t launch_final_resize(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out, int mode,
                               hipStream_t stream);
hipError_t launch_cr_noise(float* out, size_t n, hipStream_t stream);
hipError_t launch_planes_to_444(const void* p0, const void* p1, const void* p2, float* out, int h, int w, int bitdepth,
                                int frame_data_type, hipStream_t stream);
hipError_t launch_inter_recon(int frame_type, int h, int w, int n_taps, const int* gflow, const float* residue, const float* motion,
                              const float* ref0, const float* ref1, float* out, hipStream_t stream);
hipError_t launch_planes(const float* src, void* p0, void* p1, void* p2, int h, int w, int bitdepth, int frame_data_type,
                         hipStream_t stream);
hipError_t launch_widen_u8(const uint8_t* in, uint16_t* out, size_t n, hipStream_t stream);
hipError_t launch_spin(unsigned long long ticks, hipStream_t stream);
size_t inter_coef_bytes(int frame_type, int h, int w);
hipError_t launch_inter_coef8(int frame_type, int h, int w, const float* motion, void* coef, hipStream_t stream);
hipError_t launch_inter_apply8(int frame_type, int h, int w, const int* gflow, const float* residue, const float* motion, const float* ref0,
                               const float* ref1, const void* coef, float* out, hipStream_t stream);

static const uint32_t kScaleBits[kNumScale] = {
#include "../../include/ccd_scale_table.inc"
};
}  // namespace ccd

using namespace ccd;

#define HIP_TRY(expr)                           \
    do {                                        \
        hipError_t e__ = (expr);                \
        if (e__ != hipSuccess) return CCD_ERR_HIP; \
    } while (0)

namespace {

// Caches of device blocks and pinned host blocks, one per device (free lists, byte counts and caps are all per device): a batch is created, filled, run and destroyed
// per image set, and hipMalloc / hipFree / hipHostMalloc of its arenas were a fifth of the time from bytes to planes
// (24 hipFree = 4.4 ms per Kodak set; 64 arenas of 50-100 MB per 1080p GOP).  Blocks are handed out in size classes
// (power of two up to 1 MB, then eighths of a power of two: <= 12.5 % slack) and come back on destroy; the cache is capped
// (CCD_POOL_MAX_MB, default 16384; CCD_PINNED_POOL_MAX_MB, default 2048: the cache is invisible to PyTorch's allocator, so it
// stays a few percent of the device) - beyond the cap a block is really freed.
// ccd_pool_trim() empties the caches.  The current device must be the block's device (callers hipSetDevice first).
class BlockPool {
public:
    enum Kind { kDevice = 0, kPinned = 1 };
    static size_t size_class(size_t bytes) {
        if (bytes <= 4096) return 4096;
        size_t p2 = 4096;
        while (p2 < bytes) p2 <<= 1;
        if (p2 <= (size_t{1} << 20)) return p2;
        const size_t step = p2 >> 4;  // eighths of the power of two below
        return (bytes + step - 1) / step * step;
    }
    void* acquire(int device, Kind kind, size_t bytes, size_t* got) {
        const size_t cls = size_class(bytes);
        *got = cls;
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto& fl = free_[key(device, kind)];
            auto it = fl.find(cls);
            if (it != fl.end()) {
                void* p = it->second;
                fl.erase(it);
                cached_[key(device, kind)] -= cls;
                return p;
            }
        }
        void* p = nullptr;
        const hipError_t e = kind == kDevice ? hipMalloc(&p, cls) : hipHostMalloc(&p, cls, hipHostMallocDefault);
        if (e != hipSuccess) {  // out of memory with blocks of other classes cached: give them back and retry once
            (void)hipGetLastError();
            trim(device);
            if ((kind == kDevice ? hipMalloc(&p, cls) : hipHostMalloc(&p, cls, hipHostMallocDefault)) != hipSuccess) return nullptr;
        }
        return p;
    }
    void release(int device, Kind kind, void* p, size_t cls) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lock(mu_);
            size_t& cached = cached_[key(device, kind)];  // the caps are per device (and kind), like the free lists and trim()
            if (cached + cls <= cap(kind)) {
                free_[key(device, kind)].emplace(cls, p);
                cached += cls;
                return;
            }
        }
        if (kind == kDevice) (void)hipFree(p); else (void)hipHostFree(p);
    }
    void trim(int device) {
        std::vector<std::pair<Kind, void*>> drop;
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (int k = 0; k < 2; ++k) {
                auto& fl = free_[key(device, static_cast<Kind>(k))];
                for (auto& e : fl) drop.emplace_back(static_cast<Kind>(k), e.second);
                cached_[key(device, static_cast<Kind>(k))] = 0;
                fl.clear();
            }
        }
        for (auto& d : drop) { if (d.first == kDevice) (void)hipFree(d.second); else (void)hipHostFree(d.second); }
    }
private:
    static int key(int device, Kind kind) { return device * 2 + kind; }
    static size_t cap(Kind kind) {
        static const size_t caps[2] = {env_mb("CCD_POOL_MAX_MB", 16384), env_mb("CCD_PINNED_POOL_MAX_MB", 2048)};
        return caps[kind];
    }
    static size_t env_mb(const char* name, size_t dflt) {
        const char* e = std::getenv(name);
        return (e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : dflt) << 20;
    }
    std::mutex mu_;
    std::map<int, std::multimap<size_t, void*>> free_;
    std::map<int, size_t> cached_;  // bytes in free_[key], same key
};
BlockPool& pool() { static BlockPool p; return p; }

// A block from the pool with its class size (what release() needs).
struct Block {
    void* p = nullptr;
    size_t cls = 0;
    int device = 0;
    BlockPool::Kind kind = BlockPool::kDevice;
    bool get(int dev, BlockPool::Kind k, size_t bytes) { drop(); device = dev; kind = k; p = pool().acquire(dev, k, bytes, &cls); return p != nullptr; }
    void drop() { if (p) pool().release(device, kind, p, cls); p = nullptr; cls = 0; }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Per device, for the life of the process: the two Laplace-scale tables and the stream uploads run on (so that parsing
// slot k + 1 on the host overlaps the copy of slot k, and nothing waits for the caller's stream).
struct DeviceShared {
    float* d_scale_table = nullptr;
    double* d_rcp_table = nullptr;
    hipStream_t up_stream = nullptr;
    // side streams for the entropy launches of one batch (one per kernel instantiation in use: MLP width x variant): a
    // stream is a serial chain on one CU, so launches that queue behind each other on ONE stream add their durations
    static constexpr int kSide = 8;
    hipStream_t side[kSide] = {};
    // r06: which of them REALLY run at once.  HIP multiplexes its streams onto a few hardware queues (four by default,
    // GPU_MAX_HW_QUEUES), and two launches on streams that share one run one after the other: tools/ubench/queues.hip on MI355X /
    // ROCm 7.2 - the null stream + side[1] 9.7 ms where the null stream + side[0] take 4.9, three "concurrent" launches 2 x one
    // (profiles/r06/queues.txt); r05's fork over (caller's stream, side[0], side[1], ...) was serial or not by luck - with a
    // batch of 256 streams in two launches 72 ms instead of 37.  So the side streams are MEASURED once per device (calibrate):
    // conc[0 .. n_conc) are mutually concurrent ones (one per hardware queue: at most four are looked for), and the launches of a
    // batch that need to overlap go to those only - never to the caller's stream, which idles at the join and may alias any of them.
    int n_conc = 0;
    int conc[kSide] = {};
    int n_cu = 256;   // multiProcessorCount, read once (hipGetDeviceProperties is not a call for the path of every batch)
};
int device_shared(int device, DeviceShared** out);
void calibrate_side_streams(DeviceShared& d);

// Bump allocator over one pooled device block: every slot's buffers live in a single arena.
class Arena {
public:
    size_t reserve(size_t bytes) { size_t off = total_; total_ += (bytes + 255) & ~size_t{255}; return off; }
    int commit(int device) {
        if (total_ == 0) total_ = 256;
        return blk_.get(device, BlockPool::kDevice, total_) ? CCD_OK : CCD_ERR_NOMEM;
    }
    template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(blk_.as<char>() + off); }
    void release() { blk_.drop(); }
    size_t total() const { return total_; }
private:
    Block blk_;
    size_t total_ = 0;
};

struct Slot {
    ccd_cc_header hdr;
    Network net;
    int bitdepth = 0, frame_data_type = 0;
    Arena arena;
    EntropyParams ep;
    std::vector<UpsampleLevel> levels;
    int dense_c = 0, dense_h = 0, dense_w = 0;
    float* d_dense = nullptr;
    // common randomness (coolchic.py:179-183): noise pyramid + two ping-pong stacks for fixed_upsampling
    bool cr = false;
    float* d_noise = nullptr;
    float* d_nstack[2] = {nullptr, nullptr};
    std::vector<size_t> noise_off;        // per latent level, finest first
    std::vector<int> lvl_h, lvl_w;        // latent level sizes, finest first
    // synthesis
    float* d_syn_params = nullptr;
    bool use_fused_syn = false;      // whole synthesis in one kernel (ccd_synth_fused.hip)
    SynthFused fused;
    bool use_fused_dec = false;      // upsampling + synthesis + integer samples in one kernel (ccd_fused.hip)
    bool float_finite = true;        // float_path_stays_finite(): the network cannot leave the finite float32 range
    bool fdec_pre = false;           // ... whose level-1 stack comes from the batch's pyramid launch (CCD_OPT_FUSED_DEC = 2)
    FusedDec fpyr;                   // ... descriptor of that launch: the same walk one level up (level 0 = this frame's level 1)
    FusedDec fdec;
    size_t fdec_lds = 0;
    std::vector<size_t> w_off, b_off;  // per main layer
    size_t stab_w = 0, stab_b = 0, out_w = 0, out_b = 0;
    float* d_tmp[2] = {nullptr, nullptr};
    float* d_stab = nullptr;
    float* d_syn_out = nullptr;  // [C][dense_h][dense_w]
    float* d_out = nullptr;      // [C][H][W] (== d_syn_out when no resize)
    void* d_plane[3] = {nullptr, nullptr, nullptr};
    int plane_h[3] = {0, 0, 0}, plane_w[3] = {0, 0, 0};
    size_t plane_off[3] = {0, 0, 0}, planes_bytes = 0;  // the three planes sit in ONE block of the arena (one copy moves them)
    Block staging;           // pinned host copy of the arena's head (status, payload, networks) for the asynchronous upload
    int32_t* d_status = nullptr;
    bool use_pipe = false;   // pipelined entropy kernel (32-bit operands) or the generic one
    bool use_mfma = false;   // ... with the ARM's layers on the matrix cores (limb-split int8)
    bool use_dyn = false;    // ... the instantiation that checks IFCE features on the device (worst case >= 2^15, or the test hook)
    int fixed_shape = 0;     // ... the instantiation with a compile-time ARM shape (1: intra/hop.cfg = 14 + 6 inputs, two hidden layers)
    int ring_rows = 64;      // rows of the pipelined kernel's decoded-symbol ring
    size_t lds_generic = 0, lds_pipe = 0;
    int lg = -1;             // entropy launch of the batch this slot is decoded by (index into ccd_batch::pipe_groups; -1: the generic launch)
    int fl = -1;             // launch group its float-path launches are keyed by (= lg while ccd_batch_run overlaps; -1: not keyed)
    int status = CCD_OK;
    int32_t host_status[64] = {0};
};

}  // namespace

struct ccd_batch {
    int device = 0;
    std::vector<std::unique_ptr<Slot>> slots;
    EntropyParams* d_params = nullptr;   // [pipe slots..., generic slots...]
    int n_params_uploaded = 0;
    bool regroup = false;                // an option that shapes the launch tables changed: rebuild them at the next run
    // every table the launches read (entropy descriptors, fused-kernel frames and work lists, pyramid steps) and the status
    // words of all slots live in ONE pooled device block, staged through ONE pinned block: one copy up, one copy down
    Block tables, tables_staging, status_host;
    int32_t* d_status_all = nullptr;     // [slots][64]
    hipEvent_t up_done = nullptr;        // recorded on the upload stream behind the last ccd_batch_add
    hipStream_t up_stream = nullptr;     // the device's shared upload stream
    // every stream the caller handed to ccd_batch_run_stage / ccd_batch_wait / ccd_batch_copy_*: all of them are drained before a
    // block of this batch goes back to the pool (or its tables are replaced), not only the last one
    std::vector<hipStream_t> streams_used;
    void note_stream(hipStream_t st) { if (std::find(streams_used.begin(), streams_used.end(), st) == streams_used.end()) streams_used.push_back(st); }
    int drain_streams() {
        int rc = CCD_OK;
        for (hipStream_t st : streams_used) if (hipStreamSynchronize(st) != hipSuccess) rc = CCD_ERR_HIP;
        // the device's SHARED side streams are not drained (another batch in flight may be launching on them: draining would
        // make this batch's destroy wait for that batch's entropy chains) - this batch's own work on them ends at its events
        for (int k = 0; k < DeviceShared::kSide; ++k)
            if (side_pending[k] && side_done[k]) { if (hipEventSynchronize(side_done[k]) != hipSuccess) rc = CCD_ERR_HIP; side_pending[k] = false; }
        return rc;
    }
    // fork / join of the entropy launches over the device's side streams: the EVENTS belong to the batch (two host threads
    // running two batches on one GPU share the side streams, which only serialises their launches, but never an event)
    hipEvent_t fork = nullptr;
    hipEvent_t side_done[DeviceShared::kSide] = {};   // recorded behind EVERY launch of this batch on side stream k
    bool side_pending[DeviceShared::kSide] = {};      // ... and not yet known to have completed
    // the launch tables' copy (upload_params): a run on ANOTHER stream than the one that carried it waits for this event
    hipEvent_t params_up = nullptr;
    hipStream_t params_stream = nullptr;
    bool uploads_unconfirmed = false;    // slots were added since the last ccd_batch_wait: launches order themselves behind up_done
    int n_pipe = 0, n_generic = 0;
    // cg: chain group - the slots of one kernel instantiation split by the expected length of their serial chains, so that the
    // float-path launches of the streams that finish early run while the longest chains are still decoding (ccd_batch_run)
    struct PipeGroup { int nv, mfma, dyn, shape, cg, first, n; size_t lds; double est; };
    // CCD_OPT_TIME_LAUNCHES: timing events around every entropy launch, on the stream it runs on (bench.py's roofline: the launches
    // of a step overlap on side streams, so events on the caller's stream only see the whole stage)
    int opt_time_launches = 0;
    std::vector<hipEvent_t> lt0, lt1;    // per entropy launch of the last run (launch order)
    // behind the float launches of pipe group gi in the last OVERLAPPED run (ccd_decode_video: a frame's flows are ready when the
    // launch of its motion cool-chic is, long before the whole batch); lg_valid: recorded in the last run
    std::vector<hipEvent_t> lg_done;
    bool lg_valid = false;
    int n_timed = 0;
    int opt_overlap = 1;                 // CCD_OVERLAP=0 (environment; A/B and tests): one entropy launch per instantiation, float stages behind the join
    std::vector<PipeGroup> pipe_groups;
    float* d_scale_table = nullptr;
    double* d_rcp_table = nullptr;
    size_t lds_generic = 0, lds_pipe = 0;
    int force_generic = 0;               // CCD_FORCE_GENERIC=1: tests exercise the fallback kernels
    // fused-synthesis launches: frames grouped by (padded input channels, output channels)
    struct FusedGroup { int cp, c, c_in, n, first, max_tx, max_ty; };
    std::vector<FusedGroup> fused_groups;
    SynthFused* d_fused = nullptr;
    // fused float path (ccd_fused.hip): frames grouped by (latent levels, output channels); one workgroup per run of tiles
    struct FdecGroup { int c_in, c, pre, cr, fl, first_frame, first_work, n_work; size_t lds; };
    std::vector<FdecGroup> fdec_groups;
    FusedDec* d_fdec = nullptr;
    void* d_fdec_work = nullptr;
    // pyramid launches in front of the kFdPre groups (stage 1): descriptors grouped by their number of levels
    struct PyrGroup { int levels, fl, first_frame, first_work, n_work; size_t lds; };
    std::vector<PyrGroup> pyr_groups;
    FusedDec* d_pyr = nullptr;
    void* d_pyr_work = nullptr;
    int opt_fused_dec = 2;               // CCD_OPT_FUSED_DEC: 2 = fused kernel behind the pyramid launch, 1 = the whole pyramid per tile, 0 = unfused
    int opt_keep_float = 1;              // CCD_OPT_KEEP_FLOAT
    int opt_range_bits = 0;              // CCD_OPT_RANGE_BITS (tests: lowered feature limit of the dynamic operand check)
    int opt_mfma_arm = 0;                // CCD_OPT_MFMA_ARM (off: bit-exact but slower than the vector-ALU producers)
    int opt_fixed_shape = 1;             // CCD_FIXED_SHAPE=0 (environment; A/B and tests): every network through the run-time-shape instantiations
    // upsampling: step k of every slot's pyramid in one launch
    struct UpsStep { int first_z, n_z, max_w, max_h; };
    std::vector<UpsStep> ups_steps;
    UpsampleLevel* d_levels = nullptr;
    uint32_t* d_zmap = nullptr;
};

extern "C" {

const char* ccd_strerror(int code) {
    switch (code) {
        case CCD_OK: return "ok";
        case CCD_ERR_TRUNCATED: return "bitstream truncated";
        case CCD_ERR_VALUE: return "header value out of range";
        case CCD_ERR_INVALID_DATA: return "invalid compressed data";
        case CCD_ERR_UNSUPPORTED: return "feature not supported by this build";
        case CCD_ERR_NOMEM: return "out of memory";
        case CCD_ERR_HIP: return "HIP runtime error / no usable gfx950 device";
        case CCD_ERR_ARG: return "bad argument";
        default: return "unknown error";
    }
}

const char* ccd_version(void) { return "ccd 0.1.0 gfx950"; }

int ccd_read_video_header(const uint8_t* p, size_t n, ccd_video_header* h) { return (p && h) ? read_video_header(p, n, h) : CCD_ERR_ARG; }
int ccd_read_frame_header(const uint8_t* p, size_t n, ccd_frame_header* h) { return (p && h) ? read_frame_header(p, n, h) : CCD_ERR_ARG; }
int ccd_read_cc_header(const uint8_t* p, size_t n, ccd_cc_header* h) { return (p && h) ? read_cc_header(p, n, h) : CCD_ERR_ARG; }

int ccd_get_coding_structure(const ccd_video_header* h, int32_t* display_order, int32_t* frame_type, int32_t* refs, int32_t* depth) {
    if (!h) return CCD_ERR_ARG;
    std::vector<CodedFrame> cs;
    const int rc = coding_structure(*h, cs);
    if (rc < 0) return rc;
    for (size_t i = 0; i < cs.size(); ++i) {
        if (display_order) display_order[i] = cs[i].display_order;
        if (frame_type) frame_type[i] = cs[i].frame_type;
        if (refs) { refs[2 * i] = cs[i].n_refs > 0 ? cs[i].refs[0] : -1; refs[2 * i + 1] = cs[i].n_refs > 1 ? cs[i].refs[1] : -1; }
        if (depth) depth[i] = cs[i].depth;
    }
    return static_cast<int>(cs.size());
}

void ccd_free(void* p) { std::free(p); }

// -------------------------------------------------------------------------------------------------
// Batch
// -------------------------------------------------------------------------------------------------
namespace {
std::mutex g_shared_mu;
std::map<int, DeviceShared> g_shared;

// Which side streams run concurrently (see DeviceShared::conc).  Two spin kernels of ~0.2 ms, one on each of two streams, take
// ~0.2 ms when the streams sit on different hardware queues and ~0.4 ms when they share one; a greedy clique of up to four.  ~10 ms
// once per device and process.  CCD_SIDE_STREAMS=k skips the measurement and takes the first k (0 < k <= 8; r05's behaviour: 8).
// Under a profiler that serialises kernels nothing is concurrent: one stream, launches in a row - correct, only slower.
void calibrate_side_streams(DeviceShared& d) {
    d.n_conc = 1; d.conc[0] = 0;
    if (const char* e = std::getenv("CCD_SIDE_STREAMS")) {
        const int k = std::atoi(e);
        if (k >= 1 && k <= DeviceShared::kSide) { d.n_conc = k; for (int i = 0; i < k; ++i) d.conc[i] = i; return; }
    }
    const unsigned long long ticks = 400000ull;  // ~0.17 ms of the shader clock
    auto pair_ms = [&](int a, int b2) -> double {
        double best = 1e9;
        for (int trial = 0; trial < 3; ++trial) {
            if (hipStreamSynchronize(d.side[a]) != hipSuccess || hipStreamSynchronize(d.side[b2]) != hipSuccess) return 1e9;
            const auto t0 = std::chrono::steady_clock::now();
            if (launch_spin(ticks, d.side[a]) != hipSuccess || launch_spin(ticks, d.side[b2]) != hipSuccess) return 1e9;
            if (hipStreamSynchronize(d.side[a]) != hipSuccess || hipStreamSynchronize(d.side[b2]) != hipSuccess) return 1e9;
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        return best;
    };
    // one kernel alone (same stream twice = certainly serial): the yardstick, whatever the clock
    const double serial = pair_ms(0, 0);
    if (serial >= 1e8) return;
    for (int j = 1; j < DeviceShared::kSide && d.n_conc < 4; ++j) {
        bool with_all = true;
        for (int i = 0; i < d.n_conc && with_all; ++i) with_all = pair_ms(d.conc[i], j) < 0.75 * serial;
        if (with_all) d.conc[d.n_conc++] = j;
    }
    if (std::getenv("CCD_VIDEO_TIMING") || std::getenv("CCD_DEBUG_STREAMS")) {
        std::fprintf(stderr, "[ccd] side streams that run concurrently: %d (", d.n_conc);
        for (int i = 0; i < d.n_conc; ++i) std::fprintf(stderr, "%s%d", i ? " " : "", d.conc[i]);
        std::fprintf(stderr, "), two kernels in a row %.3f ms\n", serial);
    }
}

int device_shared(int device, DeviceShared** out) {
    std::lock_guard<std::mutex> lock(g_shared_mu);
    DeviceShared& d = g_shared[device];
    if (!d.up_stream) {
        float* st = nullptr;
        double* rt = nullptr;
        hipStream_t us = nullptr;
        std::vector<double> rcp(kNumScale);
        for (int i = 0; i < kNumScale; ++i) {
            float f;
            std::memcpy(&f, &kScaleBits[i], 4);
            rcp[i] = 1.0 / static_cast<double>(f);  // IEEE division on the host: correctly rounded
        }
        if (hipMalloc(&st, sizeof(kScaleBits)) != hipSuccess || hipMalloc(&rt, sizeof(double) * kNumScale) != hipSuccess) {
            if (st) (void)hipFree(st);
            return CCD_ERR_NOMEM;
        }
        if (hipMemcpy(st, kScaleBits, sizeof(kScaleBits), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(rt, rcp.data(), sizeof(double) * kNumScale, hipMemcpyHostToDevice) != hipSuccess ||
            hipStreamCreateWithFlags(&us, hipStreamNonBlocking) != hipSuccess) {
            (void)hipFree(st); (void)hipFree(rt);
            return CCD_ERR_HIP;
        }
        d.d_scale_table = st; d.d_rcp_table = rt; d.up_stream = us;
        bool ok = true;
        for (int k = 0; k < DeviceShared::kSide && ok; ++k) ok = hipStreamCreateWithFlags(&d.side[k], hipStreamNonBlocking) == hipSuccess;
        if (!ok) return CCD_ERR_HIP;
        calibrate_side_streams(d);
        { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) d.n_cu = v; }
    }
    *out = &d;
    return CCD_OK;
}
}  // namespace

int ccd_batch_create(int device, ccd_batch** out) {
    if (!out) return CCD_ERR_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return CCD_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    DeviceShared* sh = nullptr;
    const int rc = device_shared(device, &sh);
    if (rc < 0) return rc;
    ccd_batch* b = new (std::nothrow) ccd_batch();
    if (!b) return CCD_ERR_NOMEM;
    b->device = device;
    if (const char* e = std::getenv("CCD_FORCE_GENERIC")) b->force_generic = std::atoi(e);
    if (const char* e = std::getenv("CCD_FUSED_DEC")) {  // 0 / 1 / 2 like the option; anything else leaves the default
        const int v = std::atoi(e);
        if (v >= 0 && v <= 2) b->opt_fused_dec = v;
    }
    if (const char* e = std::getenv("CCD_MFMA_ARM")) b->opt_mfma_arm = std::atoi(e);
    if (const char* e = std::getenv("CCD_FIXED_SHAPE")) b->opt_fixed_shape = std::atoi(e);
    if (const char* e = std::getenv("CCD_OVERLAP")) b->opt_overlap = std::atoi(e);
    b->d_scale_table = sh->d_scale_table;
    b->d_rcp_table = sh->d_rcp_table;
    b->up_stream = sh->up_stream;
    if (hipEventCreateWithFlags(&b->up_done, hipEventDisableTiming) != hipSuccess) { delete b; return CCD_ERR_HIP; }
    *out = b;
    return CCD_OK;
}

void ccd_batch_destroy(ccd_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    // blocks go back to the pool for the next batch: nothing of this one may still be in flight
    if (b->up_done) { (void)hipEventSynchronize(b->up_done); (void)hipEventDestroy(b->up_done); }
    (void)b->drain_streams();  // launches and copies on EVERY stream the caller used with this batch
    if (b->fork) (void)hipEventDestroy(b->fork);
    if (b->params_up) (void)hipEventDestroy(b->params_up);
    for (hipEvent_t e : b->lg_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->lt0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->lt1) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->side_done) if (e) (void)hipEventDestroy(e);
    for (auto& s : b->slots) { s->arena.release(); s->staging.drop(); }
    b->tables.drop(); b->tables_staging.drop(); b->status_host.drop();
    delete b;
}

void ccd_pool_trim(int device) {
    if (hipSetDevice(device) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    pool().trim(device);
}

int ccd_batch_size(const ccd_batch* b) { return b ? static_cast<int>(b->slots.size()) : CCD_ERR_ARG; }

int ccd_batch_header(const ccd_batch* b, int slot, ccd_cc_header* h) {
    if (!b || !h || slot < 0 || slot >= static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    *h = b->slots[slot]->hdr;
    return CCD_OK;
}

int ccd_batch_add(ccd_batch* b, const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                  const uint8_t* bytes_latent, size_t n_lat, int bitdepth, int frame_data_type) {
    if (!b || !cc_header || !bytes_nn || (!bytes_latent && n_lat)) return CCD_ERR_ARG;
    if (bitdepth != 0 && (bitdepth < 8 || bitdepth > 16)) return CCD_ERR_ARG;
    HIP_TRY(hipSetDevice(b->device));
    std::unique_ptr<Slot> sp(new (std::nothrow) Slot());
    if (!sp) return CCD_ERR_NOMEM;
    Slot& s = *sp;
    int rc = read_cc_header(cc_header, n_hdr, &s.hdr);
    if (rc < 0) return rc;
    const ccd_cc_header& h = s.hdr;
    if (n_lat % 4) return CCD_ERR_VALUE;  // np.frombuffer(dtype=uint32) raises (rangecoder.py:81)
    // The header parses, but the reference cannot decode it: after ONE x2 nearest upsample + crop of the decoded stack its
    // torch.cat raises when consecutive grids differ by more than one level (latent and hyperlatent ranges that do not
    // touch).  The entropy kernels index the coarser grid with (y >> 1, x >> 1): a larger gap would read past it.
    // (The transmitted n_latent_grids is not checked: the reference recomputes the count from the resolutions and never
    // reads the field, component/core/coolchic.py:170-185, header.py:354-377.)
    for (int g = 1; g < h.n_grids; ++g) {
        const bool same = h.grid_h[g] == h.grid_h[g - 1] && h.grid_w[g] == h.grid_w[g - 1];
        const bool half = h.grid_h[g] == (h.grid_h[g - 1] + 1) / 2 && h.grid_w[g] == (h.grid_w[g - 1] + 1) / 2;
        if (!same && !half) return CCD_ERR_VALUE;
    }
    rc = decode_network(h, bytes_nn, n_nn, s.net);
    if (rc < 0) return rc;
    s.bitdepth = bitdepth; s.frame_data_type = frame_data_type;
    const Network& net = s.net;

    // ---- pack the integer networks ------------------------------------------------------------------
    std::vector<int64_t> arm_blob;
    for (const FixedLayer& L : net.arm.layers) {
        arm_blob.insert(arm_blob.end(), L.w.begin(), L.w.end());
        arm_blob.insert(arm_blob.end(), L.b.begin(), L.b.end());
    }
    arm_blob.insert(arm_blob.end(), net.arm.ws.begin(), net.arm.ws.end());
    arm_blob.insert(arm_blob.end(), net.arm.bs.begin(), net.arm.bs.end());
    std::vector<int64_t> ifce_blob;
    std::vector<int32_t> ifce_off(h.n_grids, 0);
    for (int g = 0; g < h.n_grids; ++g) {
        if (net.ifce[g].dim == 0) continue;
        ifce_off[g] = static_cast<int32_t>(ifce_blob.size());
        const FixedLayer& L = net.ifce[g].layers[0];
        ifce_blob.insert(ifce_blob.end(), L.w.begin(), L.w.end());
        ifce_blob.insert(ifce_blob.end(), L.b.begin(), L.b.end());
    }
    int max_w = 0;
    for (int g = 0; g < h.n_grids; ++g) max_w = std::max(max_w, static_cast<int>(h.grid_w[g]));
    s.use_pipe = !b->force_generic && entropy_pipe_supports(h.total_context_arm, h.n_hidden_layers_arm + 1, (net.arm.w32 && net.feat_i32 && !net.arm.dyn_act) ? 1 : 0, max_w);
    {
        long long max_w_abs = 0;
        for (const FixedLayer& L : net.arm.layers) for (int64_t w : L.w) max_w_abs = std::max<long long>(max_w_abs, w < 0 ? -w : w);
        for (int64_t w : net.arm.ws) max_w_abs = std::max<long long>(max_w_abs, w < 0 ? -w : w);
        s.use_mfma = s.use_pipe && b->opt_mfma_arm &&
                     entropy_pipe_supports_mfma(h.total_context_arm, h.n_hidden_layers_arm + 1, h.output_feature_ifce, net.arm.narrow ? 1 : 0, max_w, max_w_abs);
    }
    s.use_dyn = s.use_pipe && !s.use_mfma && (net.arm.dyn_feat || b->opt_range_bits != 0);
    s.fixed_shape = (s.use_pipe && !s.use_mfma && b->opt_fixed_shape) ? entropy_pipe_fixed_shape(h.total_context_arm, h.n_hidden_layers_arm + 1, h.spatial_context_arm) : 0;
    s.ring_rows = s.use_mfma ? std::max(entropy_pipe_ring_rows(max_w), 64) : 512;  // the matrix-core variant needs the LDS for its operand tables
    s.lds_pipe = entropy_pipe_lds_bytes(h.total_context_arm, h.n_hidden_layers_arm + 1, s.ring_rows, s.use_mfma ? 1 : 0);
    s.lds_generic = entropy_lds_bytes(h.total_context_arm, static_cast<int>(arm_blob.size()));
    if (!s.use_pipe && s.lds_generic > 160 * 1024) return CCD_ERR_UNSUPPORTED;  // ARM too large for the LDS-resident kernels

    // ---- geometry of the float stages ------------------------------------------------------------------
    std::vector<int> lat_grids;
    for (int g = 0; g < h.n_grids; ++g) if (!h.is_hyperlatent[g]) lat_grids.push_back(g);
    const int n_levels = static_cast<int>(lat_grids.size());
    s.cr = h.flag_common_randomness != 0;
    if (n_levels < 1 || n_levels * (s.cr ? 2 : 1) != h.input_feature_synthesis) return CCD_ERR_VALUE;
    if (n_levels > 1 && net.n_ups < 1) return CCD_ERR_VALUE;
    s.dense_c = h.input_feature_synthesis; s.dense_h = h.grid_h[lat_grids[0]]; s.dense_w = h.grid_w[lat_grids[0]];
    const int H = h.img_size[0], W = h.img_size[1];
    const bool need_resize = (s.dense_h != H || s.dense_w != W);
    // the reference concatenates [.., dense_h, dense_w] with noise resized to img_size: torch.cat raises unless equal
    if (s.cr && need_resize) return CCD_ERR_VALUE;
    if (bitdepth != 0 && h.out_channels < 3) return CCD_ERR_ARG;

    // ---- arena layout --------------------------------------------------------------------------------
    // The head of the arena - status block, payload words, integer networks, synthesis parameters - is what the host
    // uploads: contiguous, staged in one pinned block, moved by ONE asynchronous copy (zeros included: the status words and
    // the two payload words the decoder may read past the end).  Everything behind it is written by kernels before it is read.
    Arena& A = s.arena;
    const size_t n_words = n_lat / 4;
    const size_t o_status = A.reserve(512);
    const size_t o_words = A.reserve((n_words + 2) * 4);
    const size_t o_arm = A.reserve(arm_blob.size() * 8);
    const size_t o_ifce = A.reserve(std::max<size_t>(ifce_blob.size(), 1) * 8);
    size_t feat_px = 1;
    for (int g = 0; g < h.n_grids; ++g)
        if (h.input_features_ifce[g] > 0) {
            const int fg = (g == h.n_grids - 1) ? g : g + 1;
            feat_px = std::max(feat_px, static_cast<size_t>(h.grid_h[fg]) * h.grid_w[fg]);
        }
    const size_t dense_elems = static_cast<size_t>(s.dense_c) * s.dense_h * s.dense_w;
    size_t n_noise = 0, nstack_elems[2] = {1, 1};
    if (s.cr) {
        for (int i = 0; i < n_levels; ++i) {
            s.lvl_h.push_back(h.grid_h[lat_grids[i]]); s.lvl_w.push_back(h.grid_w[lat_grids[i]]);
            s.noise_off.push_back(n_noise);
            n_noise += static_cast<size_t>(s.lvl_h[i]) * s.lvl_w[i];
        }
        s.noise_off.push_back(n_noise);
        // intermediate stacks: level 1 holds n_levels-1 planes, level 2 n_levels-2 planes (the finest goes to dense)
        for (int k = 0; k < 2; ++k) {
            const int lv = k + 1;
            nstack_elems[k] = lv < n_levels ? static_cast<size_t>(n_levels - lv) * s.lvl_h[lv] * s.lvl_w[lv] : 1;
        }
    }
    size_t stack_b_elems = 1;
    if (n_levels >= 3) {
        const int g1 = lat_grids[1];
        stack_b_elems = static_cast<size_t>(n_levels - 1) * h.grid_h[g1] * h.grid_w[g1];
    }
    // synthesis parameters blob
    std::vector<float> syn_blob;
    auto push = [&](const std::vector<float>& v) { size_t off = syn_blob.size(); syn_blob.insert(syn_blob.end(), v.begin(), v.end()); return off; };
    s.w_off.clear(); s.b_off.clear();
    int max_c = h.out_channels;
    for (const SynLayerParams& L : net.syn) { s.w_off.push_back(push(L.w)); s.b_off.push_back(push(L.b)); max_c = std::max(max_c, L.c_out); }
    if (net.syn_stab.c_out) { s.stab_w = push(net.syn_stab.w); s.stab_b = push(net.syn_stab.b); }
    s.out_w = push(net.syn_out.w); s.out_b = push(net.syn_out.b);
    // ---- fused-synthesis layout (zero-padded copies of the first two layers and the stabiliser) -------------
    {
        SynthFused& F = s.fused;
        std::memset(&F, 0, sizeof(F));
        const auto& L = net.syn;
        bool ok = L.size() >= 2 && L.size() <= 5 && L[0].k == 1 && L[1].k == 1 && !L[0].residual && !L[1].residual &&
                  L[1].c_out == h.out_channels && !b->force_generic;
        int halo = 0;
        for (size_t l = 2; ok && l < L.size(); ++l) {
            ok = L[l].c_in == h.out_channels && L[l].c_out == h.out_channels && (L[l].k & 1) && L[l].k <= 7;
            halo += (L[l].k - 1) / 2;
        }
        ok = ok && syn_fused_supports(s.dense_c, h.out_channels, halo) && (!net.syn_stab.c_out || net.syn_stab.c_in <= s.dense_c);
        if (ok) {
            const int cp = ((s.dense_c + 3) / 4) * 4, C = h.out_channels, N = L[0].c_out;
            auto push_padded = [&](const std::vector<float>& w, int rows, int cols) {
                const size_t off = syn_blob.size();
                for (int r = 0; r < rows; ++r)
                    for (int c = 0; c < cp; ++c) syn_blob.push_back(c < cols ? w[static_cast<size_t>(r) * cols + c] : 0.0f);
                return static_cast<int32_t>(off);
            };
            F.c_in = s.dense_c; F.c = C; F.n_hidden = N; F.relu0 = L[0].relu; F.relu1 = L[1].relu;
            F.w0_off = push_padded(L[0].w, N, s.dense_c); F.b0_off = static_cast<int32_t>(s.b_off[0]);
            F.w1_off = static_cast<int32_t>(s.w_off[1]); F.b1_off = static_cast<int32_t>(s.b_off[1]);
            F.n_conv = static_cast<int32_t>(L.size()) - 2;
            for (int l = 0; l < F.n_conv; ++l) {
                F.conv_k[l] = L[l + 2].k; F.conv_residual[l] = L[l + 2].residual; F.conv_relu[l] = L[l + 2].relu;
                F.conv_w_off[l] = static_cast<int32_t>(s.w_off[l + 2]); F.conv_b_off[l] = static_cast<int32_t>(s.b_off[l + 2]);
            }
            F.has_stab = net.syn_stab.c_out ? 1 : 0;
            if (F.has_stab) {
                F.stab_c_in = net.syn_stab.c_in;
                F.stab_w_off = push_padded(net.syn_stab.w, C, net.syn_stab.c_in);
                F.stab_b_off = static_cast<int32_t>(s.stab_b);
            }
            F.out_w_off = static_cast<int32_t>(s.out_w); F.out_b_off = static_cast<int32_t>(s.out_b);
            F.halo = halo;
            s.use_fused_syn = true;
        }
    }
    // ---- fused float path (ccd_fused.hip): parameters in MFMA order, appended to the blob ------------------------------
    {
        FusedDec& D = s.fdec;
        std::memset(&D, 0, sizeof(D));
        const auto& L = net.syn;
        const int C = h.out_channels;
        // common randomness: n_levels noise planes behind the latent channels (the kFdPre instantiations with NZ = CIN; pictures only)
        const int NZ = s.cr ? n_levels : 0;
        // the matrix-core kernel is exact for FINITE values (its zero-weight padding: fma(v, 0, acc) == acc); a network that
        // could overflow float32 for some latents runs the vector-ALU kernels, which evaluate the oracle's taps only
        s.float_finite = float_path_stays_finite(net, n_levels, NZ);
        bool ok = b->opt_fused_dec && !b->force_generic && s.float_finite && n_levels + NZ == s.dense_c && n_levels >= 2 && n_levels <= kFdMaxLevels &&
                  fused_dec_supports(n_levels, C) && (!s.cr || (b->opt_fused_dec == 2 && fused_dec_cr_supports(n_levels, C))) &&
                  net.ups_k == 8 && net.pre_k == 7 && L.size() >= 2 &&
                  L.size() <= 2 + static_cast<size_t>(kFdMaxConv) && L[0].k == 1 && L[1].k == 1 && !L[0].residual && !L[1].residual &&
                  L[0].c_in == n_levels + NZ && L[1].c_in == L[0].c_out && L[1].c_out == C && (!net.syn_stab.c_out || net.syn_stab.c_in <= n_levels);
        for (size_t l = 2; ok && l < L.size(); ++l) ok = L[l].k == 3 && L[l].c_in == C && L[l].c_out == C;
        if (ok) {
            const int CIN = n_levels, N = L[0].c_out, CT = (C + 3) / 4, NT = (N + 3) / 4;
            const int CI = CIN + NZ;  // inputs of the first 1x1 layer
            int nwv, nws, nwc, nwo;
            fused_dec_param_shape(CIN, C, &nwv, &nws, &nwc, &nwo);
            nwv = (CI + 4 * CT + 15) / 16;
            while (syn_blob.size() % 4) syn_blob.push_back(0.0f);  // the kernel copies the block with 16-byte loads
            const size_t base = syn_blob.size();
            auto alloc = [&](size_t n) { const size_t off = syn_blob.size() - base; syn_blob.resize(syn_blob.size() + n, 0.0f); return static_cast<int32_t>(off); };
            float* P = nullptr;
            auto quad = [&](int32_t off, int q, int i) -> float& { return P[off + q * 4 + i]; };
            D.n_tiles_hidden = NT;
            D.wq_off = alloc(static_cast<size_t>(NT) * nwv * 64); D.b0_off = alloc(static_cast<size_t>(NT) * 4);
            D.b1_off = alloc(static_cast<size_t>(CT) * 4);
            D.stab_off = alloc(static_cast<size_t>(nws) * 64); D.stabb_off = alloc(static_cast<size_t>(CT) * 4);
            D.n_conv = static_cast<int32_t>(L.size()) - 2;
            for (int l = 0; l < D.n_conv; ++l) { D.conv_off[l] = alloc(static_cast<size_t>(nwc) * 64); D.convb_off[l] = alloc(static_cast<size_t>(CT) * 4); }
            D.out_off = alloc(static_cast<size_t>(nwo) * 64); D.outb_off = alloc(static_cast<size_t>(CT) * 4);
            D.n_params = static_cast<int32_t>(syn_blob.size() - base);
            P = syn_blob.data() + base;
            for (int n = 0; n < NT; ++n) {
                const int32_t wq = D.wq_off + n * nwv * 64;
                for (int i = 0; i < 4; ++i) {
                    const int hu = 4 * n + i;  // hidden unit = row i of the tile
                    if (hu >= N) continue;
                    for (int c = 0; c < CI; ++c) quad(wq, c, i) = L[0].w[static_cast<size_t>(hu) * CI + c];
                    P[D.b0_off + 4 * n + i] = L[0].b[hu];
                }
                for (int t = 0; t < CT; ++t)
                    for (int r = 0; r < 4; ++r)
                        for (int i = 0; i < 4; ++i) {
                            const int oc = 4 * t + i, hu = 4 * n + r;
                            if (oc < C && hu < N) quad(wq, CI + t * 4 + r, i) = L[1].w[static_cast<size_t>(oc) * N + hu];
                        }
            }
            for (int oc = 0; oc < C; ++oc) P[D.b1_off + oc] = L[1].b[oc];
            D.has_stab = net.syn_stab.c_out ? 1 : 0;
            if (D.has_stab)
                for (int oc = 0; oc < C; ++oc) {
                    for (int c = 0; c < net.syn_stab.c_in; ++c) quad(D.stab_off, c * CT + oc / 4, oc % 4) = net.syn_stab.w[static_cast<size_t>(oc) * net.syn_stab.c_in + c];
                    P[D.stabb_off + oc] = net.syn_stab.b[oc];
                }
            for (int l = 0; l < D.n_conv; ++l) {
                const SynLayerParams& Lc = L[l + 2];
                D.conv_residual[l] = Lc.residual; D.conv_relu[l] = Lc.relu;
                for (int oc = 0; oc < C; ++oc) {
                    for (int k = 0; k < 9 * C; ++k) quad(D.conv_off[l], k * CT + oc / 4, oc % 4) = Lc.w[static_cast<size_t>(oc) * 9 * C + k];
                    P[D.convb_off[l] + oc] = Lc.b[oc];
                }
            }
            for (int oc = 0; oc < C; ++oc) {
                for (int i = 0; i < C; ++i) quad(D.out_off, i * CT + oc / 4, oc % 4) = net.syn_out.w[static_cast<size_t>(oc) * C + i];
                P[D.outb_off + oc] = net.syn_out.b[oc];
            }
            D.relu0 = L[0].relu; D.relu1 = L[1].relu;
            D.n_lv = n_levels; D.c = C; D.h = s.dense_h; D.w = s.dense_w;
            D.margin = D.n_conv == 0 ? 0 : (D.n_conv <= 2 ? 2 : 4);
            D.tiles_x = (D.w + (64 - 2 * D.margin) - 1) / (64 - 2 * D.margin);
            D.tiles_y = (D.h + (32 - 2 * D.margin) - 1) / (32 - 2 * D.margin);
            // kron products of the symmetric 1-D filters (upsampling.py:42-64, 189-196, 312-325): level i is produced by step
            // n_levels - 2 - i (coarsest first), whose filters are those of index step % n_ups
            for (int i = 0; i + 1 < n_levels; ++i) {
                const int kidx = (n_levels - 2 - i) % net.n_ups;
                const float* uw = &net.ups_w[static_cast<size_t>(kidx) * net.ups_k];
                const float* pw = &net.pre_w[static_cast<size_t>(kidx) * net.pre_k];
                for (int a2 = 0; a2 < 4; ++a2)
                    for (int b2 = a2; b2 < 4; ++b2) {
                        volatile float pu = uw[a2] * uw[b2], pp = pw[a2] * pw[b2];  // rounded to f32 like the kernels' own products
                        D.k2u[i][k2_index(a2, b2)] = pu; D.k2p[i][k2_index(a2, b2)] = pp;
                    }
            }
            s.fdec_lds = fused_dec_lds_bytes(n_levels, C, D.n_conv, D.n_params, (b->opt_fused_dec == 2 && n_levels >= 5) ? 1 : 0);
            s.use_fused_dec = s.fdec_lds <= 160 * 1024;
            if (!s.use_fused_dec) syn_blob.resize(base);
            else D.params = reinterpret_cast<const float*>(base);  // offset for now; becomes a pointer once the arena exists
        }
    }
    const size_t o_synp = A.reserve(syn_blob.size() * 4);
    const size_t head_bytes = A.total();
    std::vector<size_t> o_lat(h.n_grids);
    for (int g = 0; g < h.n_grids; ++g) o_lat[g] = A.reserve(static_cast<size_t>(h.grid_h[g]) * h.grid_w[g]);
    // int32 planes (generic kernel) or int16 planes + int32 side planes in the second half (pipelined kernel)
    const size_t feat_elems = feat_px * std::max(h.output_feature_ifce, 1);
    const size_t o_feat = A.reserve(feat_elems * 8);
    size_t o_noise = 0, o_nstack[2] = {0, 0};
    if (s.cr) {
        o_noise = A.reserve(n_noise * 4);
        for (int k = 0; k < 2; ++k) o_nstack[k] = A.reserve(nstack_elems[k] * 4);
    }
    const size_t plane_px = static_cast<size_t>(s.dense_h) * s.dense_w;
    // per-layer scratch of the generic synthesis path and the dense stacks of the unfused upsampling: only when that path runs
    const bool need_dense = !s.use_fused_dec || s.cr;  // (the noise planes are channels [n_levels, 2 n_levels) of the dense stack)
    const bool need_layers = !s.use_fused_dec && !s.use_fused_syn;
    // fused kernel behind the pyramid launch: the level-1 stack (channels 1 .. n_levels - 1 at level 1's size) is stack B
    s.fdec_pre = s.use_fused_dec && b->opt_fused_dec == 2 && n_levels >= 5;
    const size_t o_stack_a = A.reserve(need_dense ? dense_elems * 4 : 16);
    const size_t o_stack_b = A.reserve((need_dense || s.fdec_pre) ? stack_b_elems * 4 : 16);
    const size_t o_tmp0 = A.reserve(need_layers ? plane_px * max_c * 4 : 16);
    const size_t o_tmp1 = A.reserve(need_layers ? plane_px * max_c * 4 : 16);
    const size_t o_stab = A.reserve(need_layers ? plane_px * std::max(h.out_channels, 1) * 4 : 16);
    const size_t o_synout = A.reserve(plane_px * std::max(h.out_channels, 1) * 4);
    const size_t o_out = need_resize ? A.reserve(static_cast<size_t>(H) * W * h.out_channels * 4) : o_synout;
    // the three integer planes in one block (plane p at a 256-byte aligned offset): one copy takes them to the host
    size_t o_planes = 0;
    const size_t sample_bytes = bitdepth == 8 ? 1 : 2;
    if (bitdepth) {
        size_t off = 0;
        for (int p = 0; p < 3; ++p) {
            const bool chroma420 = (frame_data_type == 1 && p > 0);
            s.plane_h[p] = chroma420 ? H / 2 : H;
            s.plane_w[p] = chroma420 ? W / 2 : W;
            s.plane_off[p] = off;
            off += (static_cast<size_t>(s.plane_h[p]) * s.plane_w[p] * sample_bytes + 16 + 255) & ~size_t{255};
        }
        s.planes_bytes = off;
        o_planes = A.reserve(off);
    }
    rc = A.commit(b->device);
    if (rc < 0) return rc;

    // ---- upload (inputs become resident in HBM here): the head, staged in pinned memory, one asynchronous copy on the
    // device's upload stream; launches order themselves behind it with an event (ccd_batch_run_stage) ----------------
    auto fail = [&](int code) { A.release(); s.staging.drop(); return code; };
    if (!s.staging.get(b->device, BlockPool::kPinned, head_bytes)) return fail(CCD_ERR_NOMEM);
    {
        char* st = s.staging.as<char>();
        std::memset(st + o_status, 0, 512);
        if (n_words) std::memcpy(st + o_words, bytes_latent, n_words * 4);
        std::memset(st + o_words + n_words * 4, 0, 8);
        std::memcpy(st + o_arm, arm_blob.data(), arm_blob.size() * 8);
        if (!ifce_blob.empty()) std::memcpy(st + o_ifce, ifce_blob.data(), ifce_blob.size() * 8);
        if (!syn_blob.empty()) std::memcpy(st + o_synp, syn_blob.data(), syn_blob.size() * 4);
        if (hipMemcpyAsync(A.at<void>(0), st, head_bytes, hipMemcpyHostToDevice, b->up_stream) != hipSuccess) return fail(CCD_ERR_HIP);
        if (hipEventRecord(b->up_done, b->up_stream) != hipSuccess) return fail(CCD_ERR_HIP);
        b->uploads_unconfirmed = true;
    }

    // ---- entropy stage description -----------------------------------------------------------------------
    EntropyParams& E = s.ep;
    std::memset(&E, 0, sizeof(E));
    E.words = A.at<uint32_t>(o_words); E.n_words = static_cast<uint32_t>(n_words);
    E.n_grids = h.n_grids;
    int level = 0;
    for (int g = 0; g < h.n_grids; ++g) {
        E.grid_h[g] = h.grid_h[g]; E.grid_w[g] = h.grid_w[g];
        E.latent[g] = A.at<int8_t>(o_lat[g]);
        E.ifce_in[g] = h.input_features_ifce[g];
        E.ifce_off[g] = ifce_off[g];
        if (g > 0 && (h.grid_h[g] != h.grid_h[g - 1] || h.grid_w[g] != h.grid_w[g - 1])) ++level;
        E.level[g] = level;
    }
    E.dim = h.total_context_arm; E.n_spatial = h.spatial_context_arm; E.n_ifce_out = h.output_feature_ifce;
    E.n_layers = h.n_hidden_layers_arm + 1;
    E.narrow = net.arm.narrow ? 1 : 0;
    E.ring_rows = s.ring_rows;
    E.mfma = s.use_mfma ? std::min(std::max(b->opt_mfma_arm == 1 ? 23 : b->opt_mfma_arm, 1), 23) : 0;
    E.has_ifce = h.has_ifce_resolution;
    context_offsets(h.spatial_context_arm, E.ctx_dy, E.ctx_dx);
    E.arm = A.at<int64_t>(o_arm); E.arm_len = static_cast<int32_t>(arm_blob.size());
    E.ifce = A.at<int64_t>(o_ifce);
    E.ifce_feat = A.at<int32_t>(o_feat);
    E.ifce_wide = A.at<int32_t>(o_feat) + feat_elems;
    E.feat_bits = b->opt_range_bits ? std::min(std::max(b->opt_range_bits, 8), 15) : 15;
    E.ifce_w32 = net.ifce_w32 ? 1 : 0;
    E.scale_table = b->d_scale_table;
    E.rcp_table = b->d_rcp_table;
    E.status = A.at<int32_t>(o_status);
    s.d_status = E.status;

    // ---- upsampling levels (upsampling.py:486-498): coarsest -> finest -------------------------------------
    s.levels.clear();
    float* stack_a = A.at<float>(o_stack_a);
    float* stack_b = A.at<float>(o_stack_b);
    const int n_steps = n_levels - 1;
    const float* prev = nullptr;
    for (int step = 0; step < n_steps; ++step) {
        const int g_in = lat_grids[n_levels - 1 - step], g_out = lat_grids[n_levels - 2 - step];
        UpsampleLevel L;
        std::memset(&L, 0, sizeof(L));
        L.in_f32 = prev;
        L.in_i8 = (step == 0) ? E.latent[g_in] : nullptr;
        L.target = E.latent[g_out];
        L.out = ((n_steps - 1 - step) % 2 == 0) ? stack_a : stack_b;
        L.c_in = step + 1;
        L.h_in = h.grid_h[g_in]; L.w_in = h.grid_w[g_in];
        L.h_out = h.grid_h[g_out]; L.w_out = h.grid_w[g_out];
        L.ups_k = net.ups_k; L.pre_k = net.pre_k;
        const int kidx = step % net.n_ups;
        std::copy_n(&net.ups_w[static_cast<size_t>(kidx) * net.ups_k], net.ups_k, L.ups_w);
        std::copy_n(&net.pre_w[static_cast<size_t>(kidx) * net.pre_k], net.pre_k, L.pre_w);
        s.levels.push_back(L);
        prev = L.out;
    }
    s.d_dense = stack_a;
    if (s.cr) {
        s.d_noise = A.at<float>(o_noise);
        s.d_nstack[0] = A.at<float>(o_nstack[0]); s.d_nstack[1] = A.at<float>(o_nstack[1]);
    }
    s.d_syn_params = A.at<float>(o_synp);
    s.d_tmp[0] = A.at<float>(o_tmp0); s.d_tmp[1] = A.at<float>(o_tmp1);
    s.d_stab = A.at<float>(o_stab);
    s.d_syn_out = A.at<float>(o_synout);
    s.d_out = A.at<float>(o_out);
    for (int p = 0; p < 3; ++p) s.d_plane[p] = bitdepth ? A.at<void>(o_planes + s.plane_off[p]) : nullptr;
    if (s.use_fused_syn) {
        SynthFused& F = s.fused;
        F.dense = s.d_dense; F.params = s.d_syn_params; F.h = s.dense_h; F.w = s.dense_w;
        F.bitdepth = bitdepth ? bitdepth : 8;
        // integer samples straight from the synthesis kernel for RGB / 4:4:4 frames at full resolution
        F.write_planes = (bitdepth != 0 && frame_data_type != 1 && !need_resize) ? 1 : 0;
        F.out = s.d_syn_out;
        for (int p = 0; p < 3; ++p) F.plane[p] = s.d_plane[p];
    }

    if (s.use_fused_dec) {
        FusedDec& D = s.fdec;
        D.params = s.d_syn_params + reinterpret_cast<size_t>(D.params);
        for (int i = 0; i < n_levels; ++i) { D.lat[i] = E.latent[lat_grids[i]]; D.lh[i] = h.grid_h[lat_grids[i]]; D.lw[i] = h.grid_w[lat_grids[i]]; }
        D.bitdepth = bitdepth ? bitdepth : 8;
        // integer samples straight from the kernel's epilogue: 1 = three full-size planes (rgb / yuv444), 2 = yuv420 (luma +
        // the 2 x 2 means of the chroma quads)
        D.write_planes = (bitdepth != 0 && !need_resize) ? (frame_data_type == 1 ? (h.out_channels >= 3 ? 2 : 0) : 1) : 0;
        // float samples: always when nothing else is produced (or a later stage reads them); otherwise by CCD_OPT_KEEP_FLOAT
        D.out = (!D.write_planes || b->opt_keep_float) ? s.d_syn_out : nullptr;
        for (int p = 0; p < 3; ++p) D.plane[p] = s.d_plane[p];
        D.l1 = nullptr;
        D.noise = s.cr ? s.d_dense + static_cast<size_t>(n_levels) * s.dense_h * s.dense_w : nullptr;
        if (s.fdec_pre) {
            // the pyramid launch's descriptor: this frame's levels 1 .. n - 1 as levels 0 .. n - 2, 64 x 32 tiles without margin,
            // output = the level-1 stack the main launch's tiles load
            FusedDec& Y = s.fpyr;
            std::memset(&Y, 0, sizeof(Y));
            Y.n_lv = n_levels - 1;
            for (int i = 0; i + 1 < n_levels; ++i) {
                Y.lat[i] = D.lat[i + 1]; Y.lh[i] = D.lh[i + 1]; Y.lw[i] = D.lw[i + 1];
                std::memcpy(Y.k2u[i], D.k2u[i + 1], sizeof(Y.k2u[i]));
                std::memcpy(Y.k2p[i], D.k2p[i + 1], sizeof(Y.k2p[i]));
            }
            Y.h = Y.lh[0]; Y.w = Y.lw[0]; Y.c = 2; Y.bitdepth = 8;
            Y.tiles_x = (Y.w + 63) / 64; Y.tiles_y = (Y.h + 31) / 32;
            Y.out = stack_b;
            D.l1 = stack_b;
        }
        s.levels.clear();  // no unfused pyramid steps for this slot
        s.use_fused_syn = false;
    }
    if (s.use_pipe) b->lds_pipe = std::max(b->lds_pipe, s.lds_pipe);
    else b->lds_generic = std::max(b->lds_generic, s.lds_generic);
    b->slots.push_back(std::move(sp));
    return static_cast<int>(b->slots.size()) - 1;
}

// XCD-aware order of one launch's work list (r06).  The hardware deals the workgroups of a launch out to the 8 XCDs round-robin
// (workgroup i -> XCD i mod 8, tools/ubench/queues.hip prints it) and every XCD has its own L2: with the work items in raster
// order, neighbouring tiles - which share the halo rows / columns of the level-1 stack and the latent tile - always sat on
// DIFFERENT XCDs and each fetched the shared cache lines from HBM for itself (main launch of the fused float path: 165 MB
// fetched per 24 Kodak frames for ~70 MB of stack + latents, profiles/r05/kodak24_pmc_traffic.json).  Here item j of the natural
// order goes to a workgroup of XCD x = the eighth of the list it lies in: every XCD walks ONE contiguous run of tiles.
// Chain groups of a batch (pure: ccd_debug_chain_groups exposes it to the CPU tests).  est[i]: expected chain of slot i; inst[i]: its
// kernel instantiation (0 .. k - 1; -1: the generic kernel's launch); n_conc: streams that really run at once; n_cu: CUs of the
// device.  cg[i] = 0 for the slots within 3 % of the batch's longest chain, 1 within 20 %, 2 for the rest - capped so that
// (a) instantiations x groups <= n_conc: a launch per group only helps on a stream of its own;
// (b) every workgroup still finds a CU at once.  A stream's workgroup owns a CU (139 KB of LDS) and the hardware deals the
//     workgroups of ONE launch out to the 8 XCDs round-robin: two launches of 63 + 193 workgroups put 8 + 25 on one 32-CU XCD, the
//     33rd waits for a whole chain - 68 ms instead of 36.6 (profiles/r06/streams_in_flight_overlap_before_xcd_rule.txt; a single
//     launch of 256 deals 32 to each): sum over launches of ceil(n / 8) <= CUs / 8 - else the group boundaries are moved to
//     multiples of 8 streams (c, below), else fewer groups.
static void plan_chain_groups(const double* est, const int* inst, int n, int n_conc, int n_cu, int* cg) {
    constexpr int kXcd = 8;  // gfx950
    double est_max = 0.0;
    int n_inst = 0, n_generic = 0;
    for (int i = 0; i < n; ++i) {
        if (inst[i] < 0) { ++n_generic; continue; }
        est_max = std::max(est_max, est[i]);
        n_inst = std::max(n_inst, inst[i] + 1);
    }
    int max_cg = std::max(1, std::min(3, n_conc / std::max(1, n_inst)));
    const auto fits = [&](int groups) {
        int per_xcd = (n_generic + kXcd - 1) / kXcd;
        for (int k = 0; k < n_inst; ++k)
            for (int g = 0; g < groups; ++g) {
                int cnt = 0;
                for (int i = 0; i < n; ++i) cnt += (inst[i] == k && cg[i] == g) ? 1 : 0;
                per_xcd += (cnt + kXcd - 1) / kXcd;
            }
        return per_xcd <= n_cu / kXcd;
    };
    for (; max_cg >= 1; --max_cg) {
        for (int i = 0; i < n; ++i) {
            const int c = est[i] >= 0.97 * est_max ? 0 : (est[i] >= 0.80 * est_max ? 1 : 2);
            cg[i] = inst[i] < 0 ? 0 : std::min(c, max_cg - 1);
        }
        if (max_cg == 1 || fits(max_cg)) break;
        // (c) a nearly full chip: the same split with every group boundary moved to a multiple of 8 streams - the slowest streams of
        //     the next group join the slower one (their frames are synthesised a little later, nothing else changes) - deals whole
        //     rounds to the XCDs: 63 + 193 becomes 64 + 192 = 8 + 24 per XCD
        for (int k = 0; k < n_inst; ++k) {
            std::vector<int> idx;
            for (int i = 0; i < n; ++i) if (inst[i] == k) idx.push_back(i);
            std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return cg[x] != cg[y] ? cg[x] < cg[y] : est[x] > est[y]; });
            int bound = 0, prev = 0;
            for (int g = 0; g + 1 < max_cg; ++g) {
                for (int i : idx) bound += cg[i] == g ? 1 : 0;
                const int rounded = std::min(static_cast<int>(idx.size()), std::max(prev, (bound + kXcd - 1) / kXcd * kXcd));
                for (int r = prev; r < rounded; ++r) cg[idx[r]] = g;
                prev = rounded;
            }
            for (int r = prev; r < static_cast<int>(idx.size()); ++r) cg[idx[r]] = max_cg - 1;
        }
        if (fits(max_cg)) break;
    }
}

// Expected length of a slot's serial chain in decoder ticks (only the ORDER and rough ratios matter: it decides which streams
// share an entropy launch).  Per grid ~120 ticks per symbol + ~1.8 k per wavefront step (latent.py:66-140: W + 10 (H - 1) steps):
// profiles/r06/prof_grids_base.txt - a portrait Kodak stream comes out 1.09 x a landscape one (measured 1.06).
static double chain_estimate(const EntropyParams& ep) {
    double t = 0.0;
    for (int g = 0; g < ep.n_grids; ++g) {
        const double h = ep.grid_h[g], w = ep.grid_w[g];
        t += 120.0 * h * w + 1800.0 * (w > 9.0 ? w + 10.0 * (h - 1.0) : h * w);
    }
    return t;
}

static int upload_params(ccd_batch* b, hipStream_t st) {
    const int n = static_cast<int>(b->slots.size());
    if (b->n_params_uploaded == n && !b->regroup) return CCD_OK;
    b->regroup = false;
    std::vector<EntropyParams> host;
    std::vector<int> host_slot;  // slot of host[k]: its status words are words [64 slot, 64 slot + 64) of the batch's status array
    b->pipe_groups.clear();  // the pipelined kernel is instantiated per input width nv = ceil(dim / 4): one launch per width
    // ---- chain groups (r06).  A stream is one serial chain on one CU and a launch ends with its slowest stream; the float path of
    // a frame only needs THAT frame's latents.  So the slots of an instantiation are split by expected chain length into up to three
    // launches - the streams within 3 % of the batch's longest chain, those within 20 %, the rest - and ccd_batch_run puts each
    // launch's pyramid + fused launches directly behind it on its own stream: the float stage of the streams that finish early
    // hides behind the longest chains (kodak24: 18 landscape pictures are done 2 ms before the 6 portrait ones).  At most ~4
    // launches per batch: HIP streams share a handful of hardware queues.
    std::vector<double> est(n, 0.0);
    std::vector<int> cg_of(n, 0), inst_of(n, -1);
    {
        std::vector<std::array<int, 3>> insts;
        for (int i = 0; i < n; ++i) {
            const Slot& sl = *b->slots[i];
            est[i] = chain_estimate(sl.ep);
            if (!sl.use_pipe) continue;  // (-1: the generic launch)
            const std::array<int, 3> key{(sl.ep.dim + 3) / 4, sl.use_mfma ? 2 : (sl.use_dyn ? 1 : 0), sl.fixed_shape};
            auto it = std::find(insts.begin(), insts.end(), key);
            inst_of[i] = static_cast<int>(it - insts.begin());
            if (it == insts.end()) insts.push_back(key);
        }
        // as many launches as streams really run at once (DeviceShared::n_conc, measured), shared between the instantiations
        int n_conc = 1, n_cu = 256;
        { DeviceShared* shd = nullptr; if (device_shared(b->device, &shd) >= 0) { n_conc = shd->n_conc; n_cu = shd->n_cu; } }
        plan_chain_groups(est.data(), inst_of.data(), n, b->opt_overlap ? n_conc : 1, n_cu, cg_of.data());
    }
    for (int nv = 1; nv <= 8; ++nv)
        for (int var = 0; var < 3; ++var)  // vector ALU without / with the device check of the features, matrix cores
            for (int shape = 0; shape < 2; ++shape)  // run-time ARM shape / the compile-time instantiation of the HOP shape
                for (int cg = 0; cg < 3; ++cg) {
                    const int mf = var == 2 ? 1 : 0, dyn = var == 1 ? 1 : 0;
                    const int first = static_cast<int>(host.size());
                    size_t lds = 0;
                    double est_g = 0.0;
                    for (int i = 0; i < n; ++i) {
                        Slot& sl = *b->slots[i];
                        if (sl.use_pipe && (sl.ep.dim + 3) / 4 == nv && (sl.use_mfma ? 1 : 0) == mf && (sl.use_dyn ? 1 : 0) == dyn && sl.fixed_shape == shape && cg_of[i] == cg) {
                            host.push_back(sl.ep); host_slot.push_back(i); lds = std::max(lds, sl.lds_pipe);
                            est_g = std::max(est_g, est[i]);
                            sl.lg = static_cast<int>(b->pipe_groups.size());
                        }
                    }
                    if (static_cast<int>(host.size()) > first) b->pipe_groups.push_back({nv, mf, dyn, shape, cg, first, static_cast<int>(host.size()) - first, lds, est_g});
                }
    for (int i = 0; i < n; ++i) {
        Slot& sl = *b->slots[i];
        if (!sl.use_pipe) sl.lg = -1;
        // float launches keyed by the entropy launch they follow - only worth it (and only done) when there are several launches
        sl.fl = (b->opt_overlap && sl.use_pipe) ? sl.lg : -1;
    }
    b->n_pipe = static_cast<int>(host.size());
    for (int i = 0; i < n; ++i) if (!b->slots[i]->use_pipe) { host.push_back(b->slots[i]->ep); host_slot.push_back(i); }
    b->n_generic = n - b->n_pipe;
    // fused synthesis: one launch per (CP, C) group over all of its frames
    b->fused_groups.clear();
    std::vector<SynthFused> fused;
    for (int i = 0; i < n; ++i) {
        const Slot& s = *b->slots[i];
        if (!s.use_fused_syn) continue;
        const int cp = (s.fused.c_in + 3) / 4;
        bool placed = false;
        for (auto& g : b->fused_groups) placed = placed || (g.cp == cp && g.c == s.fused.c);
        if (!placed) b->fused_groups.push_back({cp, s.fused.c, s.fused.c_in, 0, 0, 0, 0});
    }
    for (auto& g : b->fused_groups) {
        g.first = static_cast<int>(fused.size());
        for (int i = 0; i < n; ++i) {
            const Slot& s = *b->slots[i];
            if (!s.use_fused_syn || (s.fused.c_in + 3) / 4 != g.cp || s.fused.c != g.c) continue;
            int tx = 0, ty = 0;
            syn_fused_tiles(s.fused.h, s.fused.w, s.fused.halo, &tx, &ty);
            g.max_tx = std::max(g.max_tx, tx); g.max_ty = std::max(g.max_ty, ty);
            fused.push_back(s.fused);
            ++g.n;
        }
    }
    // fused float path: frames grouped by (levels, channels); a workgroup takes a run of `per_wg` tiles of one frame
    b->fdec_groups.clear();
    struct Work { int32_t frame, tile_first, tile_count, pad; };
    const auto xcd_interleave = [](auto& work, size_t first, size_t n) {
        using W = typename std::remove_reference<decltype(work)>::type::value_type;
        constexpr size_t kXcd = 8;
        if (n < 2 * kXcd || std::getenv("CCD_NO_XCD_ORDER")) return;
        std::vector<W> nat(work.begin() + static_cast<std::ptrdiff_t>(first), work.begin() + static_cast<std::ptrdiff_t>(first + n));
        size_t start = 0;
        for (size_t x = 0; x < kXcd; ++x) {
            const size_t cnt = (n - x + kXcd - 1) / kXcd;  // workgroups x, x + 8, ... of the launch
            for (size_t k = 0; k < cnt; ++k) work[first + x + kXcd * k] = nat[start + k];
            start += cnt;
        }
    };
    std::vector<FusedDec> frames;
    std::vector<Work> work;
    {
        for (int i = 0; i < n; ++i) {
            const Slot& s = *b->slots[i];
            if (!s.use_fused_dec) continue;
            bool placed = false;
            for (auto& g : b->fdec_groups) placed = placed || (g.c_in == s.fdec.n_lv && g.c == s.fdec.c && g.pre == (s.fdec_pre ? 1 : 0) && g.cr == (s.cr ? 1 : 0) && g.fl == s.fl);
            if (!placed) b->fdec_groups.push_back({s.fdec.n_lv, s.fdec.c, s.fdec_pre ? 1 : 0, s.cr ? 1 : 0, s.fl, 0, 0, 0, 0});
        }
        for (auto& g : b->fdec_groups) {
            g.first_frame = static_cast<int>(frames.size());
            g.first_work = static_cast<int>(work.size());
            long total_tiles = 0;
            for (int i = 0; i < n; ++i) {
                const Slot& s = *b->slots[i];
                if (s.use_fused_dec && s.fdec.n_lv == g.c_in && s.fdec.c == g.c && (s.fdec_pre ? 1 : 0) == g.pre && (s.cr ? 1 : 0) == g.cr && s.fl == g.fl) total_tiles += static_cast<long>(s.fdec.tiles_x) * s.fdec.tiles_y;
            }
            // ~8 workgroups per CU keep the tail short; a run of tiles amortises the parameter staging
            const int per_wg = static_cast<int>(std::min<long>(8, std::max<long>(1, (total_tiles + 2047) / 2048)));
            for (int i = 0; i < n; ++i) {
                const Slot& s = *b->slots[i];
                if (!s.use_fused_dec || s.fdec.n_lv != g.c_in || s.fdec.c != g.c || (s.fdec_pre ? 1 : 0) != g.pre || (s.cr ? 1 : 0) != g.cr || s.fl != g.fl) continue;
                const int f = static_cast<int>(frames.size()) - g.first_frame;
                frames.push_back(s.fdec);
                g.lds = std::max(g.lds, s.fdec_lds);
                const int nt = s.fdec.tiles_x * s.fdec.tiles_y;
                for (int t0 = 0; t0 < nt; t0 += per_wg) work.push_back({f, t0, std::min(per_wg, nt - t0), 0});
            }
            g.n_work = static_cast<int>(work.size()) - g.first_work;
            xcd_interleave(work, static_cast<size_t>(g.first_work), static_cast<size_t>(g.n_work));
        }
    }
    // pyramid launches of the kFdPre slots: one per number of levels; a workgroup takes a run of tiles of one frame
    b->pyr_groups.clear();
    std::vector<FusedDec> pyr_frames;
    std::vector<Work> pyr_work;
    for (int lv = 2; lv < kFdMaxLevels; ++lv)
        for (int fl = -1; fl < static_cast<int>(b->pipe_groups.size()); ++fl) {
            ccd_batch::PyrGroup g{lv, fl, static_cast<int>(pyr_frames.size()), static_cast<int>(pyr_work.size()), 0, fused_pyr_lds_bytes(lv + 1)};
            long total_tiles = 0;
            for (int i = 0; i < n; ++i) {
                const Slot& s = *b->slots[i];
                if (s.fdec_pre && s.fpyr.n_lv == lv && s.fl == fl) total_tiles += static_cast<long>(s.fpyr.tiles_x) * s.fpyr.tiles_y;
            }
            if (!total_tiles) continue;
            const int per_wg = static_cast<int>(std::min<long>(8, std::max<long>(1, (total_tiles + 2047) / 2048)));
            for (int i = 0; i < n; ++i) {
                const Slot& s = *b->slots[i];
                if (!s.fdec_pre || s.fpyr.n_lv != lv || s.fl != fl) continue;
                const int f = static_cast<int>(pyr_frames.size()) - g.first_frame;
                pyr_frames.push_back(s.fpyr);
                const int nt = s.fpyr.tiles_x * s.fpyr.tiles_y;
                for (int t0 = 0; t0 < nt; t0 += per_wg) pyr_work.push_back({f, t0, std::min(per_wg, nt - t0), 0});
            }
            g.n_work = static_cast<int>(pyr_work.size()) - g.first_work;
            xcd_interleave(pyr_work, static_cast<size_t>(g.first_work), static_cast<size_t>(g.n_work));
            b->pyr_groups.push_back(g);
        }
    // upsampling steps: step k (k-th from the coarsest level) of all slots together
    b->ups_steps.clear();
    std::vector<UpsampleLevel> levels;
    std::vector<uint32_t> zmap;
    size_t max_steps = 0;
    for (int i = 0; i < n; ++i) max_steps = std::max(max_steps, b->slots[i]->levels.size());
    for (size_t k = 0; k < max_steps; ++k) {
        ccd_batch::UpsStep st{static_cast<int>(zmap.size()), 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const Slot& s = *b->slots[i];
            if (k >= s.levels.size()) continue;
            if (levels.size() >= 65535) return CCD_ERR_UNSUPPORTED;
            const uint32_t li = static_cast<uint32_t>(levels.size());
            levels.push_back(s.levels[k]);
            // group 0 = pre-concat conv; group g >= 1 = transposed conv of input channels 2g-2, 2g-1
            for (int grp = 0; grp <= (s.levels[k].c_in + 1) / 2; ++grp) zmap.push_back((li << 16) | static_cast<uint32_t>(grp));
            st.max_w = std::max(st.max_w, static_cast<int>(s.levels[k].w_out));
            st.max_h = std::max(st.max_h, static_cast<int>(s.levels[k].h_out));
        }
        st.n_z = static_cast<int>(zmap.size()) - st.first_z;
        b->ups_steps.push_back(st);
    }
    // ---- one pooled device block for every table and the status words of all slots, filled by ONE copy from ONE pinned block
    auto up256 = [](size_t v) { return (v + 255) & ~size_t{255}; };
    const size_t o_params = 0;
    const size_t o_fusedt = o_params + up256(sizeof(EntropyParams) * std::max(n, 1));
    const size_t o_fdec = o_fusedt + up256(sizeof(SynthFused) * std::max<size_t>(fused.size(), 1));
    const size_t o_work = o_fdec + up256(sizeof(FusedDec) * std::max<size_t>(frames.size(), 1));
    const size_t o_levels = o_work + up256(sizeof(Work) * std::max<size_t>(work.size(), 1));
    const size_t o_zmap = o_levels + up256(sizeof(UpsampleLevel) * std::max<size_t>(levels.size(), 1));
    const size_t o_pyr = o_zmap + up256(sizeof(uint32_t) * std::max<size_t>(zmap.size(), 1));
    const size_t o_pyrw = o_pyr + up256(sizeof(FusedDec) * std::max<size_t>(pyr_frames.size(), 1));
    const size_t o_stat = o_pyrw + up256(sizeof(Work) * std::max<size_t>(pyr_work.size(), 1));
    const size_t total = o_stat + up256(static_cast<size_t>(std::max(n, 1)) * 64 * sizeof(int32_t));
    // the previous tables may still be read by launches in flight on the caller's stream (a batch that grew between runs)
    if (b->tables.p && b->drain_streams() < 0) return CCD_ERR_HIP;
    if (!b->tables.get(b->device, BlockPool::kDevice, total) || !b->tables_staging.get(b->device, BlockPool::kPinned, total) ||
        !b->status_host.get(b->device, BlockPool::kPinned, static_cast<size_t>(std::max(n, 1)) * 64 * sizeof(int32_t)))
        return CCD_ERR_NOMEM;
    char* dev = b->tables.as<char>();
    char* stg = b->tables_staging.as<char>();
    b->d_params = reinterpret_cast<EntropyParams*>(dev + o_params);
    b->d_fused = reinterpret_cast<SynthFused*>(dev + o_fusedt);
    b->d_fdec = reinterpret_cast<FusedDec*>(dev + o_fdec);
    b->d_fdec_work = dev + o_work;
    b->d_levels = reinterpret_cast<UpsampleLevel*>(dev + o_levels);
    b->d_zmap = reinterpret_cast<uint32_t*>(dev + o_zmap);
    b->d_pyr = reinterpret_cast<FusedDec*>(dev + o_pyr);
    b->d_pyr_work = dev + o_pyrw;
    b->d_status_all = reinterpret_cast<int32_t*>(dev + o_stat);
    for (int k = 0; k < n; ++k) {
        host[k].status = b->d_status_all + static_cast<size_t>(host_slot[k]) * 64;
        b->slots[host_slot[k]]->d_status = host[k].status;
    }
    if (n) std::memcpy(stg + o_params, host.data(), sizeof(EntropyParams) * n);
    if (!fused.empty()) std::memcpy(stg + o_fusedt, fused.data(), sizeof(SynthFused) * fused.size());
    if (!frames.empty()) std::memcpy(stg + o_fdec, frames.data(), sizeof(FusedDec) * frames.size());
    if (!work.empty()) std::memcpy(stg + o_work, work.data(), sizeof(Work) * work.size());
    if (!levels.empty()) std::memcpy(stg + o_levels, levels.data(), sizeof(UpsampleLevel) * levels.size());
    if (!zmap.empty()) std::memcpy(stg + o_zmap, zmap.data(), sizeof(uint32_t) * zmap.size());
    if (!pyr_frames.empty()) std::memcpy(stg + o_pyr, pyr_frames.data(), sizeof(FusedDec) * pyr_frames.size());
    if (!pyr_work.empty()) std::memcpy(stg + o_pyrw, pyr_work.data(), sizeof(Work) * pyr_work.size());
    std::memset(stg + o_stat, 0, total - o_stat);
    HIP_TRY(hipMemcpyAsync(dev, stg, total, hipMemcpyHostToDevice, st));
    // a later run on ANOTHER stream (ccd_batch_prepare on one, ccd_batch_run on the next) orders itself behind this copy
    if (!b->params_up) HIP_TRY(hipEventCreateWithFlags(&b->params_up, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->params_up, st));
    b->params_stream = st;
    b->n_params_uploaded = n;
    return CCD_OK;
}

// Common-randomness planes (coolchic.py:179-183): Gaussian grids at every latent level, then
// fixed_upsampling(mode="bicubic") coarsest -> finest into channels [n_levels, 2 n_levels) of the dense stack.
static int run_common_randomness(Slot& s, hipStream_t st) {
    const int n = static_cast<int>(s.lvl_h.size());
    HIP_TRY(launch_cr_noise(s.d_noise, s.noise_off[n], st));
    const size_t plane0 = static_cast<size_t>(s.dense_h) * s.dense_w;
    // stack at level lv: [target noise lv, upsampled planes of levels lv+1 .. n-1]
    auto stack_at = [&](int lv) { return lv == 0 ? s.d_dense + static_cast<size_t>(n) * plane0 : s.d_nstack[(lv - 1) & 1]; };
    const float* cur = s.d_noise + s.noise_off[n - 1];
    int ch = s.lvl_h[n - 1], cw = s.lvl_w[n - 1], cc = 1;
    if (n == 1) {
        HIP_TRY(hipMemcpyAsync(stack_at(0), cur, plane0 * 4, hipMemcpyDeviceToDevice, st));
        return CCD_OK;
    }
    for (int lv = n - 2; lv >= 0; --lv) {
        const int th = s.lvl_h[lv], tw = s.lvl_w[lv];
        const size_t tp = static_cast<size_t>(th) * tw;
        float* dst = stack_at(lv);  // levels >= 3 fit in the alternating level-1 / level-2 stacks (sizes shrink with lv)
        HIP_TRY(hipMemcpyAsync(dst, s.d_noise + s.noise_off[lv], tp * 4, hipMemcpyDeviceToDevice, st));
        if (th != ch || tw != cw) HIP_TRY(launch_resize_interp(cur, dst + tp, cc, ch, cw, th, tw, 1, 0.5f, 0.5f, st));
        else HIP_TRY(hipMemcpyAsync(dst + tp, cur, static_cast<size_t>(cc) * tp * 4, hipMemcpyDeviceToDevice, st));
        cur = dst; ch = th; cw = tw; ++cc;
    }
    return CCD_OK;
}

static int run_upsampling(Slot& s, hipStream_t st) {
    if (s.cr) { const int rc = run_common_randomness(s, st); if (rc < 0) return rc; }
    if (s.use_fused_dec) return CCD_OK;  // the pyramid is evaluated inside the fused kernel (stage 2)
    if (s.levels.empty()) {
        const int g = [&] { for (int i = 0; i < s.hdr.n_grids; ++i) if (!s.hdr.is_hyperlatent[i]) return i; return 0; }();
        HIP_TRY(launch_i8_to_f32(s.ep.latent[g], s.d_dense, static_cast<size_t>(s.dense_h) * s.dense_w, st));
        return CCD_OK;
    }
    return CCD_OK;  // the pyramid steps were launched for the whole batch (ccd_batch_run_stage)
}

// a slot whose whole float path - fused kernel, final resize, integer planes - can follow its entropy launch on that launch's stream
static bool tail_keyed(const Slot& s) { return s.fl >= 0 && s.use_fused_dec && !s.cr; }

static int run_synthesis(Slot& s, hipStream_t st, bool keyed_done = false) {
    const Network& net = s.net;
    const int h = s.dense_h, w = s.dense_w;
    if (s.use_fused_syn || s.use_fused_dec) {  // the fused kernel itself was launched for the whole group (ccd_batch_run_stage)
        if (keyed_done && tail_keyed(s)) return CCD_OK;  // ... and so were its resize / planes launches (launch_entropy_groups)
        const int H = s.hdr.img_size[0], W = s.hdr.img_size[1];
        if (s.d_out != s.d_syn_out) HIP_TRY(launch_final_resize(s.d_syn_out, s.d_out, s.hdr.out_channels, h, w, H, W, s.hdr.final_upsampling_type, st));
        if (s.bitdepth && !(s.use_fused_dec ? s.fdec.write_planes : s.fused.write_planes))
            HIP_TRY(launch_planes(s.d_out, s.d_plane[0], s.d_plane[1], s.d_plane[2], H, W, s.bitdepth, s.frame_data_type, st));
        return CCD_OK;
    }
    const float* x = s.d_dense;
    int cur = 0;
    for (size_t l = 0; l < net.syn.size(); ++l) {
        const SynLayerParams& L = net.syn[l];
        HIP_TRY(launch_syn_layer(x, nullptr, s.d_syn_params + s.w_off[l], s.d_syn_params + s.b_off[l], s.d_tmp[cur], L.c_in,
                                 L.c_out, L.k, L.residual, L.relu, h, w, st));
        x = s.d_tmp[cur];
        cur ^= 1;
    }
    const float* stab = nullptr;
    if (net.syn_stab.c_out) {
        HIP_TRY(launch_syn_layer(s.d_dense, nullptr, s.d_syn_params + s.stab_w, s.d_syn_params + s.stab_b, s.d_stab,
                                 net.syn_stab.c_in, net.syn_stab.c_out, 1, 0, 0, h, w, st));
        stab = s.d_stab;
    }
    HIP_TRY(launch_syn_layer(x, stab, s.d_syn_params + s.out_w, s.d_syn_params + s.out_b, s.d_syn_out, net.syn_out.c_in,
                             net.syn_out.c_out, 1, 0, 0, h, w, st));
    const int H = s.hdr.img_size[0], W = s.hdr.img_size[1];
    if (s.d_out != s.d_syn_out)
        HIP_TRY(launch_final_resize(s.d_syn_out, s.d_out, s.hdr.out_channels, h, w, H, W, s.hdr.final_upsampling_type, st));
    if (s.bitdepth)
        HIP_TRY(launch_planes(s.d_out, s.d_plane[0], s.d_plane[1], s.d_plane[2], H, W, s.bitdepth, s.frame_data_type, st));
    return CCD_OK;
}

// Entropy launches of a batch (and, with `with_float`, each launch's own float-path launches right behind it) forked over the
// device's side streams and joined on `st`.  Launch 0 - the one with the longest expected chains - stays on the caller's stream.
// With ONE launch there is nothing to fork: it goes to `st` and the float stages follow it there (the caller enqueues them).
static int launch_entropy_groups(ccd_batch* b, hipStream_t st, bool with_float) {
    // launch order: longest expected chains first (they start first where launches queue behind each other)
    std::vector<int> order(b->pipe_groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->pipe_groups[x].est > b->pipe_groups[y].est; });
    const int n_launch = static_cast<int>(order.size()) + (b->n_generic > 0 ? 1 : 0);
    DeviceShared* sh = nullptr;
    if (n_launch > 1) {
        const int rc = device_shared(b->device, &sh);
        if (rc < 0) return rc;
        if (!b->fork) HIP_TRY(hipEventCreateWithFlags(&b->fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->fork, st));
    }
    std::vector<int> used;
    // one launch: the caller's stream.  Several: ALL of them on the side streams that were measured to run concurrently
    // (DeviceShared::conc), launch k on conc[k mod n_conc] - the caller's stream only forks and joins.
    auto stream_for = [&](int idx, int* side_out) -> hipStream_t {
        *side_out = -1;
        if (!sh) return st;
        const int side = sh->conc[idx % sh->n_conc];
        if (std::find(used.begin(), used.end(), side) == used.end()) {
            used.push_back(side);
            (void)hipStreamWaitEvent(sh->side[side], b->fork, 0);
        }
        *side_out = side;
        return sh->side[side];
    };
    // an event behind everything this batch has put on a side stream so far: what a destroy / a table replacement waits for, also
    // after an error between the fork and the join below (the shared side streams themselves are never drained)
    auto mark = [&](int side) -> int {
        if (side < 0) return CCD_OK;
        if (!b->side_done[side]) HIP_TRY(hipEventCreateWithFlags(&b->side_done[side], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->side_done[side], sh->side[side]));
        b->side_pending[side] = true;
        return CCD_OK;
    };
    int k = 0;
    if (b->opt_time_launches) {
        while (static_cast<int>(b->lt0.size()) < n_launch) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIP_TRY(hipEventCreate(&e0));
            b->lt0.push_back(e0);
            HIP_TRY(hipEventCreate(&e1));
            b->lt1.push_back(e1);
        }
        b->n_timed = n_launch;
    }
    for (int gi : order) {
        const auto& g = b->pipe_groups[gi];
        int side = -1;
        const int kk = k;
        hipStream_t s = stream_for(k++, &side);
        if (b->opt_time_launches) HIP_TRY(hipEventRecord(b->lt0[kk], s));
        hipError_t e = launch_entropy_pipe(b->d_params + g.first, g.n, g.nv, g.mfma, g.dyn, g.shape, g.lds, s);
        if (b->opt_time_launches) HIP_TRY(hipEventRecord(b->lt1[kk], s));
        if (e == hipSuccess && with_float) {
            // this launch's frames: pyramid launch(es), then the fused kernel - on the SAME stream, so they start when this launch's
            // slowest stream is done, whatever the other launches are doing.  (Common randomness needs its noise planes first:
            // those groups run behind the join like every per-slot launch.)
            for (const auto& pg : b->pyr_groups)
                if (pg.fl == gi && e == hipSuccess)
                    e = launch_fused_pyramid(b->d_pyr + pg.first_frame, static_cast<const char*>(b->d_pyr_work) + static_cast<size_t>(pg.first_work) * 16, pg.n_work, pg.levels, pg.lds, s);
            for (const auto& fg : b->fdec_groups)
                if (fg.fl == gi && !fg.cr && e == hipSuccess)
                    e = launch_fused_dec(b->d_fdec + fg.first_frame, static_cast<const char*>(b->d_fdec_work) + static_cast<size_t>(fg.first_work) * 16, fg.n_work, fg.c_in, fg.c, fg.pre, fg.lds, s);
        }
        if (e == hipSuccess && with_float) {
            // ... and what follows the fused kernel per slot: the final resize (the motion cool-chics' nearest x 4) and, where the
            // fused kernel did not write them, the integer planes
            for (auto& sp : b->slots)
                if (sp->lg == gi && tail_keyed(*sp) && run_synthesis(*sp, s) < 0) { e = hipErrorUnknown; break; }
        }
        if (e == hipSuccess && with_float) {
            while (b->lg_done.size() <= static_cast<size_t>(gi)) b->lg_done.push_back(nullptr);
            if (!b->lg_done[gi] && hipEventCreateWithFlags(&b->lg_done[gi], hipEventDisableTiming) != hipSuccess) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipEventRecord(b->lg_done[gi], s);
        }
        const int rc = mark(side);
        if (e != hipSuccess) return CCD_ERR_HIP;
        if (rc < 0) return rc;
    }
    b->lg_valid = with_float;
    if (b->n_generic > 0) {
        int side = -1;
        const int kk = k;
        hipStream_t s = stream_for(k++, &side);
        if (b->opt_time_launches) HIP_TRY(hipEventRecord(b->lt0[kk], s));
        const hipError_t e = launch_entropy(b->d_params + b->n_pipe, b->n_generic, b->lds_generic, s);
        if (b->opt_time_launches) HIP_TRY(hipEventRecord(b->lt1[kk], s));
        const int rc = mark(side);
        if (e != hipSuccess) return CCD_ERR_HIP;
        if (rc < 0) return rc;
    }
    for (int side : used) HIP_TRY(hipStreamWaitEvent(st, b->side_done[side], 0));
    return CCD_OK;
}

// the float-path launches of stage 1 / stage 2 that belong to the whole batch; `keyed_done`: the pyramid / fused launches keyed by
// an entropy launch were already enqueued behind it (launch_entropy_groups with_float)
static int launch_float_stage(ccd_batch* b, hipStream_t st, int stage, bool keyed_done) {
    if (stage == 1) {
        for (const auto& u : b->ups_steps)
            HIP_TRY(launch_upsample_step(b->d_levels, b->d_zmap + u.first_z, u.n_z, u.max_w, u.max_h, st));
        for (const auto& g : b->pyr_groups) {
            if (keyed_done && g.fl >= 0) continue;
            HIP_TRY(launch_fused_pyramid(b->d_pyr + g.first_frame, static_cast<const char*>(b->d_pyr_work) + static_cast<size_t>(g.first_work) * 16,
                                         g.n_work, g.levels, g.lds, st));
        }
    }
    if (stage == 2) {
        for (const auto& g : b->fused_groups)
            HIP_TRY(launch_syn_fused(b->d_fused + g.first, g.n, g.c_in, g.c, g.max_tx, g.max_ty, st));
        for (const auto& g : b->fdec_groups) {
            if (keyed_done && g.fl >= 0 && !g.cr) continue;
            const FusedDec* fr = b->d_fdec + g.first_frame;
            const char* wk = static_cast<const char*>(b->d_fdec_work) + static_cast<size_t>(g.first_work) * 16;
            if (g.cr) HIP_TRY(launch_fused_dec_cr(fr, wk, g.n_work, g.c_in, g.c, g.lds, st));
            else HIP_TRY(launch_fused_dec(fr, wk, g.n_work, g.c_in, g.c, g.pre, g.lds, st));
        }
    }
    for (auto& sp : b->slots) {
        const int rc = (stage == 1) ? run_upsampling(*sp, st) : run_synthesis(*sp, st, keyed_done);
        if (rc < 0) return rc;
    }
    return CCD_OK;
}

// common head of a run: device, the slots' uploads, the launch tables (and their copy, if another stream carried it)
static int run_prologue(ccd_batch* b, hipStream_t st) {
    HIP_TRY(hipSetDevice(b->device));
    if (b->uploads_unconfirmed) HIP_TRY(hipStreamWaitEvent(st, b->up_done, 0));  // the slots' uploads (ccd_batch_add) come first
    b->note_stream(st);
    const int rc = upload_params(b, st);
    if (rc < 0) return rc;
    if (b->params_up && b->params_stream != st) HIP_TRY(hipStreamWaitEvent(st, b->params_up, 0));
    return CCD_OK;
}

int ccd_batch_run_stage(ccd_batch* b, void* stream, int stage) {
    if (!b || stage < 0 || stage > 2) return CCD_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rc = run_prologue(b, st);
    if (rc < 0) return rc;
    // stage 0: one launch per kernel instantiation and chain group in use.  The first goes to the caller's stream; the others fork
    // to side streams and join again, so that they overlap (each stream of a launch occupies one CU for its whole serial chain:
    // queued on one stream, a GOP whose I frames need another instantiation than its B frames took the SUM of the two).
    if (stage == 0) return launch_entropy_groups(b, st, false);
    return launch_float_stage(b, st, stage, false);
}

int ccd_batch_prepare(ccd_batch* b, void* stream) {
    if (!b) return CCD_ERR_ARG;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    b->note_stream(st);
    return upload_params(b, st);
}

// All three stages.  Unlike three ccd_batch_run_stage calls, the float path of a frame does not wait for the slowest stream of
// the BATCH: every entropy launch (kernel instantiation x chain group, upload_params) is followed on its own stream by the
// pyramid + fused launches of its own frames, and the streams join once at the end (decode.py:67-81: frames are independent).
int ccd_batch_run(ccd_batch* b, void* stream) {
    if (!b) return CCD_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = run_prologue(b, st);
    if (rc < 0) return rc;
    const bool overlap = b->opt_overlap && b->pipe_groups.size() + (b->n_generic > 0 ? 1 : 0) > 1;
    rc = launch_entropy_groups(b, st, overlap);
    if (rc < 0) return rc;
    for (int stage = 1; stage <= 2; ++stage) {
        rc = launch_float_stage(b, st, stage, overlap);
        if (rc < 0) return rc;
    }
    return CCD_OK;
}

int ccd_batch_wait(ccd_batch* b, void* stream) {
    if (!b) return CCD_ERR_ARG;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    b->note_stream(st);
    // slots added after the last run have no status yet: the words of the slots that DID run are refreshed all the same (their
    // array and its pinned copy were sized for them)
    const size_t n = std::min(b->slots.size(), static_cast<size_t>(b->n_params_uploaded));
    if (n && b->d_status_all) {
        // the status words of all slots are one array: one copy into pinned memory, one wait
        HIP_TRY(hipMemcpyAsync(b->status_host.p, b->d_status_all, n * 64 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n == b->slots.size()) b->uploads_unconfirmed = false;  // every launch behind the uploads has finished
        const int32_t* hs = b->status_host.as<int32_t>();
        int first = CCD_OK;
        for (size_t i = 0; i < n; ++i) {
            Slot& sl = *b->slots[i];
            std::memcpy(sl.host_status, hs + i * 64, sizeof(sl.host_status));
            sl.status = sl.host_status[0];
            if (first == CCD_OK && sl.status != CCD_OK) first = sl.status;
        }
        return first;
    }
    HIP_TRY(hipStreamSynchronize(st));  // nothing was run yet
    return CCD_OK;
}

int ccd_batch_slot_status(const ccd_batch* b, int slot) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    return b->slots[slot]->status;
}

int ccd_batch_slot_stats(const ccd_batch* b, int slot, int32_t* out64) {
    if (!b || !out64 || slot < 0 || slot >= static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    std::memcpy(out64, b->slots[slot]->host_status, sizeof(b->slots[slot]->host_status));
    return CCD_OK;
}

int ccd_debug_chain_groups(const double* est, const int32_t* inst, int n, int n_conc, int n_cu, int32_t* cg) {
    if (!est || !inst || !cg || n < 0 || n_conc < 1 || n_cu < 8) return CCD_ERR_ARG;
    plan_chain_groups(est, inst, n, n_conc, n_cu, cg);
    int groups = 0;
    for (int i = 0; i < n; ++i) groups = std::max(groups, cg[i] + 1);
    return groups;
}

int ccd_concurrent_streams(int device) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return CCD_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    DeviceShared* sh = nullptr;
    const int rc = device_shared(device, &sh);
    return rc < 0 ? rc : sh->n_conc;
}

int ccd_batch_launch_ms(ccd_batch* b, float* ms, int* n_streams, int cap) {
    if (!b || !ms || cap < 0) return CCD_ERR_ARG;
    if (!b->opt_time_launches) return 0;
    // launch order = longest expected chains first (launch_entropy_groups); the generic launch last
    std::vector<int> order(b->pipe_groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->pipe_groups[x].est > b->pipe_groups[y].est; });
    const int n = std::min(b->n_timed, cap);
    for (int k = 0; k < n; ++k) {
        if (hipEventSynchronize(b->lt1[k]) != hipSuccess || hipEventElapsedTime(&ms[k], b->lt0[k], b->lt1[k]) != hipSuccess) return CCD_ERR_HIP;
        if (n_streams) n_streams[k] = k < static_cast<int>(order.size()) ? b->pipe_groups[order[k]].n : b->n_generic;
    }
    return n;
}

int ccd_batch_entropy_launches(const ccd_batch* b) {
    if (!b) return CCD_ERR_ARG;
    return static_cast<int>(b->pipe_groups.size()) + (b->n_generic > 0 ? 1 : 0);
}

int ccd_batch_slot_kernels(const ccd_batch* b, int slot) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    const Slot& s = *b->slots[slot];
    return (s.use_pipe ? 1 : 0) | (s.use_fused_syn ? 2 : 0) | (s.use_fused_dec ? 4 : 0) | (s.use_mfma ? 8 : 0) | (s.use_dyn ? 16 : 0) | (s.fixed_shape ? 32 : 0) | (s.fdec_pre ? 64 : 0) | (s.float_finite ? 0 : 128);
}

const float* ccd_batch_output(const ccd_batch* b, int slot) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size())) return nullptr;
    const Slot& s = *b->slots[slot];
    if (s.use_fused_dec && !s.fdec.out) return nullptr;  // CCD_OPT_KEEP_FLOAT = 0: integer samples only
    return s.d_out;
}
const float* ccd_batch_dense(const ccd_batch* b, int slot) {
    // the dense stack only exists on the unfused path (ccd_batch_set_option(b, CCD_OPT_FUSED_DEC, 0) before adding the slot)
    return (b && slot >= 0 && slot < static_cast<int>(b->slots.size()) && !b->slots[slot]->use_fused_dec) ? b->slots[slot]->d_dense : nullptr;
}
int ccd_batch_set_option(ccd_batch* b, int option, int value) {
    if (!b) return CCD_ERR_ARG;
    switch (option) {
        case CCD_OPT_FUSED_DEC:
            if (value < 0 || value > 2) return CCD_ERR_ARG;
            b->opt_fused_dec = value; return CCD_OK;
        case CCD_OPT_KEEP_FLOAT: b->opt_keep_float = value; return CCD_OK;
        case CCD_OPT_MFMA_ARM: b->opt_mfma_arm = value; return CCD_OK;
        case CCD_OPT_RANGE_BITS: b->opt_range_bits = value; return CCD_OK;
        case CCD_OPT_TIME_LAUNCHES: b->opt_time_launches = value ? 1 : 0; return CCD_OK;
        case CCD_OPT_OVERLAP:
            // (decides how the launch tables are grouped: a change re-builds them at the next run)
            if (b->opt_overlap != (value ? 1 : 0)) { b->opt_overlap = value ? 1 : 0; b->regroup = true; }
            return CCD_OK;
        default: return CCD_ERR_ARG;
    }
}
const int8_t* ccd_batch_latent(const ccd_batch* b, int slot, int grid) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size())) return nullptr;
    const Slot& s = *b->slots[slot];
    return (grid >= 0 && grid < s.hdr.n_grids) ? s.ep.latent[grid] : nullptr;
}
const void* ccd_batch_plane(const ccd_batch* b, int slot, int plane, int* h, int* w) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size()) || plane < 0 || plane > 2) return nullptr;
    const Slot& s = *b->slots[slot];
    if (h) *h = s.plane_h[plane];
    if (w) *w = s.plane_w[plane];
    return s.d_plane[plane];
}

// Results of a slot whose entropy stage reported an error (corrupt / truncated payload) are whatever the arena held: never
// handed out.  (Status is known after ccd_batch_wait; before it the copy reflects the caller's own ordering.)
static int slot_failed(const ccd_batch* b, int slot) {
    return (b && slot >= 0 && slot < static_cast<int>(b->slots.size()) && b->slots[slot]->status < 0) ? b->slots[slot]->status : 0;
}

static int copy_out(ccd_batch* b, const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst) return CCD_ERR_ARG;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    b->note_stream(st);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CCD_OK;
}

int ccd_batch_copy_latent(ccd_batch* b, int slot, int grid, int8_t* host, void* stream) {
    if (const int failed = slot_failed(b, slot)) return failed;
    const int8_t* p = ccd_batch_latent(b, slot, grid);
    if (!p) return CCD_ERR_ARG;
    const ccd_cc_header& h = b->slots[slot]->hdr;
    return copy_out(b, p, host, static_cast<size_t>(h.grid_h[grid]) * h.grid_w[grid], stream);
}
int ccd_batch_copy_plane(ccd_batch* b, int slot, int plane, void* host, void* stream) {
    if (const int failed = slot_failed(b, slot)) return failed;
    int ph = 0, pw = 0;
    const void* p = ccd_batch_plane(b, slot, plane, &ph, &pw);
    if (!p) return CCD_ERR_ARG;
    return copy_out(b, p, host, static_cast<size_t>(ph) * pw * (b->slots[slot]->bitdepth == 8 ? 1 : 2), stream);
}
int ccd_batch_planes_layout(const ccd_batch* b, int slot, size_t* total_bytes, size_t* off3) {
    if (!b || slot < 0 || slot >= static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    const Slot& s = *b->slots[slot];
    if (!s.d_plane[0]) return CCD_ERR_ARG;  // added with bitdepth = 0
    if (total_bytes) *total_bytes = s.planes_bytes;
    if (off3) for (int p = 0; p < 3; ++p) off3[p] = s.plane_off[p];
    return CCD_OK;
}

int ccd_batch_copy_planes_async(ccd_batch* b, int first_slot, int n_slots, void* const* host_blocks, void* stream) {
    if (!b || !host_blocks || first_slot < 0 || n_slots < 0 || first_slot + n_slots > static_cast<int>(b->slots.size())) return CCD_ERR_ARG;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    b->note_stream(st);  // the copies read the arenas: drained before the blocks are recycled (ccd_batch_destroy)
    int rc = CCD_OK;
    for (int i = 0; i < n_slots; ++i) {
        const Slot& s = *b->slots[first_slot + i];
        if (!s.d_plane[0] || !host_blocks[i]) { if (rc == CCD_OK) rc = CCD_ERR_ARG; continue; }
        if (s.status < 0) { if (rc == CCD_OK) rc = s.status; continue; }  // a failed slot's planes are never handed out
        if (hipMemcpyAsync(host_blocks[i], s.d_plane[0], s.planes_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return CCD_ERR_HIP;
    }
    return rc;
}

int ccd_batch_copy_output(ccd_batch* b, int slot, float* host, void* stream) {
    if (const int failed = slot_failed(b, slot)) return failed;
    const float* p = ccd_batch_output(b, slot);
    if (!p) return CCD_ERR_ARG;
    const ccd_cc_header& h = b->slots[slot]->hdr;
    return copy_out(b, p, host, static_cast<size_t>(h.out_channels) * h.img_size[0] * h.img_size[1] * 4, stream);
}
int ccd_batch_copy_dense(ccd_batch* b, int slot, float* host, void* stream) {
    if (const int failed = slot_failed(b, slot)) return failed;
    const float* p = ccd_batch_dense(b, slot);
    if (!p) return CCD_ERR_ARG;
    const Slot& s = *b->slots[slot];
    return copy_out(b, p, host, static_cast<size_t>(s.dense_c) * s.dense_h * s.dense_w * 4, stream);
}

// -------------------------------------------------------------------------------------------------
// One-shot conveniences
// -------------------------------------------------------------------------------------------------
int ccd_decode_coolchic(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn,
                        const uint8_t* bytes_latent, size_t n_lat, int device, void* stream, float* out,
                        int out_on_device) {
    if (!out) return CCD_ERR_ARG;
    if (!bytes_latent) return CCD_ERR_ARG;  // coolchic.py:46-51
    ccd_batch* b = nullptr;
    int rc = ccd_batch_create(device, &b);
    if (rc < 0) return rc;
    rc = ccd_batch_add(b, cc_header, n_hdr, bytes_nn, n_nn, bytes_latent, n_lat, 0, 0);
    if (rc >= 0) rc = ccd_batch_run(b, stream);
    if (rc >= 0) rc = ccd_batch_wait(b, stream);
    if (rc >= 0) {
        const ccd_cc_header& h = b->slots[0]->hdr;
        const size_t bytes = static_cast<size_t>(h.out_channels) * h.img_size[0] * h.img_size[1] * 4;
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (hipMemcpyAsync(out, b->slots[0]->d_out, bytes, out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = CCD_ERR_HIP;
    }
    ccd_batch_destroy(b);
    return rc;
}

// ccd_decode_video hands out every plane of every frame inside ONE pinned host block (filled by one copy); its handle rides
// in a hidden ccd_frame behind the last one.
void ccd_video_free(ccd_video* v) {
    if (!v || !v->frames) return;
    Block* blk = reinterpret_cast<Block*>(v->frames[v->n_frames].plane[0]);
    if (blk) { blk->drop(); delete blk; }
    std::free(v->frames);
    v->frames = nullptr; v->n_frames = 0;
}

// decode.py:156-206 for one P / B frame on `st`, no allocation, no wait: tmp = 9 h w floats (two references and the result as
// 4:4:4 f32 planes).
static int inter_reconstruct_on(hipStream_t st, float* tmp, int frame_type, int h, int w, int bitdepth, int frame_data_type,
                                const float* residue, const float* motion, const void* const* ref0_planes, const void* const* ref1_planes,
                                const int32_t* global_flow, int warp_filter_size, void* const* out_planes, const void* coef = nullptr) {
    float* ref0 = tmp;
    float* ref1 = tmp + static_cast<size_t>(3) * h * w;
    float* out = tmp + static_cast<size_t>(6) * h * w;
    int gf[4] = {global_flow[0], global_flow[1], frame_type == 2 ? global_flow[2] : 0, frame_type == 2 ? global_flow[3] : 0};
    if (launch_planes_to_444(ref0_planes[0], ref0_planes[1], ref0_planes[2], ref0, h, w, bitdepth, frame_data_type, st) != hipSuccess) return CCD_ERR_HIP;
    if (frame_type == 2 &&
        launch_planes_to_444(ref1_planes[0], ref1_planes[1], ref1_planes[2], ref1, h, w, bitdepth, frame_data_type, st) != hipSuccess) return CCD_ERR_HIP;
    if (coef) {  // the sinc-8 coefficients were computed ahead of the references (ccd_decode_video): gather only
        if (launch_inter_apply8(frame_type, h, w, gf, residue, motion, ref0, frame_type == 2 ? ref1 : ref0, coef, out, st) != hipSuccess) return CCD_ERR_HIP;
    } else if (launch_inter_recon(frame_type, h, w, warp_filter_size, gf, residue, motion, ref0, frame_type == 2 ? ref1 : ref0, out, st) != hipSuccess) return CCD_ERR_HIP;
    if (launch_planes(out, out_planes[0], out_planes[1], out_planes[2], h, w, bitdepth, frame_data_type, st) != hipSuccess) return CCD_ERR_HIP;
    return CCD_OK;
}

int ccd_inter_reconstruct(int device, void* stream, int frame_type, int h, int w, int bitdepth, int frame_data_type,
                          const float* residue, const float* motion, const void* const* ref0_planes,
                          const void* const* ref1_planes, const int32_t* global_flow, int warp_filter_size,
                          void* const* out_planes) {
    if ((frame_type != 1 && frame_type != 2) || !residue || !motion || !ref0_planes || !global_flow || !out_planes ||
        (frame_type == 2 && !ref1_planes) || h <= 0 || w <= 0 || bitdepth < 8 || bitdepth > 16)
        return CCD_ERR_ARG;
    // 2 / 4 taps = grid_sample bilinear / bicubic, 6.. = sinc (warp.py:49-56); odd or < 2 fails the reference's asserts (warp.py:41-47)
    if (warp_filter_size < 2 || warp_filter_size > 16 || (warp_filter_size & 1)) return CCD_ERR_VALUE;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Block tmp;
    if (!tmp.get(device, BlockPool::kDevice, static_cast<size_t>(9) * h * w * sizeof(float))) return CCD_ERR_NOMEM;
    int rc = inter_reconstruct_on(st, tmp.as<float>(), frame_type, h, w, bitdepth, frame_data_type, residue, motion, ref0_planes, ref1_planes,
                                  global_flow, warp_filter_size, out_planes);
    if (hipStreamSynchronize(st) != hipSuccess) rc = CCD_ERR_HIP;
    tmp.drop();
    return rc;
}

int ccd_decode_video(const uint8_t* bs, size_t n, int device, ccd_video* v) {
    if (!bs || !v) return CCD_ERR_ARG;
    v->n_frames = 0; v->frames = nullptr;
    // CCD_VIDEO_TIMING=1: host wall clock of the call's phases on stderr (tools/prof_gop.py)
    const bool timing = std::getenv("CCD_VIDEO_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (timing) std::fprintf(stderr, "[ccd_decode_video] %-28s %8.2f ms\n", what,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    std::unique_ptr<ccd_video_header> vh(new (std::nothrow) ccd_video_header());
    if (!vh) return CCD_ERR_NOMEM;
    int used = read_video_header(bs, n, vh.get());
    if (used < 0) return used;
    size_t pos = static_cast<size_t>(used);
    const int n_frames = vh->n_frames;
    ccd_batch* b = nullptr;
    int rc = ccd_batch_create(device, &b);
    if (rc < 0) return rc;
    // Every cool-chic of every frame is independent: all of them go into ONE batch and decode concurrently; the cheap
    // reconstruction then walks the frames in coding order (decode.py:67-81).
    std::vector<ccd_frame_header> fhs(n_frames);
    std::vector<int> first_slot(n_frames, 0);
    // decode.py:52-75: the coding order and every frame's references come from the VIDEO header's coding structure
    std::vector<CodedFrame> cs;
    rc = coding_structure(*vh, cs);
    for (int f = 0; f < n_frames && rc >= 0; ++f) {
        used = read_frame_header(bs + pos, n - pos, &fhs[f]);
        if (used < 0) { rc = used; break; }
        {
            // decode.py:67-75 takes the display index and the references of the frame at this coding index from the STRUCTURE and
            // never reads those fields of the frame header; the header's frame_type decides how many cool-chics follow and how the
            // frame is reconstructed (decode.py:119-128, 156-189), whatever the structure calls the frame: a header "I" at a P / B
            // position decodes as plain intra (decode_frame ignores reference_frames), a header "P" at a B position predicts from
            // the structure's first reference only (apply_global_translation zips references with flows).  Rejected is only what
            // the reference raises on: a header type that needs MORE references than the structure gives (raw_references[0] /
            // shifted_ref[1]: IndexError)
            const CodedFrame& want = cs[f];
            if (fhs[f].frame_type > want.n_refs) { rc = CCD_ERR_VALUE; break; }  // I / P / B = 0 / 1 / 2 = references needed
            fhs[f].display_index = want.display_order;
            fhs[f].n_refs = fhs[f].frame_type;
            for (int k = 0; k < fhs[f].n_refs; ++k) fhs[f].index_references[k] = want.refs[k];
        }
        pos += static_cast<size_t>(used);
        first_slot[f] = ccd_batch_size(b);
        const int n_cc = fhs[f].frame_type == 0 ? 1 : 2;  // residue (+ motion), decode.py:126-128
        for (int c = 0; c < n_cc && rc >= 0; ++c) {
            ccd_cc_header ch;
            used = read_cc_header(bs + pos, n - pos, &ch);
            if (used < 0) { rc = used; break; }
            const uint8_t* hdr = bs + pos;
            pos += static_cast<size_t>(used);
            if (pos + static_cast<size_t>(ch.nn_n_bytes) + static_cast<size_t>(ch.n_bytes_latent) > n) { rc = CCD_ERR_TRUNCATED; break; }
            const bool intra = fhs[f].frame_type == 0;
            rc = ccd_batch_add(b, hdr, static_cast<size_t>(used), bs + pos, ch.nn_n_bytes, bs + pos + ch.nn_n_bytes, ch.n_bytes_latent,
                               intra ? fhs[f].bitdepth : 0, fhs[f].frame_data_type);
            pos += static_cast<size_t>(ch.nn_n_bytes) + static_cast<size_t>(ch.n_bytes_latent);
        }
    }
    mark("parsed + added");
    if (rc >= 0) rc = ccd_batch_run(b, nullptr);
    mark("launched");
    // ---- r06, OFF by default (CCD_VIDEO_COEF_MB = scratch budget in MB): the warp coefficients of the inter frames ahead of their
    // references.  The cool-chics of a hierarchical GOP's B frames are decoded at about half of the I frames' chains, and every
    // reconstruction then waits for the I frames; a frame's sinc coefficients (f64 sin / cos) only need its flows.  With a budget they
    // are computed on the copy stream as soon as the launch of the frame's motion cool-chic is done (ccd_batch::lg_done), 64 B per pixel
    // and reference, and the reconstruction only gathers.  Same bits either way (test_video_warp_coefficients_ahead_of_the_references).
    // Measured on the 33-frame 1080p GOP (profiles/r06/gop_timing_coefficients_ahead.txt): 174.0 against 175.1 ms for 8.2 GB of
    // scratch - the gather of the 2 x 64 x 3 taps is 0.26 of the 0.34 ms a frame takes, the f64 work only the rest.  Not worth the
    // memory by default.
    std::vector<Block> coef(n_frames);
    std::vector<hipEvent_t> coef_done(n_frames, nullptr);
    if (rc >= 0 && b->lg_valid) {
        size_t budget = 0;
        if (const char* e = std::getenv("CCD_VIDEO_COEF_MB")) budget = static_cast<size_t>(std::max(0, std::atoi(e)));
        budget <<= 20;
        size_t spent = 0;
        for (int f = 0; f < n_frames; ++f) {
            const ccd_frame_header& fh = fhs[f];
            if (fh.frame_type == 0 || fh.warp_filter_size != 8) continue;
            const Slot& s0 = *b->slots[first_slot[f]];
            const Slot& s1 = *b->slots[first_slot[f] + 1];
            const int h = s0.hdr.img_size[0], w = s0.hdr.img_size[1];
            if (s1.hdr.out_channels < (fh.frame_type == 1 ? 2 : 4) || s1.hdr.img_size[0] != h || s1.hdr.img_size[1] != w) continue;  // (rejected below)
            if (s1.lg < 0 || s1.fl != s1.lg || static_cast<size_t>(s1.lg) >= b->lg_done.size() || !b->lg_done[s1.lg] || !s1.use_fused_dec || s1.cr)
                continue;  // its output is only complete behind the join
            const size_t bytes = inter_coef_bytes(fh.frame_type, h, w);
            if (spent + bytes > budget) break;
            if (!coef[f].get(device, BlockPool::kDevice, bytes)) break;
            spent += bytes;
            if (hipStreamWaitEvent(b->up_stream, b->lg_done[s1.lg], 0) != hipSuccess ||
                launch_inter_coef8(fh.frame_type, h, w, s1.d_out, coef[f].p, b->up_stream) != hipSuccess ||
                hipEventCreateWithFlags(&coef_done[f], hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(coef_done[f], b->up_stream) != hipSuccess) { rc = CCD_ERR_HIP; break; }
        }
    }
    if (rc >= 0) rc = ccd_batch_wait(b, nullptr);
    mark("cool-chics decoded");
    // ---- frame reconstruction in coding order; device planes of every decoded frame are kept for references.  r06: a frame's
    // planes start their way to the host (u16 widening + one device -> host copy per frame on the library's upload stream, behind
    // an event) as soon as the frame is reconstructed, while the following frames are still being warped: the 200 MB of a 33-frame
    // 1080p GOP used to cross PCIe after the last frame (3.6 ms of a 190 ms call, profiles/r06/gop_timing_before.txt).
    struct DevFrame { void* plane[3] = {nullptr, nullptr, nullptr}; int h = 0, w = 0, ch = 0, cw = 0, bitdepth = 0, fdt = 0; bool seen = false; Block own; };
    std::vector<DevFrame> dev(n_frames);  // by display index
    Block tmp;  // two references and the result as 4:4:4 f32 planes, reused by every inter frame (one stream: ordered)
    size_t tmp_elems = 0;
    // geometry of every frame is known from the headers: the host block and the u16 staging block are laid out up front
    Block wide;
    Block* host = nullptr;
    std::vector<size_t> off(static_cast<size_t>(n_frames) * 3, 0);
    size_t total = 0;
    hipStream_t copy_st = b->up_stream;
    hipEvent_t frame_done = nullptr;
    if (rc >= 0) {
        for (int f = 0; f < n_frames && rc >= 0; ++f) {  // sizes by display index
            const ccd_frame_header& fh = fhs[f];
            if (fh.display_index < 0 || fh.display_index >= n_frames) { rc = CCD_ERR_VALUE; break; }
            DevFrame& d = dev[fh.display_index];
            if (d.seen) { rc = CCD_ERR_VALUE; break; }  // two frames with one display index: the second would overwrite the first
            d.seen = true;
            const Slot& s0 = *b->slots[first_slot[f]];
            d.h = s0.hdr.img_size[0]; d.w = s0.hdr.img_size[1]; d.bitdepth = fh.bitdepth; d.fdt = fh.frame_data_type;
            // 4:2:0 needs even sizes: F.avg_pool2d(2) drops the odd row / column and write_yuv's chroma planes are h/2 x w/2,
            // while the reference's 4:4:4 round trip of such a frame (yuv.py:303-316) no longer matches the luma size
            if (fh.frame_data_type == 1 && ((d.h | d.w) & 1)) { rc = CCD_ERR_VALUE; break; }
            d.ch = fh.frame_data_type == 1 ? d.h / 2 : d.h; d.cw = fh.frame_data_type == 1 ? d.w / 2 : d.w;
        }
        // every display index must have been produced (a gap would leave a frame without planes)
        for (int i = 0; i < n_frames && rc >= 0; ++i) if (!dev[i].seen) rc = CCD_ERR_VALUE;
    }
    if (rc >= 0) {
        for (int i = 0; i < n_frames; ++i)
            for (int p = 0; p < 3; ++p) {
                off[static_cast<size_t>(i) * 3 + p] = total;
                total += ((p == 0 ? static_cast<size_t>(dev[i].h) * dev[i].w : static_cast<size_t>(dev[i].ch) * dev[i].cw) * 2 + 63) & ~size_t{63};
            }
        host = new (std::nothrow) Block();
        v->frames = static_cast<ccd_frame*>(std::calloc(static_cast<size_t>(n_frames) + 1, sizeof(ccd_frame)));
        if (!host || !v->frames || !wide.get(device, BlockPool::kDevice, std::max<size_t>(total, 64)) ||
            !host->get(device, BlockPool::kPinned, std::max<size_t>(total, 64)))
            rc = CCD_ERR_NOMEM;
        if (rc >= 0 && hipEventCreateWithFlags(&frame_done, hipEventDisableTiming) != hipSuccess) rc = CCD_ERR_HIP;
    }
    // planes of display index di: widened to u16 and copied to the host on the copy stream, behind everything enqueued so far
    auto send_frame = [&](int di) -> int {
        const DevFrame& d = dev[di];
        HIP_TRY(hipEventRecord(frame_done, nullptr));
        HIP_TRY(hipStreamWaitEvent(copy_st, frame_done, 0));
        for (int p = 0; p < 3; ++p) {
            const size_t px = p == 0 ? static_cast<size_t>(d.h) * d.w : static_cast<size_t>(d.ch) * d.cw;
            uint16_t* dst = reinterpret_cast<uint16_t*>(wide.as<char>() + off[static_cast<size_t>(di) * 3 + p]);
            const hipError_t e = d.bitdepth == 8 ? launch_widen_u8(static_cast<const uint8_t*>(d.plane[p]), dst, px, copy_st)
                                                 : hipMemcpyAsync(dst, d.plane[p], px * 2, hipMemcpyDeviceToDevice, copy_st);
            if (e != hipSuccess) return CCD_ERR_HIP;
        }
        const size_t o0 = off[static_cast<size_t>(di) * 3], o1 = di + 1 < n_frames ? off[static_cast<size_t>(di + 1) * 3] : total;
        HIP_TRY(hipMemcpyAsync(host->as<char>() + o0, wide.as<char>() + o0, o1 - o0, hipMemcpyDeviceToHost, copy_st));
        return CCD_OK;
    };
    for (int f = 0; f < n_frames && rc >= 0; ++f) {
        const ccd_frame_header& fh = fhs[f];
        DevFrame& d = dev[fh.display_index];
        const Slot& s0 = *b->slots[first_slot[f]];
        if (fh.frame_type == 0) {
            if (s0.hdr.out_channels < 3) { rc = CCD_ERR_VALUE; break; }
            for (int p = 0; p < 3; ++p) d.plane[p] = s0.d_plane[p];
        } else {
            const Slot& s1 = *b->slots[first_slot[f] + 1];
            const int need_res = fh.frame_type == 1 ? 4 : 5, need_mot = fh.frame_type == 1 ? 2 : 4;
            if (s0.hdr.out_channels < need_res || s1.hdr.out_channels < need_mot || s1.hdr.img_size[0] != d.h || s1.hdr.img_size[1] != d.w) { rc = CCD_ERR_VALUE; break; }
            if (fh.warp_filter_size < 2 || fh.warp_filter_size > 16 || (fh.warp_filter_size & 1)) { rc = CCD_ERR_VALUE; break; }  // warp.py:41-56
            const void* refs[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
            for (int k = 0; k < fh.n_refs && rc >= 0; ++k) {
                const int ri = fh.index_references[k];
                // a reference must be a decoded frame of the same geometry AND sample layout: its planes are read with this
                // frame's layout (a 4:2:0 reference has quarter-size chroma planes)
                if (ri < 0 || ri >= n_frames || !dev[ri].plane[0] || dev[ri].h != d.h || dev[ri].w != d.w || dev[ri].bitdepth != d.bitdepth ||
                    dev[ri].fdt != d.fdt) { rc = CCD_ERR_VALUE; break; }
                for (int p = 0; p < 3; ++p) refs[k][p] = dev[ri].plane[p];
            }
            if (rc < 0) break;
            const size_t sb = d.bitdepth == 8 ? 1 : 2;
            const size_t luma = (static_cast<size_t>(d.h) * d.w * sb + 16 + 255) & ~size_t{255};
            const size_t chroma = (static_cast<size_t>(d.ch) * d.cw * sb + 16 + 255) & ~size_t{255};
            if (!d.own.get(device, BlockPool::kDevice, luma + 2 * chroma)) { rc = CCD_ERR_NOMEM; break; }
            d.plane[0] = d.own.as<char>(); d.plane[1] = d.own.as<char>() + luma; d.plane[2] = d.own.as<char>() + luma + chroma;
            const size_t need = static_cast<size_t>(9) * d.h * d.w;
            if (need > tmp_elems) {
                if (tmp.p && hipStreamSynchronize(nullptr) != hipSuccess) { rc = CCD_ERR_HIP; break; }  // frames in flight still use the smaller one
                if (!tmp.get(device, BlockPool::kDevice, need * sizeof(float))) { rc = CCD_ERR_NOMEM; break; }
                tmp_elems = need;
            }
            if (coef_done[f] && hipStreamWaitEvent(nullptr, coef_done[f], 0) != hipSuccess) { rc = CCD_ERR_HIP; break; }
            rc = inter_reconstruct_on(nullptr, tmp.as<float>(), fh.frame_type, d.h, d.w, d.bitdepth, d.fdt, s0.d_out, s1.d_out, refs[0],
                                      fh.frame_type == 2 ? refs[1] : nullptr, fh.global_flow, fh.warp_filter_size, d.plane,
                                      coef_done[f] ? coef[f].p : nullptr);
        }
        if (rc >= 0) rc = send_frame(fh.display_index);
    }
    if (timing) { (void)hipStreamSynchronize(nullptr); mark("reconstructed"); }
    if (rc >= 0 && hipStreamSynchronize(copy_st) != hipSuccess) rc = CCD_ERR_HIP;
    mark("planes on the host");
    if (rc >= 0) {
        v->n_frames = n_frames;
        v->frames[n_frames].plane[0] = reinterpret_cast<uint16_t*>(host);  // hidden: the block every plane points into (ccd_video_free)
        for (int f = 0; f < n_frames; ++f) {
            const int di = fhs[f].display_index;
            const DevFrame& d = dev[di];
            ccd_frame& fr = v->frames[di];
            fr.display_index = di; fr.frame_type = fhs[f].frame_type; fr.frame_data_type = d.fdt; fr.bitdepth = d.bitdepth;
            fr.h = d.h; fr.w = d.w; fr.ch = d.ch; fr.cw = d.cw;
            for (int p = 0; p < 3; ++p) fr.plane[p] = reinterpret_cast<uint16_t*>(host->as<char>() + off[static_cast<size_t>(di) * 3 + p]);
        }
    } else {
        if (host) { host->drop(); delete host; }
        std::free(v->frames);
        v->frames = nullptr; v->n_frames = 0;
    }
    (void)hipStreamSynchronize(copy_st);
    if (frame_done) (void)hipEventDestroy(frame_done);
    for (hipEvent_t e : coef_done) if (e) (void)hipEventDestroy(e);
    for (auto& c : coef) c.drop();
    (void)hipStreamSynchronize(nullptr);  // nothing may still read the blocks that go back to the pool
    for (auto& d : dev) d.own.drop();
    tmp.drop(); wide.drop();
    ccd_batch_destroy(b);
    mark("batch destroyed");
    return rc < 0 ? rc : CCD_OK;
}

int ccd_network_fits_fast_path(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn) {
    if (!cc_header || !bytes_nn) return CCD_ERR_ARG;
    std::unique_ptr<ccd_cc_header> h(new (std::nothrow) ccd_cc_header());
    if (!h) return CCD_ERR_NOMEM;
    int rc = read_cc_header(cc_header, n_hdr, h.get());
    if (rc < 0) return rc;
    Network net;
    rc = decode_network(*h, bytes_nn, n_nn, net);
    if (rc < 0) return rc;
    int max_w = 0;
    for (int g = 0; g < h->n_grids; ++g) max_w = std::max(max_w, static_cast<int>(h->grid_w[g]));
    return entropy_pipe_supports(h->total_context_arm, h->n_hidden_layers_arm + 1, (net.arm.w32 && net.feat_i32 && !net.arm.dyn_act) ? 1 : 0, max_w) ? 1 : 0;
}

int ccd_network_kernel_class(const uint8_t* cc_header, size_t n_hdr, const uint8_t* bytes_nn, size_t n_nn) {
    if (!cc_header || !bytes_nn) return CCD_ERR_ARG;
    std::unique_ptr<ccd_cc_header> h(new (std::nothrow) ccd_cc_header());
    if (!h) return CCD_ERR_NOMEM;
    int rc = read_cc_header(cc_header, n_hdr, h.get());
    if (rc < 0) return rc;
    Network net;
    rc = decode_network(*h, bytes_nn, n_nn, net);
    if (rc < 0) return rc;
    int max_w = 0;
    for (int g = 0; g < h->n_grids; ++g) max_w = std::max(max_w, static_cast<int>(h->grid_w[g]));
    const bool pipe = entropy_pipe_supports(h->total_context_arm, h->n_hidden_layers_arm + 1, (net.arm.w32 && net.feat_i32 && !net.arm.dyn_act) ? 1 : 0, max_w);
    int n_levels = 0;
    for (int g = 0; g < h->n_grids; ++g) n_levels += h->is_hyperlatent[g] ? 0 : 1;
    const bool finite = float_path_stays_finite(net, n_levels, h->flag_common_randomness ? n_levels : 0);
    return (pipe ? 1 : 0) | (pipe && net.arm.dyn_feat ? 16 : 0) | (finite ? 0 : 128) | (((h->total_context_arm + 3) / 4 & 15) << 8) |
           (((h->n_hidden_layers_arm + 1) & 15) << 12);
}

int ccd_debug_fd_profile(uint64_t* out16, int reset) {
    return out16 ? fused_dec_profile(reinterpret_cast<unsigned long long*>(out16), reset) : CCD_ERR_ARG;
}

int ccd_debug_laplace_bounds(int device, const int32_t* mu_idx, const int32_t* scale_idx, const int32_t* s, int64_t n,
                             uint32_t* left, uint32_t* right) {
    if (n <= 0) return CCD_OK;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return CCD_ERR_HIP;
    HIP_TRY(hipSetDevice(device));
    int32_t *d_mu = nullptr, *d_sc = nullptr, *d_s = nullptr;
    uint32_t *d_l = nullptr, *d_r = nullptr;
    float* d_tab = nullptr;
    int rc = CCD_OK;
    const size_t nb = static_cast<size_t>(n) * 4;
    if (hipMalloc(&d_mu, nb) != hipSuccess || hipMalloc(&d_sc, nb) != hipSuccess || hipMalloc(&d_s, nb) != hipSuccess ||
        hipMalloc(&d_l, nb) != hipSuccess || hipMalloc(&d_r, nb) != hipSuccess || hipMalloc(&d_tab, sizeof(kScaleBits)) != hipSuccess)
        rc = CCD_ERR_NOMEM;
    if (rc == CCD_OK &&
        (hipMemcpy(d_mu, mu_idx, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_sc, scale_idx, nb, hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(d_s, s, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_tab, kScaleBits, sizeof(kScaleBits), hipMemcpyHostToDevice) != hipSuccess))
        rc = CCD_ERR_HIP;
    if (rc == CCD_OK && launch_laplace_bounds(d_mu, d_sc, d_s, d_tab, n, d_l, d_r, nullptr) != hipSuccess) rc = CCD_ERR_HIP;
    if (rc == CCD_OK && (hipMemcpy(left, d_l, nb, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(right, d_r, nb, hipMemcpyDeviceToHost) != hipSuccess))
        rc = CCD_ERR_HIP;
    (void)hipFree(d_mu); (void)hipFree(d_sc); (void)hipFree(d_s); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_tab);
    return rc;
}

int ccd_debug_laplace_sweep(int device, int which, int scale_first, int n_scales, uint32_t* out) {
    if (!out || scale_first < 0 || n_scales <= 0 || scale_first + n_scales > kNumScale || (which != 0 && which != 1)) return CCD_ERR_ARG;
    ccd_batch* b = nullptr;  // owns the two Laplace-scale tables on the device
    int rc = ccd_batch_create(device, &b);
    if (rc < 0) return rc;
    const size_t bytes = static_cast<size_t>(n_scales) * kNumMu * 127 * sizeof(uint32_t);
    uint32_t* d_out = nullptr;
    if (hipMalloc(&d_out, bytes) != hipSuccess) rc = CCD_ERR_NOMEM;
    if (rc == CCD_OK) {
        const hipError_t e = which == 0 ? launch_laplace_sweep_pipe(b->d_scale_table, b->d_rcp_table, scale_first, n_scales, d_out, nullptr)
                                        : launch_laplace_sweep_generic(b->d_scale_table, scale_first, n_scales, d_out, nullptr);
        if (e != hipSuccess || hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = CCD_ERR_HIP;
    }
    if (d_out) (void)hipFree(d_out);
    ccd_batch_destroy(b);
    return rc;
}

}  // extern "C"
