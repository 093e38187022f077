// dspmap_dist.hip -- the Z-slab frame driven from C++ (north star: "Host code stays C++ ... the voxel grid
// shards by Z-slab across the 8 GPUs of one node with an RCCL exchange over xGMI for particles that cross slab boundaries").
//
// One process per GPU; rank g owns the layers [z_lo, z_hi) of dspmap_config.  dspmap_mgpu_update() runs one whole frame:
// every collective is enqueued on the library's stream between the phases of dspmap_mgpu.hip, and NO host
// synchronisation sits inside a frame:
//   begin (binning, prediction)          -> export of the particles that left the slab, both faces
//   neighbour exchange                   -> FIXED-SIZE ncclSend / ncclRecv pairs with rank +- 1 (ncclGroup): record 0 of a
//                                           buffer is its header (the number of records that follow), so the receiver
//                                           reads the count on the device; the size all ranks use is derived from the
//                                           largest export of the previous frame (+50 % + 1024), which rides on the
//                                           n_static all-reduce as one extra MAX slot and reaches the host a frame later;
//                                           a particle that has to cross more than one slab (|dz| > slab height) is
//                                           forwarded in further rounds (their number follows from the pose, equal on
//                                           every rank)
//   placement, Ck pass                   -> ncclAllReduce(SUM, int64) of the fixed-point Ck sums (<= 358 kB)
//   weight update, Dempster-Shafer split -> ncclAllReduce(MAX, int32) of n_static per birth source (+ the slot above)
//   births, resampling, rollout
// RCCL is loaded with dlopen at the first dspmap_mgpu_comm_init* (libdspmap_hip.so has no link-time dependency on it; a
// single-GPU process never touches it).  The same frame driver also runs over several slabs inside ONE process
// (dspmap_mgpu_group_*: device-to-device copies and small reduction kernels in place of the collectives): that is how
// the tests prove, on one GPU, that a map sharded by this driver is bit-identical to the unsharded one.
#include "dspmap_internal.h"
#include "dspmap_device.h"

#include <dlfcn.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    // a copy the process already holds (e.g. the one torch.distributed loaded) is reused; otherwise ROCm's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (api.lib) break; }
    if (!api.lib) for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
    if (!api.lib) return &api;
#define LOAD(field, sym) *(void**)(&api.field) = dlsym(api.lib, sym)
    LOAD(GetUniqueId, "ncclGetUniqueId"); LOAD(CommInitRank, "ncclCommInitRank"); LOAD(CommDestroy, "ncclCommDestroy");
    LOAD(AllReduce, "ncclAllReduce"); LOAD(Send, "ncclSend"); LOAD(Recv, "ncclRecv");
    LOAD(GroupStart, "ncclGroupStart"); LOAD(GroupEnd, "ncclGroupEnd"); LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
    return &api;
}
}  // namespace

// per-handle state of the C++ driver
struct dspmap_dist {
    int world = 1, rank = 0;
    ncclComm_t comm = nullptr;          // RCCL transport (nullptr inside a one-process group)
    float* buf[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // [dir 0 up / 1 down][send, recv, forward]: (1 + xcap) records of 8 floats
    int xcap = 0;                       // records the exchange buffers hold
    int xsend = 0;                      // records this frame's messages carry (the same on every rank)
    double xratio = 0.75;               // exports of a frame / (cells of one layer x |dz| / res): what fraction of "a layer's slots times the vertical
                                        // step in voxels" really crossed a face in earlier frames (0.75: nothing known -- more than a saturated map's 0.5)
    int* cnt2 = nullptr;                // device: export counts {up, down}
    // what a frame learns for later ones -- its largest export over all ranks, the longest pyramid list -- reaches the host through
    // pinned, device-MAPPED memory: the frame's last small kernel writes {export max, list max, frame + 1} into slot (frame % 4); frame k
    // uses frame k - 2's values (every rank the same frame's: the decisions derived from them must agree) and normally finds them
    // there without waiting -- no D2H copy node, no event, no host synchronisation per frame (round 5)
    volatile int* pub = nullptr;        // host view [4][4]
    int* pub_dev = nullptr;             // device view
    unsigned frame_no = 0;              // frames begun
    unsigned pub_seen = 0;         // frame index + 1 of the published slot phase_begin consumed last (each slot counts once)
    int xsend_hist[4] = {0, 0, 0, 0};   // message size of the last frames
    float dz_hist[4] = {0.f, 0.f, 0.f, 0.f};   // |vertical step| of the last frames
    int nb_hi = 0;                      // longest birth cloud so far (span of the n_static all-reduce)
    int min_slab = 1;                   // thinnest slab of the partition, in layers (number of forwarding rounds)
    long long overflow_frames = 0;
    // the pyramid lists' GLOBAL capacity (:64-66,1256-1259): distributed radix select of every pyramid's CAPP-th smallest sweep key
    int* hist = nullptr;                // [np][256] digit counts of one pass (summed over the ranks)
    int2* sel = nullptr;                // [np] {digits chosen so far, entries still wanted}
    int* kstar = nullptr;               // [np] the selected key (0x7fffffff: the list is not overfull)
    int* kept = nullptr;                // [np] entries of this rank's list under the selected key
    int exact_mode = 1;                 // DSPMAP_SHARDED_EXACT_LISTS: 0 never (a full list is cut per rank), 1 when a list may overflow (default), 2 every frame
    int gcnt_max = 1 << 30;             // the longest list of an earlier frame over all ranks (nothing known yet: assume overfull)
    unsigned state_epoch_seen = ~0u;    // dspmap::state_epoch at the last frame: particles written outside a frame -> nothing known
    long long exact_frames = 0;         // frames that ran the selection
    // one-process group: per-slab, per-phase device time (dspmap_mgpu_group_set_profiling; kept on the group's first handle).  Every
    // slab of the group runs alone on the GPU for the length of its phase, which is what its own GPU would spend on it in a
    // one-process-per-GPU run: sum over the phases of the slowest slab = the frame's critical path without the transport.
    bool gprof = false;
    std::vector<hipEvent_t> gev;        // [slab][phase][start, end]
    std::vector<char> gev_set;          // recorded in the pending frame
    int gprof_spin_us = 400;            // per slab: see dspmap_mgpu_group_update
    std::vector<double> gms;            // [slab][phase] summed ms
    int gprof_n = 0, gprof_frames = 0, gprof_pending = 0;
};

// one-process group: element-wise SUM (int64 / int32) / MAX (int32) over the members' buffers, written back to all of them
struct PtrList { void* p[16]; int n; };

#define NCCLCHK(m, call)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return dspmap_fail((m), DSPMAP_E_DEVICE, "%s failed: %s", #call, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?"); \
    } while (0)

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void k_dist_headers(float* up, float* down, int* cnt2, int* nstatic, int slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        reinterpret_cast<int*>(up)[0] = cnt2[0];
        reinterpret_cast<int*>(down)[0] = cnt2[1];
        nstatic[slot] = max(cnt2[0], cnt2[1]);   // rides on the MAX all-reduce: next frames' message size
        cnt2[0] = 0; cnt2[1] = 0;                // ready for the next frame's export (no memset node per frame)
    }
}
// after the Ck all-reduce: the ranks' list lengths have been summed with it; their maximum goes to the slot behind the export
// maximum (identical on every rank, read by the host a frame later: does the NEXT frame have to run the selection?); a list that
// overflows globally in a frame that did not run it was cut per rank: counted
__global__ void k_dist_gcnt(MapDims d, const long long* __restrict__ gcnt, int* __restrict__ out, int exact, FrameScalars* fs) {
    int mx = 0;
    for (int b = threadIdx.x; b < d.np; b += 64) mx = max(mx, (int)min(gcnt[b], (long long)(1 << 30)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, WAVE));
    if (threadIdx.x == 0) {
        *out = mx;
        if (!exact && mx > d.capp) atomicAdd(&fs->n_overflow_inexact, 1);
    }
}
__global__ void k_dist_publish(const int* __restrict__ two, int* __restrict__ slot, int seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        slot[0] = two[0]; slot[1] = two[1];
        __threadfence_system();
        slot[2] = seq;
    }
}
__global__ void k_group_sum_i32(PtrList l, int count);
__global__ void k_dist_fwd_reset(float* fwd_up, float* fwd_down) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { reinterpret_cast<int*>(fwd_up)[0] = 0; reinterpret_cast<int*>(fwd_down)[0] = 0; }
}
// received records: those of this slab join the inbox of their destination tile (k_place then serves them together with
// the slab's own movers in the order of their source keys); those that have to travel on go to the forward buffer of the
// same direction; `xsend` bounds what the message carried.
__global__ void __launch_bounds__(256) k_dist_import(MapDims d, const float* __restrict__ msg, int xsend, float4* __restrict__ in_rec,
                                                     int* __restrict__ in_cnt, float* __restrict__ fwd, int fwd_cap, int* __restrict__ lost) {
    const int n = min(reinterpret_cast<const int*>(msg)[0], xsend);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool gone = false;
    if (i < n) {
        const float* r = msg + 8 * (size_t)(i + 1);
        const int nlv = __float_as_int(r[0]) - d.v_base;   // (slabs keep index-order storage: the local index IS the storage index)
        if (nlv >= 0 && nlv < d.v_loc) {
            const int tile = nlv >> 6, cap = 64 * d.slots;
            const int pos = atomicAdd(&in_cnt[tile], 1);   // beyond cap: k_place counts it as "voxel full"
            if (pos < cap) {
                const size_t o = ((size_t)tile * cap + pos) * 2;
                in_rec[o] = make_float4(__int_as_float(nlv), r[1], r[2], r[3]);   // (.x of an inbox record: the destination's STORAGE index, like k_predict's own)
                in_rec[o + 1] = make_float4(r[4], r[5], r[6], r[7]);
            }
        } else if (fwd) {
            const int pos = atomicAdd(reinterpret_cast<int*>(fwd), 1);
            if (pos < fwd_cap) { float* o = fwd + 8 * (size_t)(pos + 1); for (int k = 0; k < 8; ++k) o[k] = r[k]; }
            else gone = true;   // the forward buffer is full: counted like every other lost import (n_voxel_full)
        } else gone = true;
    }
    wave_count_add(lost, gone);
}
__global__ void k_group_sum_i32(PtrList l, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int acc = 0;
    for (int k = 0; k < l.n; ++k) acc += reinterpret_cast<int*>(l.p[k])[i];
    for (int k = 0; k < l.n; ++k) reinterpret_cast<int*>(l.p[k])[i] = acc;
}
__global__ void k_group_sum_i64(PtrList l, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    long long acc = 0;
    for (int k = 0; k < l.n; ++k) acc += reinterpret_cast<long long*>(l.p[k])[i];
    for (int k = 0; k < l.n; ++k) reinterpret_cast<long long*>(l.p[k])[i] = acc;
}
__global__ void k_group_max_i32(PtrList l, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int acc = reinterpret_cast<int*>(l.p[0])[i];
    for (int k = 1; k < l.n; ++k) acc = max(acc, reinterpret_cast<int*>(l.p[k])[i]);
    for (int k = 0; k < l.n; ++k) reinterpret_cast<int*>(l.p[k])[i] = acc;
}

// ------------------------------------------------------------------------------------------------ set-up
static int dist_alloc(dspmap* m, int world, int rank) {
    if (m->dist) return dspmap_fail(m, DSPMAP_E_STATE, "the handle already belongs to a communicator / group");
    // dspmap_mgpu_bind hands the handle CALLER-owned Ck / n_static buffers; this driver replaces the Ck buffer with one of its own
    // (np more slots): the two do not mix -- freeing the caller's buffer here would be a double free on the caller's side
    if (m->mgpu_bound && !m->mgpu_self_bound)
        return dspmap_fail(m, DSPMAP_E_STATE, "the handle is bound to caller-owned buffers (dspmap_mgpu_bind) and cannot join a communicator / group: use a handle of its own");
    const MapDims& d = m->d;
    if (world > 1 && d.z_lo == 0 && d.z_hi == d.nz) return dspmap_fail(m, DSPMAP_E_ARG, "a sharded map needs z_lo / z_hi in its configuration");
    dspmap_dist* x = new dspmap_dist();
    x->world = world; x->rank = rank;
    // capacity: two layers of slots (a frame whose vertical step exceeds two voxels overflows and is reported);
    // the MESSAGES are sized from the previous frame's exports, not from this capacity
    const long long layer = (long long)d.nx * d.ny * d.slots;
    x->xcap = (int)std::min<long long>(std::max<long long>(4096, 2 * layer), 64ll << 20);
    x->xsend = (int)std::min<long long>(x->xcap, std::max<long long>(4096, layer / 8));
    for (int dir = 0; dir < 2; ++dir)
        for (int k = 0; k < 3; ++k) {
            HIPCHK(m, hipMalloc((void**)&x->buf[dir][k], sizeof(float) * 8 * ((size_t)x->xcap + 1)));
            HIPCHK(m, hipMemset(x->buf[dir][k], 0, sizeof(float) * 8));
        }
    HIPCHK(m, hipMalloc((void**)&x->cnt2, sizeof(int) * 2));
    HIPCHK(m, hipMemset(x->cnt2, 0, sizeof(int) * 2));   // (k_dist_headers leaves it zeroed for the next frame)
    HIPCHK(m, hipMalloc((void**)&x->hist, sizeof(int) * 256 * (size_t)d.np));
    HIPCHK(m, hipMalloc((void**)&x->sel, sizeof(int2) * (size_t)d.np));
    HIPCHK(m, hipMalloc((void**)&x->kstar, sizeof(int) * (size_t)d.np));
    HIPCHK(m, hipMalloc((void**)&x->kept, sizeof(int) * (size_t)d.np));
    if (const char* e = getenv("DSPMAP_SHARDED_EXACT_LISTS")) x->exact_mode = std::max(0, std::min(2, atoi(e)));
    {   // the ranks' list lengths ride on the Ck all-reduce: np more 64-bit slots behind the np * 100 sums
        long long* ck = nullptr;
        HIPCHK(m, hipMalloc((void**)&ck, sizeof(long long) * ((size_t)d.np * DSP_OBS_CAP + d.np)));
        HIPCHK(m, hipMemset(ck, 0, sizeof(long long) * ((size_t)d.np * DSP_OBS_CAP + d.np)));
        HIPCHK(m, hipStreamSynchronize(m->stream));   // nothing queued may still write the buffer that goes
        if (m->stream2) HIPCHK(m, hipStreamSynchronize(m->stream2));
        (void)hipFree(m->s.obs_ck);
        m->s.obs_ck = ck;
        m->s.pyr_gcnt = ck + (size_t)d.np * DSP_OBS_CAP;
        m->graph_epoch++;
    }
    {
        int* pub_h = nullptr;
        HIPCHK(m, hipHostMalloc((void**)&pub_h, sizeof(int) * 16, hipHostMallocMapped));
        memset(pub_h, 0, sizeof(int) * 16);
        void* dp = nullptr;
        HIPCHK(m, hipHostGetDevicePointer(&dp, pub_h, 0));
        x->pub = pub_h; x->pub_dev = (int*)dp;
    }
    // thinnest slab of an even partition of nz over `world` ranks (dsp-map_amd/sharded.py: slab_ranges)
    x->min_slab = std::max(1, d.nz / std::max(1, world));
    if (!m->k.expmask) {
        const size_t W = (size_t)d.v_loc * d.mw;
        HIPCHK(m, hipMalloc((void**)&m->k.expmask, sizeof(u64) * W));
        HIPCHK(m, hipMemsetAsync(m->k.expmask, 0, sizeof(u64) * W, m->stream));
    }
    // the split-phase entry points of dspmap_mgpu.hip work on the library's own Ck / n_static buffers here
    m->mgpu_bound = true; m->mgpu_self_bound = true;
    m->mgpu_nstatic_cap = m->pt_cap - 1;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    m->dist = x;
    return DSPMAP_OK;
}
void dspmap_dist_free(dspmap* m) {
    dspmap_dist* x = m->dist;
    if (!x) return;
    if (x->comm && rccl()->ok) (void)rccl()->CommDestroy(x->comm);
    for (int dir = 0; dir < 2; ++dir) for (int k = 0; k < 3; ++k) if (x->buf[dir][k]) (void)hipFree(x->buf[dir][k]);
    if (x->cnt2) (void)hipFree(x->cnt2);
    if (x->hist) (void)hipFree(x->hist);
    if (x->sel) (void)hipFree(x->sel);
    if (x->kstar) (void)hipFree(x->kstar);
    if (x->kept) (void)hipFree(x->kept);
    m->s.pyr_kept = nullptr; m->s.pyr_kstar = nullptr; m->s.pyr_gcnt = nullptr;
    if (x->pub) (void)hipHostFree((void*)x->pub);
    for (hipEvent_t e : x->gev) if (e) (void)hipEventDestroy(e);
    delete x;
    m->dist = nullptr;
}

extern "C" int dspmap_mgpu_get_unique_id(char out[DSPMAP_UNIQUE_ID_BYTES]) {
    if (!out) return DSPMAP_E_ARG;
    RcclApi* r = rccl();
    if (!r->ok) return DSPMAP_E_DEVICE;
    static_assert(sizeof(ncclUniqueId) <= DSPMAP_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return DSPMAP_E_DEVICE;
    memset(out, 0, DSPMAP_UNIQUE_ID_BYTES);
    memcpy(out, &id, sizeof(id));
    return DSPMAP_OK;
}

extern "C" int dspmap_mgpu_comm_init(dspmap_t* m, int world, int rank, const char id_bytes[DSPMAP_UNIQUE_ID_BYTES]) {
    INDEX_ORDER(m);
    READY(m);
    if (world < 1 || rank < 0 || rank >= world || !id_bytes) return dspmap_fail(m, DSPMAP_E_ARG, "bad communicator arguments");
    RcclApi* r = rccl();
    if (!r->ok) return dspmap_fail(m, DSPMAP_E_DEVICE, "librccl.so could not be loaded (dlopen): %s", dlerror() ? dlerror() : "symbols missing");
    int rc = dist_alloc(m, world, rank);
    if (rc != DSPMAP_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    NCCLCHK(m, r->CommInitRank(&m->dist->comm, world, id, rank));
    {   // the thinnest slab over all ranks (slabs need not be equally high: a partition balanced by predicted work): the number of
        // forwarding rounds of a frame follows from it and must be the same everywhere -- one 4-byte all-reduce(MIN), once
        int* dmin = nullptr;
        const int mine = m->d.z_hi - m->d.z_lo;
        HIPCHK(m, hipMalloc((void**)&dmin, sizeof(int)));
        HIPCHK(m, hipMemcpyAsync(dmin, &mine, sizeof(int), hipMemcpyHostToDevice, m->stream));
        NCCLCHK(m, r->AllReduce(dmin, dmin, 1, ncclInt32, ncclMin, m->dist->comm, m->stream));
        int got = mine;
        HIPCHK(m, hipMemcpyAsync(&got, dmin, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        (void)hipFree(dmin);
        m->dist->min_slab = std::max(1, got);
    }
    return DSPMAP_OK;
}

// rendezvous through a file for launchers that only export RANK / WORLD_SIZE (torchrun, mpirun wrappers): rank 0 writes
// the unique id, the others wait for it.  DSPMAP_RDZV_FILE names the file (default /tmp/dspmap_rdzv_<uid>_<MASTER_PORT>).
// The file carries a NONCE that only the ranks of this launch share (the launcher's run id and its process id, i.e. the
// ranks' common parent), so that a file left behind by an earlier run on the same port is never taken for this run's:
// rank 0 unlinks whatever is there, creates its file exclusively (O_EXCL | O_NOFOLLOW, mode 0600: nobody else can have
// pre-created it or pointed a link at it) and removes it again once the communicator exists (every rank has read it by then).
namespace {
struct RdzvRecord {
    char magic[8];
    char nonce[120];
    char id[DSPMAP_UNIQUE_ID_BYTES];
};
void rdzv_nonce(char out[120]) {
    const char* run = getenv("DSPMAP_RDZV_NONCE");
    if (!run) run = getenv("TORCHELASTIC_RUN_ID");
    memset(out, 0, 120);   // the whole field is written to the file and compared: no stack residue behind the string
    snprintf(out, 120, "%s:%ld", run ? run : "-", (long)getppid());
}
// does a record read from the rendezvous file belong to this launch?  (magic + the nonce STRING: bytes behind its NUL do not count)
bool rdzv_matches(const RdzvRecord& r, const char nonce[120]) {
    return memcmp(r.magic, "DSPRDZV1", 8) == 0 && strnlen(r.nonce, sizeof(r.nonce)) < sizeof(r.nonce) && strncmp(r.nonce, nonce, sizeof(r.nonce)) == 0;
}
}  // namespace
// test hook (no device, no RCCL): waits up to `timeout_ms` for this launch's record in `path` like a rank != 0 does and copies
// the unique id out.  1 = found, 0 = not found.
extern "C" int dspmap_debug_rdzv_wait(const char* path, int timeout_ms, char id_out[DSPMAP_UNIQUE_ID_BYTES]) {
    char nonce[120];
    rdzv_nonce(nonce);
    for (int t = 0; t <= timeout_ms; t += 10) {
        const int fd = open(path, O_RDONLY | O_NOFOLLOW);
        if (fd >= 0) {
            struct stat st;
            RdzvRecord r2;
            const bool got = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid() &&
                             read(fd, &r2, sizeof(r2)) == (ssize_t)sizeof(r2) && rdzv_matches(r2, nonce);
            close(fd);
            if (got) { if (id_out) memcpy(id_out, r2.id, DSPMAP_UNIQUE_ID_BYTES); return 1; }
        }
        usleep(10000);
    }
    return 0;
}
// ... and what rank 0 publishes (same record layout, same nonce rule); 1 = written
extern "C" int dspmap_debug_rdzv_publish(const char* path, const char id[DSPMAP_UNIQUE_ID_BYTES]) {
    RdzvRecord rec;
    memset(&rec, 0, sizeof(rec));
    memcpy(rec.magic, "DSPRDZV1", 8);
    rdzv_nonce(rec.nonce);
    memcpy(rec.id, id, DSPMAP_UNIQUE_ID_BYTES);
    (void)unlink(path);
    const int fd = open(path, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return 0;
    const bool ok = write(fd, &rec, sizeof(rec)) == (ssize_t)sizeof(rec);
    close(fd);
    return ok ? 1 : 0;
}
extern "C" int dspmap_mgpu_comm_init_from_env(dspmap_t* m) {
    if (!m) return DSPMAP_E_ARG;
    const char* wr = getenv("WORLD_SIZE"); const char* rk = getenv("RANK");
    const int world = wr ? atoi(wr) : 1, rank = rk ? atoi(rk) : 0;
    char path[512];
    const char* f = getenv("DSPMAP_RDZV_FILE");
    if (f) snprintf(path, sizeof(path), "%s", f);
    else snprintf(path, sizeof(path), "/tmp/dspmap_rdzv_%ld_%s", (long)getuid(), getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
    RdzvRecord rec;
    memset(&rec, 0, sizeof(rec));
    char nonce[120];
    rdzv_nonce(nonce);
    if (rank == 0) {
        int rc = dspmap_mgpu_get_unique_id(rec.id);
        if (rc != DSPMAP_OK) return dspmap_fail(m, rc, "ncclGetUniqueId failed");
        memcpy(rec.magic, "DSPRDZV1", 8);
        memcpy(rec.nonce, nonce, sizeof(rec.nonce));
        char tmp[600];
        snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", path, (long)getpid());
        (void)unlink(path);   // a previous run's file (its nonce would not match anyway)
        (void)unlink(tmp);
        const int fd = open(tmp, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0) return dspmap_fail(m, DSPMAP_E_ARG, "cannot create %s", tmp);
        const bool ok = write(fd, &rec, sizeof(rec)) == (ssize_t)sizeof(rec);
        close(fd);
        if (!ok) { (void)unlink(tmp); return dspmap_fail(m, DSPMAP_E_ARG, "cannot write %s", tmp); }
        if (rename(tmp, path) != 0) { (void)unlink(tmp); return dspmap_fail(m, DSPMAP_E_ARG, "cannot publish %s", path); }
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to 60 s
            const int fd = open(path, O_RDONLY | O_NOFOLLOW);
            if (fd >= 0) {
                struct stat st;
                RdzvRecord r2;
                got = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid() &&
                      read(fd, &r2, sizeof(r2)) == (ssize_t)sizeof(r2) && rdzv_matches(r2, nonce);   // this launch's file, not a leftover
                if (got) rec = r2;
                close(fd);
            }
            if (!got) usleep(10000);
        }
        if (!got) return dspmap_fail(m, DSPMAP_E_STATE, "no unique id of this launch (nonce %s) appeared in %s", nonce, path);
    }
    const int rc = dspmap_mgpu_comm_init(m, world, rank, rec.id);
    if (rank == 0) (void)unlink(path);   // ncclCommInitRank returned: every rank has joined, i.e. has read the file
    return rc;
}

extern "C" int dspmap_mgpu_comm_destroy(dspmap_t* m) {
    if (!m) return DSPMAP_E_ARG;
    if (m->device_ready) (void)hipStreamSynchronize(m->stream);
    dspmap_dist_free(m);
    return DSPMAP_OK;
}

// ------------------------------------------------------------------------------------------------ the frame, phase by phase
// (every phase only ENQUEUES on the handle's stream)
static int phase_begin(dspmap* m, int n_points, const float* points_dev, int n_birth, const dspmap_vpoint* birth_dev,
                       const float pos[3], double stamp, const float q[4]) {
    dspmap_dist* x = m->dist;
    // the size of this frame's messages: from the largest export two frames ago at the latest (its copy has landed)
    if (x->frame_no >= 2) {
        const unsigned f = x->frame_no - 2;
        volatile int* sl = x->pub + 4 * (f & 3u);
        for (long spin = 0; (unsigned)sl[2] != f + 1u; ++spin) {   // (frame k - 2 has normally ended long ago)
            if (spin > 4000) { HIPCHK(m, hipStreamSynchronize(m->stream)); if ((unsigned)sl[2] != f + 1u) return dspmap_fail(m, DSPMAP_E_STATE, "frame %u never published its export count", f); break; }
            sched_yield();
        }
        const int g = sl[0];
        x->gcnt_max = sl[1];
        // (a frame that the gate rejects returns below without advancing frame_no: the same slot is looked at again by the next call --
        // its overflow and its crossing ratio are taken into account ONCE)
        const bool fresh = x->pub_seen != f + 1u;
        x->pub_seen = f + 1u;
        if (fresh && g > x->xsend_hist[f & 3u]) ++x->overflow_frames;   // that frame's messages were too small: particles were lost
        // what crossed a face per unit of vertical step in THAT frame: vz == 0, so a frame's exports follow its own |dz| -- the size of
        // this frame's message is derived from ITS step below, not from that count (round 5: a message sized from the last frame's exports
        // alone shrank to its floor after a frame without vertical motion and lost the particles of the next frame that had one)
        const double steps = (double)x->dz_hist[f & 3u] / m->d.res;
        const double layer_cells = (double)m->d.nx * m->d.ny * m->d.slots;
        if (fresh && steps * layer_cells >= 1.0) x->xratio = std::max((double)g / (steps * layer_cells), 0.9 * x->xratio);
    }
    {   // room for the cloud and for the two extra slots of the n_static all-reduce
        const int rcap = dspmap_ensure_point_cap(m, std::max(n_points, n_birth) + 3);
        if (rcap != DSPMAP_OK) return rcap;
        m->mgpu_nstatic_cap = m->pt_cap - 2;
    }
    // Does this frame select the pyramid lists' cut over ALL ranks?  Yes whenever a list may be overfull: nothing is known
    // yet (first frames, particles written outside a frame), or the longest list of the last frame whose count has arrived
    // was at least half the capacity -- list lengths change by the few per cent of the particles that cross a pyramid
    // boundary per frame, so a frame that skips the selection cannot overflow (if it does, n_overflow_inexact counts it).
    // Every input of the decision is identical on every rank (the count is the all-reduced one): the ranks agree.
    if (x->state_epoch_seen != m->state_epoch) { x->state_epoch_seen = m->state_epoch; x->gcnt_max = 1 << 30; }
    m->mgpu_exact_lists = x->world > 1 && (x->exact_mode == 2 || (x->exact_mode == 1 && 2ll * x->gcnt_max >= m->d.capp));
    m->s.pyr_kstar = m->mgpu_exact_lists ? x->kstar : nullptr;
    m->s.pyr_kept = m->mgpu_exact_lists ? x->kept : nullptr;
    const int rc = dspmap_mgpu_begin(m, n_points, points_dev, n_birth, birth_dev, pos, stamp, q);
    if (rc != DSPMAP_OK) return rc;
    {   // this frame's message size, from ITS vertical step (the same pose, ratio and arithmetic on every rank): 1.5 x the expected
        // exports + 2048 records; particles that still carry a vz (constructor pre-fill, first prediction) are not bounded by the step
        const double steps = std::fabs((double)m->hp.od[2]) / m->d.res;
        const double layer_cells = (double)m->d.nx * m->d.ny * m->d.slots;
        const double want = m->vz_frames_at_begin > 0 ? (double)x->xcap : 1.5 * x->xratio * steps * layer_cells + 2048.0;
        x->xsend = (int)std::min<double>((double)x->xcap, std::max<double>(4096.0, std::ceil(want)));
        x->xsend_hist[x->frame_no & 3u] = x->xsend;
        x->dz_hist[x->frame_no & 3u] = std::fabs(m->hp.od[2]);
    }
    x->nb_hi = std::max(x->nb_hi, m->last_n_birth);
    LaunchCtx c = dspmap_ctx_of(m);
    launch_export_slab(c, 0, x->buf[0][0] + 8, x->xsend, x->cnt2, x->buf[1][0] + 8);   // both faces in one pass
    hipLaunchKernelGGL(k_dist_headers, dim3(1), dim3(64), 0, m->stream, x->buf[0][0], x->buf[1][0], x->cnt2, m->s.nstatic, x->nb_hi);
    dspmap_mgpu_birth_early(m, c);   // newborn children: they only need the birth cloud (the rank rode on k_predict / on the estimator, beside it)
    m->mgpu_birth_early = true;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
static int rounds_of(const dspmap* m) {
    // vz == 0: a particle changes layer through the sensor's vertical step only; it crosses at most
    // ceil(|dz| / res) + 1 layers, i.e. that many / (thinnest slab) slab faces -- the same number on every rank
    const dspmap_dist* x = m->dist;
    if (m->vz_frames_at_begin > 0) return std::max(1, x->world - 1);   // constructor-seeded particles still carry vz
    const int layers = (int)std::ceil(std::fabs(m->hp.od[2]) / m->d.res) + 1;
    return std::max(1, std::min(x->world - 1, (layers + x->min_slab - 1) / x->min_slab));
}
static void phase_import(dspmap* m, int dir_from /* 0: message came from below (travels up), 1: from above */, bool last_round) {
    dspmap_dist* x = m->dist;
    LaunchCtx c = dspmap_ctx_of(m);
    float* fwd = last_round ? nullptr : x->buf[dir_from][2];
    hipLaunchKernelGGL(k_dist_import, dim3((x->xsend + 255) / 256), dim3(256), 0, m->stream, m->d, x->buf[dir_from][1], x->xsend,
                       c.k.in_rec, c.k.in_cnt, fwd, x->xsend, &m->s.fs->n_voxel_full_import);
}
static int phase_place(dspmap* m) { return dspmap_mgpu_place_phase(m); }
static int phase_ck(dspmap* m) { return dspmap_mgpu_ck_phase(m); }
// after the Ck all-reduce: the longest list over all ranks -> the slot behind the export maximum
static void phase_gcnt(dspmap* m) {
    dspmap_dist* x = m->dist;
    hipLaunchKernelGGL(k_dist_gcnt, dim3(1), dim3(64), 0, m->stream, m->d, m->s.pyr_gcnt, m->s.nstatic + x->nb_hi + 1, m->mgpu_exact_lists ? 1 : 0, m->s.fs);
}
static int phase_weights(dspmap* m) { return dspmap_mgpu_weights_and_split(m); }
static int phase_finish(dspmap* m) {
    dspmap_dist* x = m->dist;
    // the frame's largest export (all ranks) travels to the host behind the all-reduce; read at the start of a later frame
    hipLaunchKernelGGL(k_dist_publish, dim3(1), dim3(64), 0, m->stream, m->s.nstatic + x->nb_hi, x->pub_dev + 4 * (x->frame_no & 3u), (int)(x->frame_no + 1u));
    ++x->frame_no;
    return dspmap_mgpu_finish(m);
}

extern "C" int dspmap_mgpu_update(dspmap_t* m, int n_points, const float* points_dev, int n_birth,
                                  const dspmap_vpoint* birth_dev, const float pos[3], double stamp, const float q[4]) {
    INDEX_ORDER(m);
    READY(m);
    dspmap_dist* x = m->dist;
    if (!x || !x->comm) return dspmap_fail(m, DSPMAP_E_STATE, "call dspmap_mgpu_comm_init first");
    RcclApi* r = rccl();
    int rc = phase_begin(m, n_points, points_dev, n_birth, birth_dev, pos, stamp, q);
    if (rc != DSPMAP_OK) return rc;   // rejected frames are rejected on every rank (same pose, same stamps)
    const size_t n_msg = 8 * ((size_t)x->xsend + 1);
    const int rounds = rounds_of(m);
    for (int rd = 0; rd < rounds; ++rd) {
        if (rd + 1 < rounds) hipLaunchKernelGGL(k_dist_fwd_reset, dim3(1), dim3(64), 0, m->stream, x->buf[0][2], x->buf[1][2]);
        NCCLCHK(m, r->GroupStart());
        if (x->rank + 1 < x->world) {
            NCCLCHK(m, r->Send(x->buf[0][0], n_msg, ncclFloat, x->rank + 1, x->comm, m->stream));   // up
            NCCLCHK(m, r->Recv(x->buf[1][1], n_msg, ncclFloat, x->rank + 1, x->comm, m->stream));   // what rank + 1 sends down
        }
        if (x->rank > 0) {
            NCCLCHK(m, r->Send(x->buf[1][0], n_msg, ncclFloat, x->rank - 1, x->comm, m->stream));   // down
            NCCLCHK(m, r->Recv(x->buf[0][1], n_msg, ncclFloat, x->rank - 1, x->comm, m->stream));   // what rank - 1 sends up
        }
        NCCLCHK(m, r->GroupEnd());
        if (x->rank > 0) phase_import(m, 0, rd + 1 == rounds);
        if (x->rank + 1 < x->world) phase_import(m, 1, rd + 1 == rounds);
        if (rd + 1 < rounds) { std::swap(x->buf[0][0], x->buf[0][2]); std::swap(x->buf[1][0], x->buf[1][2]); }   // forward what has to travel on
    }
    rc = phase_place(m);
    if (rc != DSPMAP_OK) return rc;
    if (m->mgpu_exact_lists) {
        // SAFE_PARTICLE_NUM_PYRAMID over all ranks: one small all-reduce per 8-bit digit of the selection
        LaunchCtx c = dspmap_ctx_of(m);
        for (int ps = 0; ps < pyr_select_passes(); ++ps) {
            launch_pyr_hist(c, ps, x->sel, x->hist);
            NCCLCHK(m, r->AllReduce(x->hist, x->hist, (size_t)m->d.np * 256, ncclInt32, ncclSum, x->comm, m->stream));
            launch_pyr_pick(c, ps, x->hist, x->sel, x->kstar);
        }
        launch_pyr_kept(c, x->kstar, x->kept);
        ++x->exact_frames;
    }
    rc = phase_ck(m);
    if (rc != DSPMAP_OK) return rc;
    NCCLCHK(m, r->AllReduce(m->s.obs_ck, m->s.obs_ck, (size_t)m->d.np * DSP_OBS_CAP + m->d.np, ncclInt64, ncclSum, x->comm, m->stream));
    phase_gcnt(m);
    rc = phase_weights(m);
    if (rc != DSPMAP_OK) return rc;
    NCCLCHK(m, r->AllReduce(m->s.nstatic, m->s.nstatic, (size_t)x->nb_hi + 2, ncclInt32, ncclMax, x->comm, m->stream));
    rc = phase_finish(m);
    if (rc != DSPMAP_OK) return rc;
    if (x->overflow_frames) {
        const long long n = x->overflow_frames;
        x->overflow_frames = 0;
        return dspmap_fail(m, DSPMAP_E_STATE, "%lld earlier frame(s) exported more particles across a slab face than the exchange message held "
                           "(vertical step larger than the previous frames'): those particles were lost; the message size has been raised", n);
    }
    return DSPMAP_OK;
}

// ------------------------------------------------------------------------------------------------ one-process group
static void group_prof_collect(dspmap_dist* g) {
    if (!g->gprof_pending || g->gev.empty()) return;
    const int n = g->gprof_n;
    for (int i = 0; i <= n; ++i)
        for (int ph = 0; ph < DSPMAP_GROUP_PHASES; ++ph) {
            float ms = 0.f;
            const size_t e = ((size_t)i * DSPMAP_GROUP_PHASES + ph) * 2;
            if (!g->gev_set[e] || !g->gev_set[e + 1]) continue;
            (void)hipEventSynchronize(g->gev[e + 1]);
            if (hipEventElapsedTime(&ms, g->gev[e], g->gev[e + 1]) == hipSuccess && ms > 0.f) g->gms[(size_t)i * DSPMAP_GROUP_PHASES + ph] += ms;
        }
    ++g->gprof_frames;
    g->gprof_pending = 0;
}
extern "C" int dspmap_mgpu_group_set_profiling(dspmap_t** hs, int n, int on) {
    if (!hs || n < 1 || n > 16 || !hs[0] || !hs[0]->dist) return DSPMAP_E_ARG;
    dspmap* m = hs[0];
    dspmap_dist* g = m->dist;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (on && (g->gev.empty() || g->gprof_n != n)) {
        for (hipEvent_t e : g->gev) (void)hipEventDestroy(e);
        g->gev.assign((size_t)(n + 1) * DSPMAP_GROUP_PHASES * 2, nullptr);
        for (hipEvent_t& e : g->gev) HIPCHK(m, hipEventCreate(&e));
        g->gprof_n = n;
    }
    g->gev_set.assign(g->gev.size(), 0);
    g->gprof_spin_us = 2000 * n;
    g->gms.assign((size_t)(n + 1) * DSPMAP_GROUP_PHASES, 0.0);
    g->gprof_frames = 0; g->gprof_pending = 0;
    g->gprof = on != 0;
    return DSPMAP_OK;
}
extern "C" int dspmap_mgpu_group_get_phase_ms(dspmap_t** hs, int n, float* out, int* n_frames) {
    if (!hs || n < 1 || n > 16 || !hs[0] || !hs[0]->dist || !out) return DSPMAP_E_ARG;
    dspmap_dist* g = hs[0]->dist;
    if (g->gprof_n != n || g->gms.empty()) return dspmap_fail(hs[0], DSPMAP_E_STATE, "group profiling is not enabled for %d slabs", n);
    group_prof_collect(g);
    for (size_t i = 0; i < g->gms.size(); ++i) out[i] = (float)g->gms[i];
    if (n_frames) *n_frames = g->gprof_frames;
    return DSPMAP_OK;
}
extern "C" int dspmap_mgpu_group_create(dspmap_t** hs, int n) {
    if (!hs || n < 1 || n > 16) return DSPMAP_E_ARG;
    for (int i = 0; i < n; ++i) {
        dspmap* m = hs[i];
        INDEX_ORDER(m);
        READY(m);
        if (i > 0) { int rc = dspmap_set_stream(m, (void*)hs[0]->stream); if (rc != DSPMAP_OK) return rc; }   // one stream orders the group
        int rc = dist_alloc(m, n, i);
        if (rc != DSPMAP_OK) return rc;
    }
    // slabs need not be equally high (a partition balanced by predicted work, bench.py: projected_8gpu): the number of forwarding
    // rounds follows from the THINNEST one
    int thin = 1 << 30;
    for (int i = 0; i < n; ++i) thin = std::min(thin, hs[i]->d.z_hi - hs[i]->d.z_lo);
    for (int i = 0; i < n; ++i) hs[i]->dist->min_slab = std::max(1, thin);
    return DSPMAP_OK;
}
extern "C" int dspmap_mgpu_group_update(dspmap_t** hs, int n, int n_points, const float* points_dev, int n_birth,
                                        const dspmap_vpoint* birth_dev, const float pos[3], double stamp, const float q[4]) {
    if (!hs || n < 1 || n > 16) return DSPMAP_E_ARG;
    for (int i = 0; i < n; ++i) if (!hs[i] || !hs[i]->dist || hs[i]->dist->comm) return DSPMAP_E_STATE;
    hipStream_t st = hs[0]->stream;
    dspmap_dist* g0 = hs[0]->dist;
    const bool prof = g0->gprof && g0->gprof_n == n;
    if (prof && g0->gprof_pending) group_prof_collect(g0);
    auto mark = [&](int slab, int phase, int end) {   // (slab n = the group's stand-ins for the collectives)
        if (!prof) return;
        const size_t e = ((size_t)slab * DSPMAP_GROUP_PHASES + phase) * 2 + end;
        (void)hipEventRecord(g0->gev[e], st);
        g0->gev_set[e] = 1;   // (phases a frame skips are not read)
    };
    if (prof) {
        std::fill(g0->gev_set.begin(), g0->gev_set.end(), 0);
        // the intervals between two events must hold device work only: a one-wave kernel keeps the stream busy while the host queues
        // the whole frame (8 slabs x 7 phases from ONE thread take longer to queue than to run; a rank of a real run queues one slab)
        LaunchCtx c0 = dspmap_ctx_of(hs[0]);
        launch_spin(c0, g0->gprof_spin_us);
    }
    int accepted = 0;
    for (int i = 0; i < n; ++i) {
        mark(i, 0, 0);
        const int rc = phase_begin(hs[i], n_points, points_dev, n_birth, birth_dev, pos, stamp, q);
        // (profiled: the estimator's side branch is joined inside the slab's own window, see the Ck phase below)
        if (prof && rc == DSPMAP_OK && hs[i]->mgpu_side_pending) { (void)hipStreamWaitEvent(st, hs[i]->ev_join, 0); hs[i]->mgpu_side_pending = false; }
        mark(i, 0, 1);
        if (rc < 0) return rc;
        accepted += rc == DSPMAP_OK ? 1 : 0;
    }
    if (accepted == 0) return DSPMAP_REJECTED;
    if (accepted != n) return dspmap_fail(hs[0], DSPMAP_E_STATE, "the slabs of a group disagree about a frame");
    const int rounds = rounds_of(hs[0]);
    const size_t bytes = sizeof(float) * 8 * ((size_t)hs[0]->dist->xsend + 1);
    for (int rd = 0; rd < rounds; ++rd) {
        for (int i = 0; i < n; ++i)
            if (rd + 1 < rounds) hipLaunchKernelGGL(k_dist_fwd_reset, dim3(1), dim3(64), 0, st, hs[i]->dist->buf[0][2], hs[i]->dist->buf[1][2]);
        if (rd == 0) mark(n, 1, 0);
        for (int i = 0; i < n; ++i) {   // the "send / recv" pairs
            if (i + 1 < n) (void)hipMemcpyAsync(hs[i + 1]->dist->buf[0][1], hs[i]->dist->buf[0][0], bytes, hipMemcpyDeviceToDevice, st);
            if (i > 0) (void)hipMemcpyAsync(hs[i - 1]->dist->buf[1][1], hs[i]->dist->buf[1][0], bytes, hipMemcpyDeviceToDevice, st);
        }
        if (rd == 0) mark(n, 1, 1);
        for (int i = 0; i < n; ++i) {
            if (rd == 0) mark(i, 1, 0);
            if (i > 0) phase_import(hs[i], 0, rd + 1 == rounds);
            if (i + 1 < n) phase_import(hs[i], 1, rd + 1 == rounds);
            if (rd == 0) mark(i, 1, 1);
            if (rd + 1 < rounds) { std::swap(hs[i]->dist->buf[0][0], hs[i]->dist->buf[0][2]); std::swap(hs[i]->dist->buf[1][0], hs[i]->dist->buf[1][2]); }
        }
    }
    PtrList l;
    l.n = n;
    for (int i = 0; i < n; ++i) { mark(i, 2, 0); const int rc = phase_place(hs[i]); mark(i, 2, 1); if (rc != DSPMAP_OK) return rc; }
    if (hs[0]->mgpu_exact_lists) {
        mark(n, 3, 0);
        for (int i = 1; i < n; ++i) if (!hs[i]->mgpu_exact_lists) return dspmap_fail(hs[0], DSPMAP_E_STATE, "the slabs of a group disagree about the list selection");
        const int n_h = hs[0]->d.np * 256;
        for (int ps = 0; ps < pyr_select_passes(); ++ps) {
            for (int i = 0; i < n; ++i) { LaunchCtx c = dspmap_ctx_of(hs[i]); launch_pyr_hist(c, ps, hs[i]->dist->sel, hs[i]->dist->hist); l.p[i] = hs[i]->dist->hist; }
            hipLaunchKernelGGL(k_group_sum_i32, dim3((n_h + 255) / 256), dim3(256), 0, st, l, n_h);
            for (int i = 0; i < n; ++i) { LaunchCtx c = dspmap_ctx_of(hs[i]); launch_pyr_pick(c, ps, hs[i]->dist->hist, hs[i]->dist->sel, hs[i]->dist->kstar); }
        }
        for (int i = 0; i < n; ++i) { LaunchCtx c = dspmap_ctx_of(hs[i]); launch_pyr_kept(c, hs[i]->dist->kstar, hs[i]->dist->kept); ++hs[i]->dist->exact_frames; }
        mark(n, 3, 1);   // (the whole selection, all slabs: 4 x (histogram, sum, pick) -- per rank a 1 / n share of the kernels + 4 all-reduces)
    }
    for (int i = 0; i < n; ++i) {
        mark(i, 4, 0);
        const int rc = phase_ck(hs[i]);
        // a slab that places the arrivals of its tiles without a view on the side stream (large slabs) joins that launch in its weight phase --
        // which, in a group, comes after EVERY slab's Ck phase: unprofiled that is all the same (one GPU does all the work either way),
        // but a profiled group charges each slab its own time, and the side launches of the slabs before would run inside the windows of
        // the slabs behind (round 6: the Ck phases of slabs without a view measured 0.28 - 0.32 ms beside 0.02 - 0.03 for their
        // neighbours).  Profiled, a slab's side launch is joined inside its own window -- on its own GPU it overlaps its own Ck pass,
        // the all-reduce and the weight update, so this charges it in full rather than not at all
        if (prof && rc == DSPMAP_OK && hs[i]->mgpu_side_pending) { (void)hipStreamWaitEvent(st, hs[i]->ev_join, 0); hs[i]->mgpu_side_pending = false; }
        mark(i, 4, 1);
        if (rc != DSPMAP_OK) return rc;
        l.p[i] = hs[i]->s.obs_ck;
    }
    const int n_ck = hs[0]->d.np * DSP_OBS_CAP + hs[0]->d.np;
    mark(n, 4, 0);
    hipLaunchKernelGGL(k_group_sum_i64, dim3((n_ck + 255) / 256), dim3(256), 0, st, l, n_ck);
    for (int i = 0; i < n; ++i) phase_gcnt(hs[i]);
    mark(n, 4, 1);
    int span = 0;
    for (int i = 0; i < n; ++i) { mark(i, 5, 0); const int rc = phase_weights(hs[i]); mark(i, 5, 1); if (rc != DSPMAP_OK) return rc; l.p[i] = hs[i]->s.nstatic; span = std::max(span, hs[i]->dist->nb_hi + 2); }
    mark(n, 5, 0);
    hipLaunchKernelGGL(k_group_max_i32, dim3((span + 255) / 256), dim3(256), 0, st, l, span);
    mark(n, 5, 1);
    for (int i = 0; i < n; ++i) { mark(i, 6, 0); const int rc = phase_finish(hs[i]); mark(i, 6, 1); if (rc != DSPMAP_OK) return rc; }
    if (prof) g0->gprof_pending = 1;
    long long ov = 0;
    for (int i = 0; i < n; ++i) { ov += hs[i]->dist->overflow_frames; hs[i]->dist->overflow_frames = 0; }
    if (ov) return dspmap_fail(hs[0], DSPMAP_E_STATE, "an earlier frame exported more particles across a slab face than the exchange message held");
    return DSPMAP_OK;
}
extern "C" int dspmap_mgpu_message_records(const dspmap_t* m) { return (m && m->dist) ? m->dist->xsend : 0; }

// the host-buffer twin of dspmap_mgpu_update (what the drop-in class DSPMap calls when it is built with -DDSPMAP_WORLD)
extern "C" int dspmap_mgpu_update_host(dspmap_t* m, int n, int stride, const float* pts, float sx, float sy, float sz, double stamp,
                                       float qw, float qx, float qy, float qz) {
    INDEX_ORDER(m);
    READY(m);
    if (n > 0 && (!pts || stride < 3)) return dspmap_fail(m, DSPMAP_E_ARG, "bad point cloud arguments");
    const int np = n > 0 ? n : 0;
    int rc = dspmap_ensure_point_cap(m, np + 2);
    if (rc != DSPMAP_OK) return rc;
    rc = dspmap_stage_points(m, np, stride, pts);
    if (rc != DSPMAP_OK) return rc;
    const float pos[3] = {sx, sy, sz}, q[4] = {qw, qx, qy, qz};
    return dspmap_mgpu_update(m, np, m->pts_dev, 0, nullptr, pos, stamp, q);
}
