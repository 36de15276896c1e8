// opd_comm.cpp — the path's ONE exchange step inside the C-ABI: an RCCL all-gather of fixed-size detection records over xGMI
// (BASELINE.json north_star: "one frame-batch shard per rank, RCCL all-gather of detections back to the orchestrator").
//
// The reference has no distributed mode (single process, per-frame loop: src/pipeline/phases/detection.py:91-94); the port it reserves for
// batched detectors is DetectorPort.detect(frames: Sequence[FrameDTO]) (src/core/interfaces.py:30-34).  Rounds 1-3 did the exchange in
// Python on torch.distributed: two host synchronisations per step (wait for the post-process kernel, then all_gather_into_tensor on
// torch's stream, then .cpu()), and a reference-side ctypes binding could not run the sharded configurations without importing torch.
// Here the step is enqueued without a host wait in between:
//     handle's stream:        forward -> post-process kernel (writes the records into the lane's send buffer) -> event `ready`
//     communicator's stream:  wait(ready) -> ncclAllGather -> copy to page-locked host memory -> event `done`
// and the host waits once, on `done`.
//
// ONE communicator per rank, any number of LANES (round 5).  A rank that keeps several batches in flight has several detector handles
// (own stream, workspace, graph).  Round 4 gave each handle a communicator of its own and enqueued its all-gather on the handle's stream:
// three communicators per rank whose collectives run on three unordered streams -- RCCL requires every rank to issue the collectives of a
// communicator in the same order, and says nothing good about the device-side order of collectives on DIFFERENT communicators that share
// links (a documented deadlock hazard), and that arrangement had never run on more than one GPU.  Now an `opd_comm` is a lane: a handle's
// send / receive buffers and events on a communicator shared by all lanes of the rank (`opd_comm_attach`).  Every all-gather of the rank is
// enqueued on the communicator's own stream under its mutex, i.e. in the order the host submitted the exchanges, and every rank submits
// them in the same order (step i uses lane i mod N everywhere): one communicator, one in-order stream, one order -- RCCL's contract as
// written.  The handles' streams never wait for a collective: the next forward of a handle starts while its records are still travelling.  RCCL is resolved at run time (dlopen of librccl.so.1: the copy a host process has already
// loaded, e.g. PyTorch's, or ROCm's), so libopd_hip.so itself does not link it and single-GPU callers never touch it.
// Whoever launches the ranks carries the 128-byte unique id from rank 0 to the others (a file, a socket, MPI, torch.distributed: the
// Python layer uses its process-group store); the library does no rendezvous of its own.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "opd_model.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: the copy the process already has (PyTorch ships its own librccl.so); then the usual names
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) { r.error = std::string("librccl.so.1 not found (") + dlerror() + ")"; return; }
        auto sym = [&](const char* s) -> void* {
            void* p = dlsym(r.lib, s);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

#define NCCLCHK(expr)                                                                                                      \
    do {                                                                                                                   \
        ncclResult_t _r = (expr);                                                                                          \
        if (_r != ncclSuccess)                                                                                             \
            return opd::fail(OPD_EHIP, std::string(#expr) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(_r) : "?")); \
    } while (0)

}  // namespace

// what the lanes of one rank share: the RCCL communicator and the one stream its collectives are enqueued on
struct CommShared {
    ncclComm_t comm = nullptr;
    hipStream_t xstream = nullptr;
    int device = 0, rank = 0, world = 1;
    std::mutex mu;   // serialises [wait(ready), all-gather, copy, record(done)] of the lanes: submission order = enqueue order
    ~CommShared() {
        (void)hipSetDevice(device);
        if (xstream) (void)hipStreamSynchronize(xstream);
        if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
        if (xstream) (void)hipStreamDestroy(xstream);
    }
};

// One exchange's buffers and events.  A lane owns TWO of them, used in turn: while exchange k travels (its all-gather and copy on the
// communicator's stream), the handle's next forward fills the send buffer of exchange k + 1 -- the same depth the plain asynchronous path has
// with its rotating output buffers, so the bench's pipelined loop (submit step i, then collect step i - NS) is the same with and without the
// exchange.  (With one set the loop had to collect before it submitted: -3 % on one GPU, profiles/r05_ab_runs.txt.)
struct LaneSet {
    int32_t* d_send = nullptr;  // [slots * Q * 8 record words][slots counts]
    int32_t* d_recv = nullptr;  // [world] x the same
    int32_t* h_recv = nullptr;  // page-locked copy of d_recv
    size_t cap_words = 0;       // words per rank the buffers hold
    int slots = 0;              // frame slots per rank of this exchange
    hipEvent_t ready = nullptr; // recorded on the handle's stream behind the post-process kernel(s) of the exchange
    hipEvent_t done = nullptr;  // recorded on the communicator's stream behind the copy to h_recv
    int state = 0;              // 0 idle, 1 begun (being filled), 2 exchanged (travelling / arrived, not yet waited for)
};

struct opd_comm {
    std::shared_ptr<CommShared> sh;
    opd_detr* m = nullptr;      // the handle this lane is bound to (its stream produces the records); null once that handle has been destroyed
    LaneSet set[2];
    unsigned n_begun = 0, n_waited = 0;   // exchanges begun / waited for: exchange k lives in set[k & 1]
};

namespace {
std::mutex g_lanes_mu;   // guards opd_detr::comms (the lanes bound to a handle)

void free_set(LaneSet& s) {
    if (s.d_send) (void)hipFree(s.d_send);
    if (s.d_recv) (void)hipFree(s.d_recv);
    if (s.h_recv) (void)hipHostFree(s.h_recv);
    s.d_send = s.d_recv = s.h_recv = nullptr;
    s.cap_words = 0;
}

int make_lane(std::shared_ptr<CommShared> sh, opd_detr* m, opd_comm** out) {
    std::unique_ptr<opd_comm> c(new (std::nothrow) opd_comm());
    if (!c) return fail(OPD_ENOMEM, "opd_comm: out of host memory");
    c->sh = std::move(sh); c->m = m;
    for (LaneSet& s : c->set)
        if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            for (LaneSet& t : c->set) {
                if (t.ready) (void)hipEventDestroy(t.ready);
                if (t.done) (void)hipEventDestroy(t.done);
            }
            return fail(OPD_EHIP, "opd_comm: hipEventCreate failed");
        }
    {
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        m->comms.push_back(c.get());
    }
    *out = c.release();
    return OPD_OK;
}
#define LANE_ALIVE(c, what)                                                                                                      \
    do {                                                                                                                         \
        if (!(c)->m) return fail(OPD_ESTATE, what ": the detector handle this communicator lane was bound to has been destroyed"); \
    } while (0)
// the exchange being filled (begun, not yet exchanged), or null
LaneSet* open_set(opd_comm* c) { return c->n_begun > 0 && c->set[(c->n_begun - 1) & 1].state == 1 ? &c->set[(c->n_begun - 1) & 1] : nullptr; }
}  // namespace

// opd_detr_destroy: the handle's stream is about to go.  Its lanes stay valid objects (opd_comm_destroy still frees them) but refuse work.
void opd::comm_detach_all(opd_detr* m) {
    std::lock_guard<std::mutex> lk(g_lanes_mu);
    for (opd_comm* c : m->comms) {
        if (c->sh && c->sh->xstream) (void)hipStreamSynchronize(c->sh->xstream);   // (an exchange still in flight reads this handle's records)
        c->m = nullptr;
        for (LaneSet& s : c->set) s.state = 0;
    }
    m->comms.clear();
}

extern "C" {

int opd_comm_unique_id(void* id128) {
    if (!id128) return fail(OPD_EINVAL, "opd_comm_unique_id: null buffer");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(OPD_EHIP, "RCCL unavailable: " + r.error);
    static_assert(sizeof(ncclUniqueId) == OPD_COMM_ID_BYTES, "OPD_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCLCHK(r.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return OPD_OK;
}

int opd_comm_available(void) {
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(OPD_EHIP, "RCCL unavailable: " + r.error);
    return OPD_OK;
}

int opd_comm_create(const void* id128, int rank, int world, opd_detr* m, opd_comm** out) {
    // (the API lock stays held across ncclCommInitRank: it allocates device memory, which must not run beside another thread's stream capture;
    //  set-up only.  A rank whose peers never arrive blocks here: callers bound the set-up with a watchdog, see sharding.py / bench.py)
    ApiScope api_scope;
    if (out) *out = nullptr;
    if (!id128 || !m || !out || world < 1 || rank < 0 || rank >= world) return fail(OPD_EINVAL, "opd_comm_create: bad argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(OPD_EHIP, "RCCL unavailable: " + r.error);
    HIPCHK(hipSetDevice(m->device));
    std::shared_ptr<CommShared> sh(new (std::nothrow) CommShared());
    if (!sh) return fail(OPD_ENOMEM, "opd_comm_create: out of host memory");
    sh->device = m->device; sh->rank = rank; sh->world = world;
    HIPCHK(hipStreamCreateWithFlags(&sh->xstream, hipStreamNonBlocking));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCLCHK(r.CommInitRank(&sh->comm, world, id, rank));
    return make_lane(std::move(sh), m, out);
}

int opd_comm_attach(opd_comm* parent, opd_detr* m, opd_comm** out) {
    ApiScope api_scope;
    if (out) *out = nullptr;
    if (!parent || !m || !out) return fail(OPD_EINVAL, "opd_comm_attach: null argument");
    if (m->device != parent->sh->device) return fail(OPD_EINVAL, "opd_comm_attach: the handle lives on another device than the communicator");
    HIPCHK(hipSetDevice(m->device));
    return make_lane(parent->sh, m, out);
}

void opd_comm_destroy(opd_comm* c) {
    ApiScope api_scope;
    if (!c) return;
    (void)hipSetDevice(c->sh->device);
    (void)hipStreamSynchronize(c->sh->xstream);   // this lane's exchanges, if any, are over
    if (c->m) {
        (void)hipStreamSynchronize(c->m->stream);
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        auto& v = c->m->comms;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == c) { v.erase(v.begin() + i); break; }
    }
    for (LaneSet& s : c->set) {
        free_set(s);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    delete c;   // (the communicator and its stream go with the last lane: ~CommShared)
}

int opd_comm_begin(opd_comm* c, int slots) {
    ApiScope api_scope;
    if (!c || slots < 1) return fail(OPD_EINVAL, "opd_comm_begin: bad argument");
    LANE_ALIVE(c, "opd_comm_begin");
    if (open_set(c)) return fail(OPD_ESTATE, "opd_comm_begin: the exchange begun before has not been issued (opd_comm_exchange)");
    LaneSet& s = c->set[c->n_begun & 1];
    if (s.state != 0) return fail(OPD_ESTATE, "opd_comm_begin: two exchanges of this lane are outstanding (opd_comm_wait the older one first)");
    opd_detr* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    const size_t Q = (size_t)m->arch.queries, words = (size_t)slots * Q * 8 + (size_t)slots;
    if (words > s.cap_words) {   // (grow only; not on the steady-state path.  The set is idle: nothing in flight reads or writes it)
        free_set(s);
        void *a = nullptr, *b = nullptr, *h = nullptr;
        if (hipMalloc(&a, words * 4) != hipSuccess || hipMalloc(&b, words * 4 * c->sh->world) != hipSuccess ||
            hipHostMalloc(&h, words * 4 * c->sh->world, hipHostMallocDefault) != hipSuccess) {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            return fail(OPD_ENOMEM, "opd_comm_begin: exchange buffers");
        }
        s.d_send = static_cast<int32_t*>(a); s.d_recv = static_cast<int32_t*>(b); s.h_recv = static_cast<int32_t*>(h);
        s.cap_words = words;
    }
    s.slots = slots;
    // every slot starts as "no frame" (count -1): a rank with fewer frames than slots leaves the rest that way
    HIPCHK(hipMemsetAsync(s.d_send + (size_t)slots * Q * 8, 0xFF, (size_t)slots * 4, m->stream));
    s.state = 1;
    ++c->n_begun;
    return OPD_OK;
}

int opd_comm_detect(opd_comm* c, int slot0, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                    const int32_t* orig_hw) {
    ApiScope api_scope;
    if (!c) return fail(OPD_EINVAL, "opd_comm_detect: null communicator");
    LANE_ALIVE(c, "opd_comm_detect");
    opd_detr* m = c->m;
    RCCHK(check_shape(m, pixels, pixel_format, mem_kind, B, H, W));
    LaneSet* s = open_set(c);
    if (!s || slot0 < 0 || slot0 + B > s->slots) return fail(OPD_EINVAL, "opd_comm_detect: frames do not fit the slots of opd_comm_begin");
    if (m->profiling) return fail(OPD_ESTATE, "opd_comm_detect is not available in profiling mode");
    HIPCHK(hipSetDevice(m->device));
    const void* d_pixels = nullptr;
    RCCHK(stage_pixels(m, pixels, pixel_format, mem_kind, B, H, W, &d_pixels));
    RCCHK(run_forward(m, d_pixels, pixel_format, B, H, W, nullptr));
    const size_t Q = (size_t)m->arch.queries;
    opd_det* recs = reinterpret_cast<opd_det*>(s->d_send) + (size_t)slot0 * Q;
    int32_t* counts = s->d_send + (size_t)s->slots * Q * 8 + slot0;
    return enqueue_postprocess(m, threshold, orig_hw, recs, counts);
}

int opd_comm_buffers(opd_comm* c, int slot0, void** records, void** counts) {
    if (!c || !records || !counts) return fail(OPD_EINVAL, "opd_comm_buffers: null argument");
    LANE_ALIVE(c, "opd_comm_buffers");
    LaneSet* s = open_set(c);
    if (!s || slot0 < 0 || slot0 >= s->slots) return fail(OPD_EINVAL, "opd_comm_buffers: slot outside the exchange begun");
    const size_t Q = (size_t)c->m->arch.queries;
    *records = reinterpret_cast<opd_det*>(s->d_send) + (size_t)slot0 * Q;
    *counts = s->d_send + (size_t)s->slots * Q * 8 + slot0;
    return OPD_OK;
}

int opd_comm_exchange(opd_comm* c) {
    ApiScope api_scope;
    if (!c) return fail(OPD_EINVAL, "opd_comm_exchange: null communicator");
    LANE_ALIVE(c, "opd_comm_exchange");
    LaneSet* s = open_set(c);
    if (!s) return fail(OPD_EINVAL, "opd_comm_exchange: no exchange begun");
    opd_detr* m = c->m;
    CommShared& sh = *c->sh;
    HIPCHK(hipSetDevice(m->device));
    const size_t words = (size_t)s->slots * m->arch.queries * 8 + (size_t)s->slots;
    HIPCHK(hipEventRecord(s->ready, m->stream));   // behind the post-process kernel(s) that filled the send buffer
    {
        std::lock_guard<std::mutex> lk(sh.mu);     // the rank's collectives in ONE order: the order the exchanges were submitted in
        HIPCHK(hipStreamWaitEvent(sh.xstream, s->ready, 0));
        NCCLCHK(rccl().AllGather(s->d_send, s->d_recv, words, ncclInt32, sh.comm, sh.xstream));
        HIPCHK(hipMemcpyAsync(s->h_recv, s->d_recv, words * 4 * sh.world, hipMemcpyDeviceToHost, sh.xstream));
        HIPCHK(hipEventRecord(s->done, sh.xstream));
    }
    s->state = 2;
    return OPD_OK;
}

int opd_comm_wait(opd_comm* c, opd_det* out_all, int32_t* counts_all) {
    ApiScope api_scope;
    if (!c || !out_all || !counts_all) return fail(OPD_EINVAL, "opd_comm_wait: null argument");
    LANE_ALIVE(c, "opd_comm_wait");
    LaneSet& s = c->set[c->n_waited & 1];   // the OLDEST exchange not yet waited for
    if (c->n_waited == c->n_begun || s.state != 2) return fail(OPD_ESTATE, "opd_comm_wait: no exchange outstanding");
    HIPCHK(hipSetDevice(c->m->device));
    {
        ApiUnlocked unlocked;   // the step's ONE host wait -- for the remote ranks too: another thread's graph capture must not queue behind it
        HIPCHK(hipEventSynchronize(s.done));
    }
    const size_t Q = (size_t)c->m->arch.queries, nrec = (size_t)s.slots * Q * 8, words = nrec + s.slots;
    for (int r = 0; r < c->sh->world; ++r) {
        memcpy(out_all + (size_t)r * s.slots * Q, s.h_recv + (size_t)r * words, nrec * 4);
        memcpy(counts_all + (size_t)r * s.slots, s.h_recv + (size_t)r * words + nrec, (size_t)s.slots * 4);
    }
    s.state = 0;
    ++c->n_waited;
    return OPD_OK;
}

int opd_comm_info(const opd_comm* c, int* rank, int* world) {
    if (!c || !rank || !world) return fail(OPD_EINVAL, "opd_comm_info: null argument");
    *rank = c->sh->rank; *world = c->sh->world;
    return OPD_OK;
}

}  // extern "C"
