// Gradient all-reduce inside the library: dc_grad_allreduce_{enqueue,wait} over a raw RCCL communicator (replaces the
// reducer of apex / torch DistributedDataParallel at train_hdf5_ddp.py:227,363).
//
// A dc communicator owns (a) the collective transport -- an ncclComm_t created from a unique id that the host program has handed to every
// rank (dc_comm_unique_id on rank 0, any out-of-band broadcast, dc_comm_create everywhere), an adopted ncclComm_t (dc_comm_adopt), or a
// host callback (dc_comm_create_callback: how the two-rank gloo tests drive these very entry points without RCCL) -- and (b) a HIP stream of
// its own for the collectives plus the events that order it against the caller's compute stream:
//
//   enqueue(comm, buf, count, dtype, compute)   the communication stream waits for everything enqueued on `compute` so far, then all-reduces
//                                               (SUM, in place) buf[0 .. count) on it: the collective overlaps whatever `compute` runs next
//   wait(comm, compute)                         `compute` waits for everything enqueued on the communication stream so far
//
// Both are plain int-returning entry points with word-sized arguments, so a recorded launch list (dc_program_*) replays them with the kernels
// of the step: the multi-GPU step needs no Python between two launches either.
//
// RCCL is resolved at run time (dlopen / dlsym: the copy already loaded into the process -- PyTorch ships one -- else the system's), so the
// library itself has no link-time dependency on it and loads on a machine without a GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "../../include/deepcam_hip.h"

extern "C" int dc_set_error(int code, const char* file, int line);
extern "C" int dc_fail(const char* msg, const char* file, int line);

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0)
struct UniqueId {
  char internal[128];
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn error_string = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;        // the copy the process already has (PyTorch's)
    if (r.handle == nullptr)
      for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (r.handle == nullptr) return;
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
  });
  return r;
}

int rccl_fail(int code, const char* what) {
  char msg[256];
  const Rccl& r = rccl();
  snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, code, r.error_string ? r.error_string(code) : "?");
  return dc_fail(msg, __FILE__, __LINE__);
}

struct DcComm {
  int rank = 0, world = 1;
  Comm nccl = nullptr;          // null: single rank or callback transport
  bool owns_nccl = false;
  dc_allreduce_callback cb = nullptr;
  void* cb_ctx = nullptr;
  int cb_sync = 1;              // callback transport: synchronise the compute stream before calling back (device buffers)
  hipStream_t stream = nullptr; // the communication stream (RCCL transport)
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  long enqueued = 0;
};

int make_stream(DcComm* c) {
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  return 0;
}

}  // namespace

extern "C" int dc_comm_unique_id(void* id128) {
  if (id128 == nullptr) return dc_fail("dc_comm_unique_id: null argument", __FILE__, __LINE__);
  Rccl& r = rccl();
  if (!r.ok) return dc_fail("dc_comm_unique_id: RCCL (librccl.so) could not be loaded", __FILE__, __LINE__);
  UniqueId id;
  if (int rc = r.get_unique_id(&id)) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

extern "C" int dc_comm_create(const void* id128, int rank, int world, void** comm) {
  if (comm == nullptr || world < 1 || rank < 0 || rank >= world) return dc_fail("dc_comm_create: bad argument", __FILE__, __LINE__);
  DcComm* c = new DcComm;
  c->rank = rank;
  c->world = world;
  if (world > 1) {
    if (id128 == nullptr) { delete c; return dc_fail("dc_comm_create: a unique id is needed for more than one rank", __FILE__, __LINE__); }
    Rccl& r = rccl();
    if (!r.ok) { delete c; return dc_fail("dc_comm_create: RCCL (librccl.so) could not be loaded", __FILE__, __LINE__); }
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    if (int rc = r.comm_init_rank(&c->nccl, world, id, rank)) { c->nccl = nullptr; delete c; return rccl_fail(rc, "ncclCommInitRank"); }
    c->owns_nccl = true;
  } else if (id128 != nullptr) {
    // a single rank WITH an id: bring RCCL up anyway (what the single-GPU test of the N > 1 call pattern wants to see)
    Rccl& r = rccl();
    if (!r.ok) { delete c; return dc_fail("dc_comm_create: RCCL (librccl.so) could not be loaded", __FILE__, __LINE__); }
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    if (int rc = r.comm_init_rank(&c->nccl, 1, id, 0)) { c->nccl = nullptr; delete c; return rccl_fail(rc, "ncclCommInitRank"); }
    c->owns_nccl = true;
  }
  if (c->nccl != nullptr)
    if (int e = make_stream(c)) { (void)dc_comm_destroy(c); return e; }      // the one owner of stream, events and communicator
  *comm = c;
  return 0;
}

extern "C" int dc_comm_adopt(void* nccl_comm, int rank, int world, void** comm) {
  if (comm == nullptr || nccl_comm == nullptr || world < 1 || rank < 0 || rank >= world) return dc_fail("dc_comm_adopt: bad argument", __FILE__, __LINE__);
  if (!rccl().ok) return dc_fail("dc_comm_adopt: RCCL (librccl.so) could not be loaded", __FILE__, __LINE__);
  DcComm* c = new DcComm;
  c->rank = rank;
  c->world = world;
  c->nccl = nccl_comm;
  if (int e = make_stream(c)) { (void)dc_comm_destroy(c); return e; }        // (an adopted communicator is not destroyed)
  *comm = c;
  return 0;
}

extern "C" int dc_comm_create_callback(dc_allreduce_callback fn, void* ctx, int rank, int world, int sync_stream, void** comm) {
  if (comm == nullptr || fn == nullptr || world < 1 || rank < 0 || rank >= world) return dc_fail("dc_comm_create_callback: bad argument", __FILE__, __LINE__);
  DcComm* c = new DcComm;
  c->rank = rank;
  c->world = world;
  c->cb = fn;
  c->cb_ctx = ctx;
  c->cb_sync = sync_stream;
  *comm = c;
  return 0;
}

extern "C" int dc_comm_destroy(void* comm) {
  DcComm* c = (DcComm*)comm;
  if (c == nullptr) return 0;
  // every member on its own: make_stream may have stopped half way
  if (c->stream != nullptr) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->ev_in != nullptr) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out != nullptr) (void)hipEventDestroy(c->ev_out);
  if (c->nccl != nullptr && c->owns_nccl) (void)rccl().comm_destroy(c->nccl);
  delete c;
  return 0;
}

extern "C" int dc_comm_info(void* comm, int* rank, int* world, int* transport, long* enqueued) {
  DcComm* c = (DcComm*)comm;
  if (c == nullptr) return dc_fail("dc_comm_info: null communicator", __FILE__, __LINE__);
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (transport) *transport = c->nccl != nullptr ? 1 : (c->cb != nullptr ? 2 : 0);
  if (enqueued) *enqueued = c->enqueued;
  return 0;
}

extern "C" int dc_grad_allreduce_enqueue(void* comm, void* buf, size_t count, int dtype, void* compute_stream) {
  DcComm* c = (DcComm*)comm;
  if (c == nullptr || (buf == nullptr && count != 0)) return dc_fail("dc_grad_allreduce_enqueue: null argument", __FILE__, __LINE__);
  if (dtype != DC_F32 && dtype != DC_BF16) return dc_fail("dc_grad_allreduce_enqueue: dtype must be DC_F32 or DC_BF16", __FILE__, __LINE__);
  ++c->enqueued;
  if (count == 0) return 0;
  if (c->cb != nullptr) {
    if (c->cb_sync) {
      hipError_t e = hipStreamSynchronize((hipStream_t)compute_stream);
      if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
    }
    if (int rc = c->cb(c->cb_ctx, buf, count, dtype)) return dc_fail("dc_grad_allreduce_enqueue: the transport callback failed", __FILE__, __LINE__), rc;
    return 0;
  }
  if (c->nccl == nullptr) return 0;     // one rank without RCCL: the sum over one rank is the buffer itself
  hipError_t e = hipEventRecord(c->ev_in, (hipStream_t)compute_stream);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  e = hipStreamWaitEvent(c->stream, c->ev_in, 0);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  if (int rc = rccl().all_reduce(buf, buf, count, dtype == DC_BF16 ? 9 : 7, /*ncclSum*/ 0, c->nccl, c->stream)) return rccl_fail(rc, "ncclAllReduce");
  return 0;
}

extern "C" int dc_grad_allreduce_wait(void* comm, void* compute_stream) {
  DcComm* c = (DcComm*)comm;
  if (c == nullptr) return dc_fail("dc_grad_allreduce_wait: null communicator", __FILE__, __LINE__);
  if (c->stream == nullptr) return 0;   // callback transport and the RCCL-less single rank complete inside enqueue
  hipError_t e = hipEventRecord(c->ev_out, c->stream);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  e = hipStreamWaitEvent((hipStream_t)compute_stream, c->ev_out, 0);
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}
