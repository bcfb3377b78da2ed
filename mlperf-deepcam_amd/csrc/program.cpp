// The launch list of a train step as a C object: dc_program_*.
//
// The reference's hot path is a chain of operator calls issued by its Python loop; here the chain is ~700 calls of this library's entry
// points per step, issued by engine.py.  A program records such a chain ONCE -- entry point + argument words, in issue order, stream
// fences included -- and replays it from C: one dc_program_run per step, no interpreter between two launches.  What a host in any
// language needs in order to drive the step is then this file's six functions plus the buffers the recorded pointers refer to.
//   * every int-returning entry point of include/deepcam_hip.h can be recorded (program_thunks.inc, generated from the binding table);
//   * an argument is stored as an 8-byte word; a word may instead name a SLOT whose value is bound before a run (dc_program_bind: a
//     batch pointer that changes from step to step);
//   * pointer arguments that refer to HOST memory (dc_conv_desc, the fold tables, pointer arrays of grouped launches) are recorded as
//     addresses: the recorder keeps those objects alive and unchanged for the life of the program (engine.py owns them);
//   * cross-stream order is part of the list: dc_stream_fence(from, to) = hipEventRecord + hipStreamWaitEvent on an event of a pool.
#include <hip/hip_runtime.h>
#include <string.h>

#include <string>
#include <mutex>
#include <vector>

#include "../../include/deepcam_hip.h"

extern "C" int dc_fail(const char* msg, const char* file, int line);
extern "C" int dc_set_error(int code, const char* file, int line);

union dc_arg {
  long long i;
  double d;
  void* p;
};
typedef int (*dc_thunk)(const dc_arg*);
struct ThunkEntry {
  const char* name;
  dc_thunk fn;
  int nargs;
  const char* types;
};
#include "program_thunks.inc"

namespace {
struct Op {
  const ThunkEntry* t;
  int first;      // index of the first argument word
};
struct Program {
  std::vector<Op> ops;
  std::vector<dc_arg> args;
  std::vector<int> slot;      // per argument word: -1 literal, >= 0 slot index
  std::vector<dc_arg> slots;
  std::vector<char> bound;
};
const ThunkEntry* find_thunk(const char* name) {
  for (const ThunkEntry& t : kThunks)
    if (strcmp(t.name, name) == 0) return &t;
  return nullptr;
}
}  // namespace

extern "C" int dc_program_create(void** out) {
  if (out == nullptr) return dc_fail("dc_program_create: null argument", __FILE__, __LINE__);
  *out = new Program();
  return 0;
}

extern "C" int dc_program_destroy(void* prog) {
  delete static_cast<Program*>(prog);
  return 0;
}

// Append one call: `name` is an entry point of this library, `words` its arguments in declaration order (ints sign-extended, floats as
// the bit pattern of a double, pointers as they are), `slots` NULL or one int per argument (-1: use the word; s >= 0: use the value bound to slot s at run time).
extern "C" int dc_program_append(void* prog, const char* name, int nargs, const long long* words, const int* slots) {
  Program* p = static_cast<Program*>(prog);
  if (p == nullptr || name == nullptr || (nargs > 0 && words == nullptr)) return dc_fail("dc_program_append: null argument", __FILE__, __LINE__);
  const ThunkEntry* t = find_thunk(name);
  if (t == nullptr) return dc_fail((std::string("dc_program_append: not a recordable entry point: ") + name).c_str(), __FILE__, __LINE__);
  if (t->nargs != nargs) return dc_fail((std::string("dc_program_append: wrong argument count for ") + name).c_str(), __FILE__, __LINE__);
  p->ops.push_back(Op{t, (int)p->args.size()});
  for (int i = 0; i < nargs; ++i) {
    dc_arg a;
    a.i = words[i];
    p->args.push_back(a);
    const int s = slots != nullptr ? slots[i] : -1;
    p->slot.push_back(s);
    if (s >= 0 && (size_t)s >= p->slots.size()) {
      p->slots.resize(s + 1);
      p->bound.resize(s + 1, 0);
    }
  }
  return 0;
}

extern "C" int dc_program_bind(void* prog, int slot, long long word) {
  Program* p = static_cast<Program*>(prog);
  if (p == nullptr || slot < 0 || (size_t)slot >= p->slots.size()) return dc_fail("dc_program_bind: no such slot", __FILE__, __LINE__);
  p->slots[slot].i = word;
  p->bound[slot] = 1;
  return 0;
}

extern "C" int dc_program_len(void* prog) { return prog == nullptr ? -1 : (int)static_cast<Program*>(prog)->ops.size(); }

// Name of call `index` (diagnostics: which launch failed); NULL when out of range.
extern "C" const char* dc_program_op_name(void* prog, int index) {
  Program* p = static_cast<Program*>(prog);
  if (p == nullptr || index < 0 || (size_t)index >= p->ops.size()) return nullptr;
  return p->ops[index].t->name;
}

// Issue every recorded call in order.  Stops at the first call that fails and returns its code (dc_last_error has the message);
// *failed_op, when given, receives that call's index (-1: all went through).
extern "C" int dc_program_run(void* prog, int* failed_op) {
  Program* p = static_cast<Program*>(prog);
  if (p == nullptr) return dc_fail("dc_program_run: null program", __FILE__, __LINE__);
  if (failed_op) *failed_op = -1;
  for (size_t s = 0; s < p->bound.size(); ++s)
    if (!p->bound[s]) return dc_fail("dc_program_run: a slot has no value (dc_program_bind)", __FILE__, __LINE__);
  dc_arg tmp[32];
  for (size_t k = 0; k < p->ops.size(); ++k) {
    const Op& op = p->ops[k];
    const int n = op.t->nargs;
    if (n > 32) return dc_fail("dc_program_run: too many arguments", __FILE__, __LINE__);
    for (int i = 0; i < n; ++i) {
      const int s = p->slot[op.first + i];
      tmp[i] = s >= 0 ? p->slots[s] : p->args[op.first + i];      // (a float argument's word is the bit pattern of a double)
    }
    const int rc = op.t->fn(tmp);
    if (rc != 0) {
      if (failed_op) *failed_op = (int)k;
      return rc;
    }
  }
  return 0;
}

// `to` waits for everything enqueued on `from` so far (hipEventRecord + hipStreamWaitEvent).  The events come from a pool that is reused
// round robin: a wait refers to the record that preceded it, so an event may be recorded again as soon as its wait has been enqueued.
// Thread-safe (the C ABI may be called from reader threads beside the main thread): the record + wait pair of one call runs under a mutex, so two
// callers can neither pick the same pool slot nor re-record an event between another caller's record and its wait.
extern "C" int dc_stream_fence(void* from_stream, void* to_stream) {
  constexpr int POOL = 64;
  static hipEvent_t pool[POOL];
  static bool made[POOL];
  static int next = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const int k = next;
  next = (next + 1) % POOL;
  if (!made[k]) {
    hipError_t e = hipEventCreateWithFlags(&pool[k], hipEventDisableTiming);
    if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
    made[k] = true;
  }
  hipError_t e = hipEventRecord(pool[k], (hipStream_t)from_stream);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  e = hipStreamWaitEvent((hipStream_t)to_stream, pool[k], 0);
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}

extern "C" int dc_memset_async(void* dst, int byte, size_t bytes, void* stream) {
  if (dst == nullptr && bytes != 0) return dc_fail("dc_memset_async: null argument", __FILE__, __LINE__);
  if (bytes == 0) return 0;
  hipError_t e = hipMemsetAsync(dst, byte, bytes, (hipStream_t)stream);
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}
