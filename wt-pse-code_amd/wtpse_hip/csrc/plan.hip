// Launch plans: a recorded sequence of C-ABI calls (plus cross-stream dependencies) replayed from native code.
//
// A training step is ~1 900 kernel launches.  Issued one by one from Python (ctypes marshalling, output allocation,
// bookkeeping) they cost the host 25-30 us each, i.e. about as long as the GPU needs for the step; hipGraph replay on this
// runtime is no cheaper (profiles/r02_hipgraph_vs_eager.txt).  A plan stores, for every call of the step, the entry point
// and its argument values (device addresses pinned by recording under stream capture, see wtpse_hip/step.py) and replays
// them with one host call: what is left per launch is the hipLaunchKernel itself.
// The entry points are called through thunks generated from include/wtpse_hip.h at build time (_plan_thunks.inc).
#include <hip/hip_runtime.h>
#include <vector>
#include "common.h"

union PlanArg {
  void* p;
  long long i;
  unsigned long long u;
  double d;
};

#include "_plan_thunks.inc"   // extern "C" prototypes, thunk_<k>(const PlanArg*, void* stream), PLAN_FN_NAMES[], PLAN_THUNKS[], PLAN_NFN

namespace {
struct Cmd {
  int fn;              // >= 0: thunk index; -1: `stream` waits for everything issued so far on `other`
  int nargs;
  void* stream;
  void* other;
  hipEvent_t ev;
  PlanArg args[40];
};
struct Plan {
  std::vector<Cmd> cmds;
  int tuning;          // wtpse_tuning_state() when the plan was created: the recorded calls were sized for this tiling / weight format
};
}  // namespace

extern "C" int wtpse_tuning_state(void);      // conv_x3.hip

extern "C" int wtpse_plan_fn_count(void) { return PLAN_NFN; }
extern "C" const char* wtpse_plan_fn_name(int id) { return (id >= 0 && id < PLAN_NFN) ? PLAN_FN_NAMES[id] : ""; }

extern "C" void* wtpse_plan_create(void) {
  Plan* p = new Plan();
  p->tuning = wtpse_tuning_state();
  return p;
}

extern "C" int wtpse_plan_destroy(void* plan) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p) return WTPSE_EINVAL;
  for (Cmd& c : p->cmds)
    if (c.fn < 0 && c.ev) (void)hipEventDestroy(c.ev);
  delete p;
  return WTPSE_OK;
}

extern "C" int wtpse_plan_size(void* plan) { return plan ? (int)static_cast<Plan*>(plan)->cmds.size() : -1; }

extern "C" int wtpse_plan_add_call(void* plan, int fn, const void* args, int nargs, void* stream) {
  Plan* p = static_cast<Plan*>(plan);
  WTPSE_REQUIRE(p && fn >= 0 && fn < PLAN_NFN && nargs >= 0 && nargs <= 40 && (args || nargs == 0));
  Cmd c;
  c.fn = fn; c.nargs = nargs; c.stream = stream; c.other = nullptr; c.ev = nullptr;
  const PlanArg* a = static_cast<const PlanArg*>(args);
  for (int i = 0; i < nargs; ++i) c.args[i] = a[i];
  p->cmds.push_back(c);
  return WTPSE_OK;
}

extern "C" int wtpse_plan_add_wait(void* plan, void* waiter, void* waited) {
  Plan* p = static_cast<Plan*>(plan);
  WTPSE_REQUIRE(p && waiter != waited);
  Cmd c;
  c.fn = -1; c.nargs = 0; c.stream = waiter; c.other = waited; c.ev = nullptr;
  hipError_t e = hipEventCreateWithFlags(&c.ev, hipEventDisableTiming);
  if (e != hipSuccess) return (int)e;
  p->cmds.push_back(c);
  return WTPSE_OK;
}

// Issue the recorded calls on their recorded streams.  Returns the first non-zero status (and stops there).
extern "C" int wtpse_plan_replay(void* plan) {
  Plan* p = static_cast<Plan*>(plan);
  WTPSE_REQUIRE(p);
  // wtpse_x3_terms / wtpse_x3r_enable / wtpse_x3_xcd changed since the recording: the recorded buffer sizes may no longer fit the
  // tiling the entry points would choose now, the packed weights may be in another format
  if (p->tuning != wtpse_tuning_state()) return WTPSE_ESTATE;
  for (Cmd& c : p->cmds) {
    if (c.fn >= 0) {
      const int rc = PLAN_THUNKS[c.fn](c.args, c.stream);
      if (rc) return rc;
    } else {
      hipError_t e = hipEventRecord(c.ev, (hipStream_t)c.other);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)c.stream, c.ev, 0);
      if (e != hipSuccess) return (int)e;
    }
  }
  return WTPSE_OK;
}
