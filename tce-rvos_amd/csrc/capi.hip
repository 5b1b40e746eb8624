// ABI housekeeping: version, last-error string, hipGraph capture helpers.
#include <stdarg.h>
#include <stdlib.h>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "../../include/tce_rvos.h"
#include "../../include/tce_rvos_debug.h"

static thread_local char g_err[512] = "";

void tce_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// One range flag per device: a kernel launched on device B must never store to device A's flag.  The
// launch wrappers look the flag up from the calling thread's current device; registration files the pointer under
// the device that owns it (NULL: disables the check on the current device).
#define TCE_MAX_DEVICES 64
static int* g_range_flag[TCE_MAX_DEVICES] = {nullptr};
int* tce_range_flag() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TCE_MAX_DEVICES) return nullptr;
  return g_range_flag[dev];
}
extern "C" int tce_set_range_flag(int32_t* flag) {
  int dev = 0;
  if (flag != nullptr) {
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, flag);
    TCE_CHECK_ARG(e == hipSuccess && at.type == hipMemoryTypeDevice, "tce_set_range_flag: flag must be device memory");
    dev = at.device;
  } else {
    (void)hipGetDevice(&dev);
  }
  TCE_CHECK_ARG(dev >= 0 && dev < TCE_MAX_DEVICES, "tce_set_range_flag: device index out of range");
  g_range_flag[dev] = flag;
  return TCE_OK;
}

// 2: round 2 changed tce_embed_ln_f32 / the GroupNorm workspace size and dropped three round-1 entries
// 3: per-device range flag, window_attn3d on the matrix cores, per-site arithmetic (round 3)
extern "C" int tce_abi_version(void) { return 5; }
extern "C" const char* tce_last_error(void) { return g_err; }

extern "C" int tce_graph_begin(tceStream stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    tce_set_error("tce_graph_begin: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  return TCE_OK;
}

extern "C" int tce_graph_end(tceStream stream, void** graph_exec_out) {
  TCE_CHECK_ARG(graph_exec_out != nullptr, "tce_graph_end: null output");
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    tce_set_error("tce_graph_end: capture failed: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    tce_set_error("tce_graph_end: instantiate failed: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  *graph_exec_out = (void*)exec;
  return TCE_OK;
}

extern "C" int tce_graph_launch(void* graph_exec, tceStream stream) {
  TCE_CHECK_ARG(graph_exec != nullptr, "tce_graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    tce_set_error("tce_graph_launch: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  return TCE_OK;
}

// One executable that runs n captured graphs as n independent components of ONE flat graph: the nodes of every input are
// re-created in the new graph with their parameters and edges (child-graph nodes were tried first: the executor runs a
// child as a single chain, which loses the clip's own parallel branches -- 9.5 ms per clip against 7.1).
static hipError_t tce_clone_into(hipGraph_t dst, hipGraph_t src) {
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(src, nullptr, &n);
  if (e != hipSuccess) return e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && (e = hipGraphGetNodes(src, nodes.data(), &n)) != hipSuccess) return e;
  std::unordered_map<hipGraphNode_t, hipGraphNode_t> twin;
  for (hipGraphNode_t nd : nodes) {
    hipGraphNodeType ty;
    if ((e = hipGraphNodeGetType(nd, &ty)) != hipSuccess) return e;
    hipGraphNode_t nn = nullptr;
    if (ty == hipGraphNodeTypeKernel) {
      hipKernelNodeParams kp;
      if ((e = hipGraphKernelNodeGetParams(nd, &kp)) != hipSuccess) return e;
      e = hipGraphAddKernelNode(&nn, dst, nullptr, 0, &kp);
    } else if (ty == hipGraphNodeTypeMemset) {
      hipMemsetParams mp;
      if ((e = hipGraphMemsetNodeGetParams(nd, &mp)) != hipSuccess) return e;
      e = hipGraphAddMemsetNode(&nn, dst, nullptr, 0, &mp);
    } else if (ty == hipGraphNodeTypeEmpty) {
      e = hipGraphAddEmptyNode(&nn, dst, nullptr, 0);
    } else {
      tce_set_error("tce_graph_group: node type %d cannot be re-created", (int)ty);
      return hipErrorNotSupported;
    }
    if (e != hipSuccess) return e;
    twin[nd] = nn;
  }
  size_t ne = 0;
  if ((e = hipGraphGetEdges(src, nullptr, nullptr, &ne)) != hipSuccess) return e;
  if (ne) {
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if ((e = hipGraphGetEdges(src, from.data(), to.data(), &ne)) != hipSuccess) return e;
    for (size_t i = 0; i < ne; ++i) { from[i] = twin.at(from[i]); to[i] = twin.at(to[i]); }
    e = hipGraphAddDependencies(dst, from.data(), to.data(), ne);
  }
  return e;
}

extern "C" int tce_graph_group(void* const* graphs, int n, void** graph_exec_out) {
  TCE_CHECK_ARG(graphs != nullptr && n >= 1 && n <= 16 && graph_exec_out != nullptr, "tce_graph_group: bad arguments");
  for (int i = 0; i < n; ++i) TCE_CHECK_ARG(graphs[i] != nullptr, "tce_graph_group: graph %d is null", i);
  hipGraph_t parent = nullptr;
  hipError_t e = hipGraphCreate(&parent, 0);
  if (e != hipSuccess) {
    tce_set_error("tce_graph_group: hipGraphCreate: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  const bool child = getenv("TCE_GROUP_CHILD") != nullptr;  // A/B: child-graph nodes instead of the flat copy
  for (int i = 0; i < n && e == hipSuccess; ++i) {
    if (child) {
      hipGraphNode_t node = nullptr;
      e = hipGraphAddChildGraphNode(&node, parent, nullptr, 0, (hipGraph_t)graphs[i]);
    } else {
      e = tce_clone_into(parent, (hipGraph_t)graphs[i]);
    }
  }
  hipGraphExec_t exec = nullptr;
  if (e == hipSuccess) e = hipGraphInstantiate(&exec, parent, nullptr, nullptr, 0);
  (void)hipGraphDestroy(parent);
  if (e != hipSuccess) {
    if (e != hipErrorNotSupported) tce_set_error("tce_graph_group: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  *graph_exec_out = (void*)exec;
  return TCE_OK;
}

extern "C" int tce_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return TCE_OK;
}
