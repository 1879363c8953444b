// Error text + version for libbuglab_hip.
#include <stdarg.h>

#include "bl_common.h"

static thread_local char g_err[512] = "";

void bl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* bl_last_error(void) { return g_err; }
extern "C" int bl_version(void) { return 1; }

// ---- deterministic-gradient mode ----------------------------------------------------------------------------------
// Kernels that accumulate into one address from several workgroups (split-K weight gradients, column sums) flush with
// fp32 atomics: the sum depends on the order the workgroups happen to arrive in.  In deterministic mode every such
// flush takes its TURN -- a counter per output tile, workgroup t adds after workgroup t-1 (bl_ordered_enter/leave in
// bl_common.h) -- so the order, and with it every bit of the gradient, is the same in every run.  The counters come
// from a per-device ring that is zeroed on the launch stream right before the kernel.
#include <stdlib.h>

static int g_deterministic = -1;

extern "C" void bl_set_deterministic(int32_t on) { g_deterministic = on ? 1 : 0; }
// compute units of the current device (256 on MI355X; cached -- the library serves one device per process)
int bl_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount
              : 256;
  }
  return ncu;
}

// LDS bytes one workgroup may declare on the CURRENT device (160 KiB on gfx950); queried per call -- a process may drive more
// than one device
int bl_max_lds_per_block() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
  return v;
}

extern "C" int32_t bl_get_deterministic(void) {
  if (g_deterministic < 0) {
    const char* e = getenv("BL_DETERMINISTIC");
    g_deterministic = (e && e[0] && e[0] != '0') ? 1 : 0;
  }
  return g_deterministic;
}

unsigned* bl_order_counters(int n, void* stream) {
  if (!bl_get_deterministic() || n <= 0) return nullptr;
  constexpr int RING = 1 << 20, MAXDEV = 16;
  static unsigned* ring[MAXDEV] = {nullptr};
  static int pos[MAXDEV] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV || n > RING) return nullptr;
  if (ring[dev] == nullptr && hipMalloc((void**)&ring[dev], (size_t)RING * sizeof(unsigned)) != hipSuccess) return nullptr;
  if (pos[dev] + n > RING) pos[dev] = 0;  // thousands of launches later: whatever used the start of the ring is long done
  unsigned* out = ring[dev] + pos[dev];
  pos[dev] += (n + 31) / 32 * 32;
  if (hipMemsetAsync(out, 0, (size_t)n * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return nullptr;
  return out;
}
