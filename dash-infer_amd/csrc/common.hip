// common.hip -- version, error text and device queries of libdashinfer_hip.so
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "device_utils.h"

namespace dihip {

static thread_local char g_last_error[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

int cached_num_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      return n;
    (void)hipGetLastError();
    return 0;
  }();
  return cus;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
bool env_off(const char* name) {
  const char* e = getenv(name);
  return e && e[0] == '0';
}

static unsigned long long* g_trace = nullptr;
static size_t g_trace_bytes = 0;
unsigned long long* debug_trace_buffer(size_t bytes) { return (g_trace && g_trace_bytes >= bytes) ? g_trace : nullptr; }
void debug_set_trace(void* buf, size_t bytes) {
  g_trace = reinterpret_cast<unsigned long long*>(buf);
  g_trace_bytes = buf ? bytes : 0;
}

}  // namespace dihip

extern "C" {

int dihip_debug_set_trace(void* buf, size_t bytes) {
  dihip::debug_set_trace(buf, bytes);
  return DIHIP_SUCCESS;
}


const char* dihip_version(void) { return "dashinfer-hip 0.1.0 (gfx950)"; }

const char* dihip_last_error(void) { return dihip::g_last_error; }

int dihip_device_info(int* num_cus, int* lds_bytes_per_cu, char* name, size_t name_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    dihip::set_last_error("no HIP device available");
    return DIHIP_RUNTIME_ERROR;
  }
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return DIHIP_SUCCESS;
}

}  // extern "C"
