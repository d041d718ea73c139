// Library-wide state that exists once (not part of the bf16 / fp16 twin builds): error text, ABI version.
#include <stdarg.h>
#include "common.hpp"

static thread_local char g_err[512] = "";
void brats_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* brats_last_error(void) { return g_err; }
extern "C" int brats_abi_version(void) { return BRATS_ABI_VERSION; }  // include/brats_hip.h
