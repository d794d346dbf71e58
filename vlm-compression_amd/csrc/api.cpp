// Error plumbing and ABI version for the C-ABI library (include/vlmc.h).
#include "common.hpp"

namespace vlmc {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local LaunchEvents g_events = {nullptr, nullptr};
LaunchEvents take_launch_events() {
    const LaunchEvents e = g_events;
    g_events = {nullptr, nullptr};
    return e;
}
}  // namespace vlmc

extern "C" int vlmc_abi_version(void) { return VLMC_ABI_VERSION; }
extern "C" const char *vlmc_last_error(void) { return vlmc::g_err; }
extern "C" void vlmc_set_launch_events(void *start_event, void *stop_event) {
    vlmc::g_events = {static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)};
}
